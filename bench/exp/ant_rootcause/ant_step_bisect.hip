// One environment step (advance() of csrc/ant_sim.h: the sub-step loop around accelerations()) for the states of
// states_contact.bin, as csrc/ant_env.hip's kernel calls it.  Writes state (29) + reward (1) per environment.
// Variants are chosen on the command line of the build (build_step.sh): optimisation level, -DWITH_RESTRICT
// (`__restrict__` on the parameter block and the action pointer of advance() / accelerations()), -DHOIST (without the empty
// asm through which csrc/ant_sim.h passes the parameter pointer in every sub-step, so that no load from it can be
// hoisted out of the loop), and the backend's spill options.
#include <hip/hip_runtime.h>
#ifdef WITH_RESTRICT        // the qualifier that csrc/ant_sim.h no longer carries
#define ASIM_RESTRICT __restrict__
#endif
#ifdef DUMP_SYSTEM          // also writes the first sub-step's system (M, Q) and solution behind the 30 state values
#define ASIM_DEBUG
#endif
#ifdef HOIST               // without the empty asm that keeps the parameter loads inside the sub-step loop
#define ASIM_LOOP_BODY_BEGIN(P)
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ant_sim.h"

#ifdef DUMP_SYSTEM
constexpr int OUT = 30 + 133;
#else
constexpr int OUT = 30;
#endif

__global__ __launch_bounds__(64) void step_kernel(const AntParams* Pin, const float* st_all, const float* act, float* out, int E) {
    using namespace ant;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const AntParams* __restrict__ P = Pin;
    const float* st = st_all + (size_t)e * STRIDE;
    float pos[3], quat[4], qj[8], u[N], wrench[L][6];
#pragma unroll
    for (int i = 0; i < 3; ++i) pos[i] = st[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) quat[i] = st[3 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) qj[i] = st[7 + i];
#pragma unroll
    for (int i = 0; i < N; ++i) u[i] = st[15 + i];
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
    float rew;
    bool term;
    float* o = out + (size_t)e * OUT;
#ifdef DUMP_SYSTEM
    advance(P, pos, quat, qj, u, act + (size_t)e * 8, wrench, rew, term, o + 30);
#else
    advance(P, pos, quat, qj, u, act + (size_t)e * 8, wrench, rew, term);
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = pos[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[3 + i] = quat[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[7 + i] = qj[i];
#pragma unroll
    for (int i = 0; i < N; ++i) o[15 + i] = u[i];
    o[29] = rew;
}

template <typename T>
static std::vector<T> slurp(const char* dir, const char* name) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    auto pb = slurp<char>(argv[1], "params.bin");
    auto st = slurp<float>(argv[1], "states_contact.bin");
    auto act = slurp<float>(argv[1], "actions.bin");
    if (pb.size() != sizeof(AntParams)) return 2;
    const int E = (int)st.size() / ant::STRIDE;
    void *dP, *dS, *dA, *dO;
    CK(hipMalloc(&dP, pb.size())); CK(hipMalloc(&dS, st.size() * 4)); CK(hipMalloc(&dA, (size_t)E * 8 * 4)); CK(hipMalloc(&dO, (size_t)E * OUT * 4));
    CK(hipMemcpy(dP, pb.data(), pb.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dS, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, act.data(), (size_t)E * 8 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dO, 0, (size_t)E * OUT * 4));
    hipLaunchKernelGGL(step_kernel, dim3((E + 63) / 64), dim3(64), 0, 0, (const AntParams*)dP, (const float*)dS, (const float*)dA, (float*)dO, E);
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)E * OUT);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}

// accelerations() of csrc/ant_sim.h for the states of states.bin on the device: writes, per environment, the lower
// triangle of the 14 x 14 system, its right-hand side and the solution (105 + 14 + 14 floats) to argv[2].
// Built once per optimisation level (run.sh); compare.py holds each build against the host build of the same code.
#define ASIM_DEBUG
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ant_sim.h"

constexpr int OUT = 105 + 14 + 14;

__global__ __launch_bounds__(64) void acc_kernel(const AntParams* P, const float* st_all, const float* act, float* out, int E) {
    using namespace ant;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float* st = st_all + (size_t)e * STRIDE;
    float pos[3], quat[4], qj[8], u[N], tau[8], ud[N], wrench[L][6];
#pragma unroll
    for (int i = 0; i < 3; ++i) pos[i] = st[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) quat[i] = st[3 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) qj[i] = st[7 + i];
#pragma unroll
    for (int i = 0; i < N; ++i) u[i] = st[15 + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) tau[j] = P->gear * act[(size_t)e * 8 + j];
    const float h = P->dt / (float)P->substeps;
    float* o = out + (size_t)e * OUT;
    accelerations(P, pos, quat, qj, u, tau, h, ud, wrench, o);
#pragma unroll
    for (int i = 0; i < N; ++i) o[119 + i] = ud[i];
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
    hipLaunchKernelGGL(acc_kernel, dim3((E + 63) / 64), dim3(64), 0, 0, (const AntParams*)dP, (const float*)dS, (const float*)dA, (float*)dO, E);
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)E * OUT);
    CK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    printf("%s: %d environments\n", argv[2], E);
    return 0;
}

// Experiment: how fast can ONE workgroup stream a 400x416 bf16 weight matrix through
// v_mfma_f32_16x16x32_bf16, by weight layout and wave count?  (hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 13, NT = 25, LDW = 416, LDA = 424;

// MODE 0: row-major W[n][k]; MODE 1: fragment order [nt][ks][lane][8]
template <int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV) void k(const __bf16* __restrict__ W, int layers, float* out, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) __bf16 A[16 * LDA];
    for (int i = threadIdx.x; i < 16 * LDA; i += blockDim.x) A[i] = (__bf16)(0.01f * (i % 7));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kg = lane >> 4;
    f32x4 tot = {0, 0, 0, 0};
    unsigned long long t0 = wall_clock64();
    for (int l = 0; l < layers; ++l) {
        const __bf16* Wl = W + (size_t)l * NT * 16 * LDW;
        bf16x8 b0[KS], b1[KS];
        auto load = [&](bf16x8(&b)[KS], int nt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (MODE == 0) b[ks] = *reinterpret_cast<const bf16x8*>(Wl + (size_t)(nt * 16 + c) * LDW + ks * 32 + 8 * kg);
                else b[ks] = *reinterpret_cast<const bf16x8*>(Wl + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
            }
        };
        auto mma = [&](bf16x8(&b)[KS]) {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + c * LDA + ks * 32 + 8 * kg);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[ks], acc, 0, 0, 0);
            }
            tot += acc;
        };
        int nt = wave;
        if (nt < NT) load(b0, nt);
        for (; nt < NT; nt += 2 * NWV) {
            const bool has1 = nt + NWV < NT;
            if (has1) load(b1, nt + NWV);
            mma(b0);
            if (has1) {
                if (nt + 2 * NWV < NT) load(b0, nt + 2 * NWV);
                mma(b1);
            }
        }
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int MODE, int NWV>
void run(const char* name, const __bf16* W, int layers, float* out, unsigned long long* st, int grid) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<MODE, NWV>), dim3(grid), dim3(64 * NWV), 0, 0, W, layers, out, st);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), st, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-28s grid %3d waves %2d : %.2f us per 400x416 layer (mean over WGs), %.1f GB/s per WG\n", name, grid, NWV,
           s / grid / 100.0 / layers, (NT * 16 * LDW * 2.0) / (s / grid / 100.0 / layers * 1e-6) / 1e9);
}

int main() {
    const int layers = 8;
    size_t n = (size_t)layers * NT * 16 * LDW;
    __bf16* W; float* out; unsigned long long* st;
    hipMalloc(&W, n * 2); hipMemset(W, 0, n * 2);
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 256 * 8);
    for (int grid : {16, 8, 64}) {
        run<0, 8>("row-major (half lines)", W, layers, out, st, grid);
        run<1, 8>("fragment order (1 KB/load)", W, layers, out, st, grid);
        run<0, 4>("row-major (half lines)", W, layers, out, st, grid);
        run<1, 4>("fragment order (1 KB/load)", W, layers, out, st, grid);
        run<0, 16>("row-major (half lines)", W, layers, out, st, grid);
        run<1, 16>("fragment order (1 KB/load)", W, layers, out, st, grid);
    }
    return 0;
}

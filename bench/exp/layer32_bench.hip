// Micro-benchmark + numerics check of csrc/mlp32_dev.h: a chain of fp32 layers on one row tile per
// workgroup, weights streamed from the packed copies, as in the fp32 phase kernels.
//   hipcc -O3 --offload-arch=gfx950 -o layer32_bench layer32_bench.hip && ./layer32_bench
// Prints microseconds per layer for several (waves, rows per tile, K-split, grid) settings.
#include "../../d4pg_pytorch_amd/csrc/mlp32_dev.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <int NW, int MT, int NCH0, int NCH1>
__global__ __launch_bounds__(64 * NW) void chain_kernel(LayerW32 L0, LayerW32 L1, int pairs, int lda, const float* x0,
                                                        float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 4 * MT;
    float* bufA = (float*)smem;
    float* bufB = bufA + M * lda;
    float* red = bufB + M * lda;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Pre32 t0 = fetch_first<NCH0>(L0, wave, lane), t1;
    for (int i = threadIdx.x; i < M * lda; i += blockDim.x) {
        bufA[i] = x0[(size_t)blockIdx.x * M * lda + i];
        bufB[i] = 0.f;
    }
    __syncthreads();
    for (int it = 0; it < pairs; ++it) {
        layer32<NW, MT, NCH0, false>(t0, bufA, lda, L0, red, [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bufB[(r0 + r) * lda + col] = fmaxf(v[r], 0.f);
        }, [&] { t1 = fetch_first<NCH1>(L1, wave, lane); });
        __syncthreads();
        layer32<NW, MT, NCH1, false>(t1, bufB, lda, L1, red, [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bufA[(r0 + r) * lda + col] = fmaxf(v[r], 0.f);
        }, [&] { t0 = fetch_first<NCH0>(L0, wave, lane); });
        __syncthreads();
    }
    for (int i = threadIdx.x; i < M * lda; i += blockDim.x) out[(size_t)blockIdx.x * M * lda + i] = bufA[i];
}

struct HostLayer {
    int N, K, ks, ntiles;
    std::vector<float> W, b, packed;
};
static HostLayer make_layer(int N, int K, unsigned seed, int ksplit = 1) {
    HostLayer h;
    h.N = N, h.K = K, h.ks = ((K + 31) / 32 + ksplit - 1) / ksplit * ksplit, h.ntiles = (N + 63) / 64;
    h.W.resize((size_t)N * K), h.b.resize(N);
    srand(seed);
    const float bound = 1.f / sqrtf((float)K);
    for (auto& w : h.W) w = (2.f * rand() / RAND_MAX - 1.f) * bound;
    for (auto& w : h.b) w = (2.f * rand() / RAND_MAX - 1.f) * bound;
    h.packed.assign((size_t)h.ntiles * h.ks * CHUNK_FLOATS, 0.f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) h.packed[packed32_off(n, k, h.ks)] = h.W[(size_t)n * K + k];
    return h;
}
static LayerW32 upload(const HostLayer& h, int ksplit) {
    LayerW32 L;
    float *w, *b;
    CK(hipMalloc(&w, h.packed.size() * 4));
    CK(hipMalloc(&b, h.b.size() * 4));
    CK(hipMemcpy(w, h.packed.data(), h.packed.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.b.data(), h.b.size() * 4, hipMemcpyHostToDevice));
    L.W = w, L.bias = b, L.ks = h.ks, L.ntiles = h.ntiles, L.N = h.N, L.ksplit = ksplit, L.nch = h.ks / ksplit, L.kg = 8 * L.nch;
    return L;
}

template <int NW, int MT, int NCH0, int NCH1>
static void run(const HostLayer& h0, const HostLayer& h1, int ks0, int ks1, int grid, int pairs, bool check) {
    constexpr int M = 4 * MT;
    const int width = std::max(std::max(h0.K, h0.N), std::max(h1.K, h1.N));
    int lda = (width + 31) / 32 * 32 + 32;
    while (lda % 64 < 4 || lda % 64 > 20) lda += 4;     // 4 rows of one wide LDS read on disjoint banks
    LayerW32 L0 = upload(h0, ks0), L1 = upload(h1, ks1);
    if (std::max(L0.ntiles, L0.ksplit) > NW || std::max(L1.ntiles, L1.ksplit) > NW || L0.nch != NCH0 || L1.nch != NCH1 ||
        (L0.ksplit > 1 && L0.ntiles > 1) || (L1.ksplit > 1 && L1.ntiles > 1)) {
        printf("skip: bad decomposition\n");
        return;
    }
    std::vector<float> x((size_t)grid * M * lda, 0.f);
    for (int g = 0; g < grid; ++g)
        for (int r = 0; r < M; ++r)
            for (int k = 0; k < h0.K; ++k) x[((size_t)g * M + r) * lda + k] = 2.f * rand() / RAND_MAX - 1.f;
    float *dx, *dout;
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMalloc(&dout, x.size() * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    const int lds = (2 * M * lda + NW * MT * 256) * 4;
    auto kern = chain_kernel<NW, MT, NCH0, NCH1>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (check) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, L0, L1, 1, lda, dx, dout);
        CK(hipDeviceSynchronize());
        std::vector<float> out(x.size());
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int g = 0; g < grid; g += std::max(1, grid / 3)) {
            for (int r = 0; r < M; ++r) {
                std::vector<double> y1(h0.N), y2(h1.N);
                for (int n = 0; n < h0.N; ++n) {
                    double s = h0.b[n];
                    for (int k = 0; k < h0.K; ++k) s += (double)x[((size_t)g * M + r) * lda + k] * h0.W[(size_t)n * h0.K + k];
                    y1[n] = s > 0 ? s : 0;
                }
                for (int n = 0; n < h1.N; ++n) {
                    double s = h1.b[n];
                    for (int k = 0; k < h1.K; ++k) s += y1[k] * h1.W[(size_t)n * h1.K + k];
                    y2[n] = s > 0 ? s : 0;
                    worst = std::max(worst, fabs(y2[n] - (double)out[((size_t)g * M + r) * lda + n]));
                }
            }
        }
        printf("  check NW=%d MT=%d ksplit=%d/%d: max |err| vs fp64 = %.3g\n", NW, MT, ks0, ks1, worst);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, L0, L1, pairs, lda, dx, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    printf("NW=%2d rows=%2d  %dx%d (ksplit %d) / %dx%d (ksplit %d)  grid=%3d : %.2f us per layer pair, %.2f us per layer\n", NW,
           M, h0.N, h0.K, ks0, h1.N, h1.K, ks1, grid, 1000.f * best / pairs, 500.f * best / pairs);
    CK(hipFree(dx));
    CK(hipFree(dout));
    CK(hipFree((void*)L0.W));
    CK(hipFree((void*)L0.bias));
    CK(hipFree((void*)L1.W));
    CK(hipFree((void*)L1.bias));
}

int main() {
    HostLayer a = make_layer(400, 400, 1), b = make_layer(400, 400, 2);
    const int pairs = 64;
#ifndef D4PG_EXP_NO_MFMA
    run<8, 1, 13, 13>(a, b, 1, 1, 4, pairs, true);
#endif
    for (int grid : {1, 64}) {
        run<8, 1, 13, 13>(a, b, 1, 1, grid, pairs, false);
        run<8, 2, 13, 13>(a, b, 1, 1, grid, pairs, false);
    }
    return 0;
}

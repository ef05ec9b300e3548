// Grouped small-GEMM kernel for the actor/critic MLPs on gfx950.
//
// One launch executes up to MAX_GROUP independent problems  C = epi(A·B + bias)
// (e.g. layer 1 of the target actor, the online actor and the online critic, or
// the dW and dX products of one layer's backward).  The shapes are tiny
// (B=256, H=400, N=51), so the design goal is latency, not peak throughput:
//
//  * output tile 32x32 on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain);
//  * a 256-thread workgroup = 4 waves that SPLIT K of one tile and combine their
//    accumulators through LDS, so a 256x400x400 layer yields 104 workgroups with
//    the K loop 4x shorter (one wave per SIMD, every load of the K range in
//    flight at once);
//  * operands go global -> VGPR directly (every matrix is L2 resident): a
//    K-contiguous operand is read as one 16-byte load per lane per 8 k, an
//    M/N-contiguous operand as 128-byte row segments per half wave;
//  * bias, ReLU / tanh, ReLU-mask and tanh-gradient multiplies, and the bias
//    gradient (column sums of dY) are fused into the tile epilogue;
//  * blockIdx -> tile mapping gives each XCD a contiguous range of tiles so
//    tiles that share an operand panel share an L2.
//
// Layout conventions (floats): every leading dimension is a multiple of 4 and
// base pointers of K-contiguous operands are 16-byte aligned.
//
//   GEMM_NT  forward     C[m,n] = sum_k A[m*lda+k] * B[n*ldb+k]      (x · W^T)
//   GEMM_NN  dX          C[m,n] = sum_k A[m*lda+k] * B[k*ldb+n]      (dY · W)
//   GEMM_TN  dW          C[m,n] = sum_k A[k*lda+m] * B[k*ldb+n]      (dY^T · x)
#include "common.h"

enum { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_TANH = 2, EPI_MUL_RELU_MASK = 3, EPI_MUL_TANH_GRAD = 4 };

struct GemmDesc {
    const float* A;
    const float* B;
    float* C;
    const float* bias;    // [N] added before the activation (may be null)
    const float* aux;     // [M, ldaux] operand of the mask epilogues (may be null)
    float* colsum;        // GEMM_TN only: out[m] = sum_k A(m,k)  (bias gradient; may be null)
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int kind, epi;
    int tiles_m, tiles_n;
    int tile_base;        // first global tile id of this problem
    int pad_;
};

constexpr int MAX_GROUP = 6;
struct GemmGroup {
    GemmDesc p[MAX_GROUP];
    int count;
    int total_tiles;
};

constexpr int CH = 6;   // k-units (of 8) per register chunk; two chunks are in flight

template <bool KC>
__device__ __forceinline__ f32x4 load_frag(const float* __restrict__ P, int ld, int row_c, bool ok,
                                           int k, int K) {
    // KC:  P(row, k) = P[row*ld + k]   (one 16-byte load)
    // !KC: P(row, k) = P[k*ld + row]   (four 4-byte loads, coalesced across lanes)
    // Branch-free: every lane always loads from a clamped, in-bounds address and the
    // out-of-range elements are zeroed with selects.  (Predicating the loads instead
    // puts each one in its own exec-masked block and the compiler then waits for
    // every load before issuing the next: the K loop degenerates into serial L2
    // round trips.)
    f32x4 v;
    if (KC) {
        const int ks = k < K ? k : 0;
        v = *reinterpret_cast<const f32x4*>(P + (size_t)row_c * ld + ks);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ks = k + t < K ? k + t : 0;
            v[t] = P[(size_t)ks * ld + row_c];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = (ok && k + t < K) ? v[t] : 0.f;
    return v;
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void load_chunk(const GemmDesc& d, int m_c, bool m_ok, int n_c, bool n_ok, int h,
                                           int u, int u1, f32x4 (&a)[CH], f32x4 (&b)[CH]) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int uu = u + j;
        const int k = uu < u1 ? uu * 8 + 4 * h : d.K;   // k >= K: loads a dummy in-bounds element, zeroed
        a[j] = load_frag<AKC>(d.A, d.lda, m_c, m_ok, k, d.K);
        b[j] = load_frag<BKC>(d.B, d.ldb, n_c, n_ok, k, d.K);
    }
}

__device__ __forceinline__ void mma_chunk(f32x16& acc, float& asum, int u, int u1, const f32x4 (&a)[CH],
                                          const f32x4 (&b)[CH]) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        if (u + j < u1) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][t], b[j][t], acc, 0, 0, 0);
                asum += a[j][t];
            }
        }
    }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile(const GemmDesc& d, int tm, int tn, float* red, float* cs) {
    const int lane = threadIdx.x & 63;
    // readfirstlane makes the wave index (and the K range derived from it) provably
    // wave-uniform, so the K-loop bookkeeping stays in scalar registers / branches
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m = tm * 32 + r, n = tn * 32 + r;
    const bool m_ok = m < d.M, n_ok = n < d.N;
    const int m_c = m_ok ? m : d.M - 1, n_c = n_ok ? n : d.N - 1;   // clamped rows for the loads

    const int units = (d.K + 7) >> 3;
    const int per = (units + 3) >> 2;
    const int u0 = wave * per;
    const int u1 = min(units, u0 + per);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float asum = 0.f;

    f32x4 aA[CH], bA[CH], aB[CH], bB[CH];
    load_chunk<AKC, BKC>(d, m_c, m_ok, n_c, n_ok, h, u0, u1, aA, bA);
    for (int u = u0; u < u1; u += 2 * CH) {
        load_chunk<AKC, BKC>(d, m_c, m_ok, n_c, n_ok, h, u + CH, u1, aB, bB);
        mma_chunk(acc, asum, u, u1, aA, bA);
        load_chunk<AKC, BKC>(d, m_c, m_ok, n_c, n_ok, h, u + 2 * CH, u1, aA, bA);
        mma_chunk(acc, asum, u + CH, u1, aB, bB);
    }

    // ---- combine the four K slices through LDS: red[wave][reg][lane] ----
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    if (d.colsum != nullptr) {
        asum += __shfl_xor(asum, 32, 64);
        if (lane < 32) cs[wave * 32 + lane] = asum;
    }
    __syncthreads();

    // thread (wave, lane) finishes accumulator registers 4*wave .. 4*wave+3:
    // C row = i + 8*wave + 4*h, C col = r  (32x32 f32 C/D map)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = 4 * wave + i;
        float v = red[(0 * 16 + reg) * 64 + lane] + red[(1 * 16 + reg) * 64 + lane] +
                  red[(2 * 16 + reg) * 64 + lane] + red[(3 * 16 + reg) * 64 + lane];
        const int cm = tm * 32 + i + 8 * wave + 4 * h;
        const int cn = tn * 32 + r;
        if (cm < d.M && cn < d.N) {
            if (d.bias != nullptr) v += d.bias[cn];
            switch (d.epi) {
                case EPI_RELU: v = fmaxf(v, 0.f); break;
                case EPI_TANH: v = tanhf(v); break;
                case EPI_MUL_RELU_MASK: v = d.aux[(size_t)cm * d.ldaux + cn] > 0.f ? v : 0.f; break;
                case EPI_MUL_TANH_GRAD: {
                    const float y = d.aux[(size_t)cm * d.ldaux + cn];
                    v *= (1.f - y * y);
                } break;
                default: break;
            }
            d.C[(size_t)cm * d.ldc + cn] = v;
        }
    }
    if (d.colsum != nullptr && tn == 0 && threadIdx.x < 32) {
        const int cm = tm * 32 + threadIdx.x;
        if (cm < d.M)
            d.colsum[cm] = cs[threadIdx.x] + cs[32 + threadIdx.x] + cs[64 + threadIdx.x] + cs[96 + threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void gemm_group_kernel(const GemmGroup g) {
    __shared__ float red[4 * 16 * 64];
    __shared__ float cs[4 * 32];

    // XCD-aware bijective remap: blocks with equal blockIdx % 8 share an XCD
    // under round-robin dispatch; give each such group a contiguous tile range.
    const int total = g.total_tiles;
    const int id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int q = total >> 3, rem = total & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;

    int p = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
        if (i < g.count && tile >= g.p[i].tile_base) p = i;
    const GemmDesc& d = g.p[p];
    const int t = tile - d.tile_base;
    const int tm = t / d.tiles_n, tn = t - tm * d.tiles_n;

    if (d.kind == GEMM_NT)
        gemm_tile<true, true>(d, tm, tn, red, cs);
    else if (d.kind == GEMM_NN)
        gemm_tile<true, false>(d, tm, tn, red, cs);
    else
        gemm_tile<false, false>(d, tm, tn, red, cs);
}

// Host entry: validates shapes (a bad leading dimension would read out of
// bounds on the device), assigns tile ranges and launches on `stream`.
D4PG_API int d4pg_gemm_group(GemmGroup* g, hipStream_t stream) {
    if (g->count < 1 || g->count > MAX_GROUP) return -1;
    int base = 0;
    for (int i = 0; i < g->count; ++i) {
        GemmDesc& d = g->p[i];
        if (d.M < 1 || d.N < 1 || d.K < 1) return -2;
        if (d.A == nullptr || d.B == nullptr || d.C == nullptr) return -3;
        const int kp = (d.K + 3) & ~3;
        if (d.kind == GEMM_NT) {
            if (d.lda < kp || d.ldb < kp || (d.lda & 3) || (d.ldb & 3)) return -4;
            if (((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15)) return -5;
        } else if (d.kind == GEMM_NN) {
            if (d.lda < kp || (d.lda & 3) || d.ldb < d.N) return -4;
            if ((uintptr_t)d.A & 15) return -5;
        } else if (d.kind == GEMM_TN) {
            if (d.lda < d.M || d.ldb < d.N) return -4;
        } else {
            return -6;
        }
        if (d.ldc < d.N) return -4;
        if ((d.epi == EPI_MUL_RELU_MASK || d.epi == EPI_MUL_TANH_GRAD) && (d.aux == nullptr || d.ldaux < d.N)) return -7;
        d.tiles_m = (d.M + 31) / 32;
        d.tiles_n = (d.N + 31) / 32;
        d.tile_base = base;
        base += d.tiles_m * d.tiles_n;
    }
    g->total_tiles = base;
    hipLaunchKernelGGL(gemm_group_kernel, dim3(base), dim3(256), 0, stream, *g);
    return (int)hipGetLastError();
}

D4PG_SIZEOF(GemmDesc)
D4PG_SIZEOF(GemmGroup)

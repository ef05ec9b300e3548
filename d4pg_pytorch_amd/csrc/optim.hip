// Multi-tensor optimiser kernels over the flat parameter buffer (gfx950).
//
//  adam_polyak_kernel : one launch applies Adam to a contiguous parameter range
//                       AND the Polyak average of the matching target range:
//                          m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2
//                          p <- p - (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
//                          t <- (1-tau) t + tau p
//                       (torch.optim.Adam default arithmetic; reference
//                       models/d4pg/d4pg.py:55-56,115-117,126-138 runs ~60 tiny
//                       launches for the same work).  float4 vectorised.
//  polyak_kernel      : stand-alone soft update (used by tests / D2H-free target sync).
//  tick_kernel        : advances the device step counter and bias corrections when the
//                       update is driven without the replay-sampling kernel.
#include "common.h"
#include "mlp32_dev.h"
#include "mlp16_dev.h"

// bf16 "shadow" copies of the weight matrices for the fused bf16 pipeline (csrc/fused.hip).
// The optimiser kernel refreshes them in the same pass that updates the fp32 masters:
//   forward shadow   Wf(n = o, k = i)        rows padded to 16, K padded to 32 (B operand of y = x W^T)
//   backward shadow  Wb(n = i - c0, k = o)   the transpose of input columns [c0, c1) (B operand of dX = dY W)
// both stored in the packed fragment order of common.h (packed_off).  Pads are zero and never written.
// The fused fp32 pipeline (csrc/fused32.hip) keeps the same two copies in fp32, in the fragment order of
// csrc/mlp32_dev.h (packed32_off); ShadowArgs::f32 selects it (f_ld / b_ld = the padded K either way).
struct ShadowMat {
    long long w_off;     // offset (floats) of W inside the range handed to the kernel
    long long f_off;     // forward shadow offset (bf16 elements), -1: none
    long long b_off;     // backward (transposed) shadow offset, -1: none
    int out_dim, in_dim, ld;
    int f_ld, b_ld, b_c0, b_c1;
    int pad_;
};
struct ShadowArgs {
    void* p_shadow;      // shadows of the online parameters (forward + backward): bf16, or fp32 when f32 != 0
    void* t_shadow;      // shadows of the target parameters (forward only), may be null
    ShadowMat m[3];
    int count;           // 0: no shadows (one-launch-per-layer fp32 pipeline)
    int f32;
};

// One wavefront per workgroup: a 190 k-parameter network is then 730 workgroups over the 256 CUs (with 256 threads it is
// 182, fewer than CUs, and the kernel is nothing but memory latency).
constexpr int ADAM_BLOCK = 64;
struct AdamPolyakArgs {
    float* p;          // online parameters   [n]
    const float* g;    // gradients           [n]
    float* m;          // first moment        [n]
    float* v;          // second moment       [n]
    float* t;          // target parameters   [n] (may be null: no Polyak)
    DevState* st;      // device state: bc1 / bc2 of the update in flight
    float* metrics;    // metric accumulator; the finishing launch reduces the losses into it (may be null)
    const float* per_loss;   // [B] per-sample critic loss            (finishing launch only)
    const float* weights;    // [B] importance weights or null
    const float* q_val;      // [B] Q(s, pi(s)); policy loss = -mean
    int B;
    float inv_B;
    long long n;       // multiple of 4
    float lr, beta1, beta2, eps, tau;
    float gscale;      // gradient pre-scale (1/world_size after a sum all-reduce)
    int finish_step;   // 1: this is the last kernel of the update -> publish step_done
    int pad_;
    ShadowArgs sh;
};

// ``t`` is read only where ``has_t``; both by value, so that neither vector is forced into scratch memory for its address.
// 32-bit index arithmetic (a network has far fewer than 2^31 parameters; 64-bit division costs hundreds of instructions).
__device__ __forceinline__ void write_shadows(const ShadowArgs& sh, long long e64, const f32x4 p, const f32x4 t, const bool has_t) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j >= sh.count) break;
        const ShadowMat& m = sh.m[j];
        const long long rel64 = e64 - m.w_off;
        if (rel64 < 0 || rel64 >= (long long)m.out_dim * m.ld) continue;
        const unsigned rel = (unsigned)rel64, ld = (unsigned)m.ld;
        const int o = (int)(rel / ld), i0 = (int)(rel - (unsigned)o * ld);   // 4 elements share row o
        float* const ps32 = (float*)sh.p_shadow;
        float* const ts32 = (sh.t_shadow != nullptr && has_t) ? (float*)sh.t_shadow : nullptr;
        __bf16* const ps16 = (__bf16*)sh.p_shadow;
        __bf16* const ts16 = (sh.t_shadow != nullptr && has_t) ? (__bf16*)sh.t_shadow : nullptr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            if (i >= m.in_dim) continue;
            const bool back = m.b_off >= 0 && i >= m.b_c0 && i < m.b_c1;
            if (sh.f32 == 2) {             // packed16_off (csrc/mlp16_dev.h); f_ld / b_ld = padded K
                if (m.f_off >= 0) {
                    const long long so = m.f_off + packed16_off(o, i, m.f_ld >> 4);
                    ps32[so] = p[k];
                    if (ts32 != nullptr) ts32[so] = t[k];
                }
                if (back) ps32[m.b_off + packed16_off(i - m.b_c0, o, m.b_ld >> 4)] = p[k];
            } else if (sh.f32) {           // packed32_off (csrc/mlp32_dev.h)
                if (m.f_off >= 0) {
                    const long long so = m.f_off + packed32_off(o, i, m.f_ld >> 5);
                    ps32[so] = p[k];
                    if (ts32 != nullptr) ts32[so] = t[k];
                }
                if (back) ps32[m.b_off + packed32_off(i - m.b_c0, o, m.b_ld >> 5)] = p[k];
            } else {                       // bf16 copies (csrc/common.h::packed_off)
                if (m.f_off >= 0) {
                    const long long so = m.f_off + packed_off(o, i, m.f_ld >> 5);
                    ps16[so] = (__bf16)p[k];
                    if (ts16 != nullptr) ts16[so] = (__bf16)t[k];
                }
                if (back) ps16[m.b_off + packed_off(i - m.b_c0, o, m.b_ld >> 5)] = (__bf16)p[k];
            }
        }
        break;
    }
}

__global__ __launch_bounds__(ADAM_BLOCK) void adam_polyak_kernel(const AdamPolyakArgs a) {
    const float bc1 = a.st->bc1, bc2 = a.st->bc2;
    const float step_size = a.lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    const long long n4 = a.n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
        f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
        f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (a.t != nullptr) t = reinterpret_cast<f32x4*>(a.t)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float mk = m[k], vk = v[k], pk = p[k], tk = t[k];
            adam_polyak_math(mk, vk, pk, tk, g[k] * a.gscale, a.beta1, a.beta2, a.eps, a.tau, step_size, inv_sqrt_bc2);
            m[k] = mk, v[k] = vk, p[k] = pk, t[k] = tk;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.t != nullptr) reinterpret_cast<f32x4*>(a.t)[i] = t;
        if (a.sh.count > 0) write_shadows(a.sh, i * 4, p, t, a.t != nullptr);
    }
    // step_done is read only by the first kernel of the next update
    if (a.finish_step && blockIdx.x == 0) {
        if (a.metrics != nullptr) {
            // block 0 folds this update's losses into the running metric sums
            float vl = 0.f, pl = 0.f;
            for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
                vl += a.per_loss[b] * (a.weights ? a.weights[b] : 1.f);
                pl -= a.q_val[b];
            }
            vl = wave_sum(vl);
            pl = wave_sum(pl);
            if (threadIdx.x == 0) {
                a.metrics[MET_VALUE_LOSS] += vl * a.inv_B;
                a.metrics[MET_POLICY_LOSS] += pl * a.inv_B;
                a.metrics[MET_COUNT] += 1.f;
            }
        }
        if (threadIdx.x == 0) a.st->step_done = a.st->step_cur;
    }
}

D4PG_API int d4pg_adam_polyak(const AdamPolyakArgs* a, hipStream_t stream) {
    if (a->n < 4 || (a->n & 3)) return -1;
    if (((uintptr_t)a->p | (uintptr_t)a->g | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->t) & 15) return -2;
    const long long n4 = a->n >> 2;
    int blocks = (int)((n4 + ADAM_BLOCK - 1) / ADAM_BLOCK);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_polyak_kernel, dim3(blocks), dim3(ADAM_BLOCK), 0, stream, *a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ t, const float* __restrict__ p,
                                                     long long n4, float tau) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 tv = reinterpret_cast<f32x4*>(t)[i];
        const f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) tv[k] = (1.f - tau) * tv[k] + tau * pv[k];
        reinterpret_cast<f32x4*>(t)[i] = tv;
    }
}

D4PG_API int d4pg_polyak(float* target, const float* online, long long n, float tau, hipStream_t stream) {
    if (n < 4 || (n & 3) || (((uintptr_t)target | (uintptr_t)online) & 15)) return -1;
    const long long n4 = n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(polyak_kernel, dim3(blocks), dim3(256), 0, stream, target, online, n4, tau);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// The tick itself (devstate_tick) lives in common.h so the sampling kernel can inline it.
__global__ void tick_kernel(DevState* st, float beta1, float beta2, float per_b0, float per_b1,
                            long long total_steps) {
    if (blockIdx.x == 0 && threadIdx.x == 0) devstate_tick(st, beta1, beta2, per_b0, per_b1, total_steps);
}

D4PG_API int d4pg_tick(DevState* st, float beta1, float beta2, float per_b0, float per_b1, long long total_steps,
                       hipStream_t stream) {
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, stream, st, beta1, beta2, per_b0, per_b1, total_steps);
    return (int)hipGetLastError();
}

D4PG_API int d4pg_devstate_size() { return (int)sizeof(DevState); }

D4PG_SIZEOF(ShadowMat)
D4PG_SIZEOF(ShadowArgs)
D4PG_SIZEOF(AdamPolyakArgs)

// Fused loss kernels of the learner update (gfx950, one 64-lane wave per batch row).
//
//  d4pg_critic_loss_kernel : softmax(target logits) -> categorical projection of r + g*z
//                            -> softmax(online logits) -> BCE (reference) or CE loss
//                            -> per-sample loss, PER priority, dlogits, metric.
//  d4pg_actor_loss_kernel  : softmax(logits) . z -> expected Q, policy-loss metric and
//                            dlogits = -(1/B) p (z - q).
//  d3pg_critic_loss_kernel : y = clip(r + (1-d) g Q'), per-sample (Q-y)^2, priority, dQ.
//
// The kernels only write per-sample results; the scalar loss metrics are reduced from
// per_loss / q_out by the optimiser kernel that finishes the update (one same-address
// float atomic per batch row costs more than the whole kernel body).
//
// Math (written from the definitions, see also ops/ref.py):
//   z_j = v_min + j dz;  beta_j = (clip(r + (1-d) g z_j) - v_min)/dz;
//   m_i = sum_j p_j max(0, 1 - |beta_j - i|)            (dense, atomics-free projection)
//   BCE: L = (1/N) sum_i -[m_i max(log q_i,-100) + (1-m_i) max(log(1-q_i),-100)]
//        dL/dlogit_i = q_i (g_i - sum_k g_k q_k),  g_i = (1/N)(-(m_i/q_i)[log q_i>-100] + ((1-m_i)/(1-q_i))[log(1-q_i)>-100])
//   CE : L = -sum_i m_i log q_i;  dL/dlogit_i = q_i sum(m) - m_i
#include "common.h"
#include "loss_dev.h"

constexpr int MAX_ATOM_SLOTS = 4;   // supports up to 256 atoms (4 per lane); kernels are instantiated per slot count

struct D4pgLossArgs {
    const float* logits_t;   // [B, ld] target-critic logits at (s', pi'(s'))
    const float* logits;     // [B, ld] online-critic logits at (s, a)
    const float* reward;     // [B]
    const float* done;       // [B]
    const float* gamma;      // [B] per-sample discount**k (ignored if fixed_gamma >= 0)
    const float* weights;    // [B] importance weights or null
    float* dlogits;          // [B, ld]
    float* per_loss;         // [B]
    float* priority;         // [B]
    float* proj_out;         // [B, ld] optional: projected target (tests), may be null
    float* metrics;          // device metric accumulator
    int B, N, ld;
    int use_ce;
    float v_min, v_max;
    float fixed_gamma;       // < 0: use per-sample gamma
    float inv_B;             // 1 / (global batch) — gradient scale
};

template <int SLOTS>
__global__ __launch_bounds__(256) void d4pg_critic_loss_kernel(const D4pgLossArgs a) {
    __shared__ float s_p[4][256];
    __shared__ float s_beta[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= a.B) return;   // whole wave exits together (no block-level barrier below)
    const int N = a.N;
    const float g = a.fixed_gamma >= 0.f ? a.fixed_gamma : a.gamma[b];
    const float w = a.weights ? a.weights[b] : 1.f;
    float lt[SLOTS], lo[SLOTS], m[SLOTS], dl[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        lt[s] = j < N ? a.logits_t[(size_t)b * a.ld + j] : -INFINITY;
        lo[s] = j < N ? a.logits[(size_t)b * a.ld + j] : -INFINITY;
    }
    const float loss = d4pg_critic_row<SLOTS>(lt, lo, a.reward[b], 1.f - a.done[b], g, N, a.v_min, a.v_max, a.use_ce,
                                              s_p[wave], s_beta[wave], lane, m, dl);
    const float scale = w * a.inv_B;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        if (i < N) {
            if (a.proj_out) a.proj_out[(size_t)b * a.ld + i] = m[s];
            a.dlogits[(size_t)b * a.ld + i] = dl[s] * scale;
        }
    }
    if (lane == 0) {
        a.per_loss[b] = loss;
        a.priority[b] = fabsf(loss) + 1e-4f;
    }
}

D4PG_API int d4pg_critic_loss(const D4pgLossArgs* a, hipStream_t stream) {
    if (a->N < 2 || a->N > 64 * MAX_ATOM_SLOTS || a->ld < a->N || a->B < 1) return -1;
    const dim3 grid((a->B + 3) / 4), block(256);
    switch ((a->N + 63) / 64) {
        case 1: hipLaunchKernelGGL(d4pg_critic_loss_kernel<1>, grid, block, 0, stream, *a); break;
        case 2: hipLaunchKernelGGL(d4pg_critic_loss_kernel<2>, grid, block, 0, stream, *a); break;
        case 3: hipLaunchKernelGGL(d4pg_critic_loss_kernel<3>, grid, block, 0, stream, *a); break;
        default: hipLaunchKernelGGL(d4pg_critic_loss_kernel<4>, grid, block, 0, stream, *a); break;
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
struct D4pgActorLossArgs {
    const float* logits;   // [B, ld] online critic logits at (s, pi(s))
    float* dlogits;        // [B, ld]
    float* q_out;          // [B] expected Q
    float* metrics;
    int B, N, ld;
    float v_min, v_max;
    float inv_B;
};

template <int SLOTS>
__global__ __launch_bounds__(256) void d4pg_actor_loss_kernel(const D4pgActorLossArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= a.B) return;
    const int N = a.N;
    float e[SLOTS], dl[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        e[s] = j < N ? a.logits[(size_t)b * a.ld + j] : -INFINITY;
    }
    const float qv = d4pg_actor_row<SLOTS>(e, N, a.v_min, a.v_max, lane, dl);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        if (j < N) a.dlogits[(size_t)b * a.ld + j] = a.inv_B * dl[s];
    }
    if (lane == 0) a.q_out[b] = qv;
}

D4PG_API int d4pg_actor_loss(const D4pgActorLossArgs* a, hipStream_t stream) {
    if (a->N < 2 || a->N > 64 * MAX_ATOM_SLOTS || a->ld < a->N || a->B < 1) return -1;
    const dim3 grid((a->B + 3) / 4), block(256);
    switch ((a->N + 63) / 64) {
        case 1: hipLaunchKernelGGL(d4pg_actor_loss_kernel<1>, grid, block, 0, stream, *a); break;
        case 2: hipLaunchKernelGGL(d4pg_actor_loss_kernel<2>, grid, block, 0, stream, *a); break;
        case 3: hipLaunchKernelGGL(d4pg_actor_loss_kernel<3>, grid, block, 0, stream, *a); break;
        default: hipLaunchKernelGGL(d4pg_actor_loss_kernel<4>, grid, block, 0, stream, *a); break;
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Scalar-critic (D3PG / DDPG) losses.
struct D3pgLossArgs {
    const float* q_t;       // [B, ldq] target critic output at (s', pi'(s'))
    const float* q;         // [B, ldq] online critic output at (s, a)
    const float* reward;
    const float* done;
    const float* gamma;
    const float* weights;   // may be null
    float* dq;              // [B, ldq] gradient of the critic loss w.r.t. q
    float* per_loss;        // [B]
    float* priority;        // [B]
    float* metrics;
    int B, ldq;
    float fixed_gamma;      // < 0: per-sample
    float min_value, max_value;
    float inv_B;
};

__global__ __launch_bounds__(256) void d3pg_critic_loss_kernel(const D3pgLossArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float contrib = 0.f;
    if (b < a.B) {
        const float g = a.fixed_gamma >= 0.f ? a.fixed_gamma : a.gamma[b];
        float y = a.reward[b] + (1.f - a.done[b]) * g * a.q_t[(size_t)b * a.ldq];
        y = fminf(a.max_value, fmaxf(a.min_value, y));
        const float diff = a.q[(size_t)b * a.ldq] - y;
        const float w = a.weights ? a.weights[b] : 1.f;
        const float l = diff * diff;
        a.per_loss[b] = l;
        a.priority[b] = fabsf(l) + 1e-4f;
        a.dq[(size_t)b * a.ldq] = 2.f * diff * w * a.inv_B;
        contrib = l * w * a.inv_B;
    }
    (void)contrib;
}

D4PG_API int d3pg_critic_loss(const D3pgLossArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->ldq < 1) return -1;
    hipLaunchKernelGGL(d3pg_critic_loss_kernel, dim3((a->B + 255) / 256), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// Actor loss of the scalar critic: L = -mean Q(s, pi(s))  =>  dQ = -1/B, metric -= Q/B.
struct D3pgActorLossArgs {
    const float* q;   // [B, ldq]
    float* dq;        // [B, ldq]
    float* q_out;     // [B] copy of Q for the metric reduction
    int B, ldq;
    float inv_B;
};

__global__ __launch_bounds__(256) void d3pg_actor_loss_kernel(const D3pgActorLossArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float contrib = 0.f;
    if (b < a.B) {
        a.dq[(size_t)b * a.ldq] = -a.inv_B;
        a.q_out[b] = a.q[(size_t)b * a.ldq];
    }
    (void)contrib;
}

D4PG_API int d3pg_actor_loss(const D3pgActorLossArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->ldq < 1) return -1;
    hipLaunchKernelGGL(d3pg_actor_loss_kernel, dim3((a->B + 255) / 256), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

D4PG_SIZEOF(D4pgLossArgs)
D4PG_SIZEOF(D4pgActorLossArgs)
D4PG_SIZEOF(D3pgLossArgs)
D4PG_SIZEOF(D3pgActorLossArgs)

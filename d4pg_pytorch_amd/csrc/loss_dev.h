// Per-row loss math shared by the stand-alone loss kernels (loss.hip) and the fused
// bf16 phase kernels (fused.hip).  One 64-lane wave owns one batch row; lane l holds
// atoms l, l+64, ... (SLOTS of them).
#pragma once
#include "common.h"

// Single-instruction transcendental forms (v_rcp_f32 / v_log_f32, 1 ulp): the per-row loss
// runs on one wave and is bound by instruction issue, where the IEEE-exact library
// expansions of division and log cost 10-60 instructions each.
// FAST = false keeps the IEEE-exact forms (the fp32 pipeline tracks eager PyTorch to 2e-5).
// Ordering point between LDS accesses of different lanes of ONE wave to the wave's private
// scratch row.  The hardware executes a wave's LDS instructions in program order, and the
// compiler keeps may-aliasing accesses in program order, so nothing needs to be emitted;
// the exact build keeps explicit wave-level fences, the fast build leaves the scheduler
// free to interleave the independent rows a wave processes.
template <bool FAST>
__device__ __forceinline__ void lds_order() {
    if constexpr (!FAST) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
template <bool FAST>
__device__ __forceinline__ float rcp_(float x) {
    if constexpr (FAST) return __builtin_amdgcn_rcpf(x);
    else return 1.f / x;
}
template <bool FAST>
__device__ __forceinline__ float log_(float x) {
    if constexpr (FAST) return __logf(x);
    else return logf(x);
}
template <bool FAST>
__device__ __forceinline__ float log1m_(float q) {     // log(1 - q)
    if constexpr (FAST) return __logf(1.f - q);
    else return log1pf(-q);
}

// Categorical projection + BCE/CE loss + logit gradient for one row.
//   lt, lo : raw target / online logits of this lane's atoms (-inf beyond N)
//   sp     : this wave's LDS scratch, >= N floats (sb is unused, kept for the call sites)
// Outputs: m (projected target), dl (d loss / d logit, unscaled by w/B), returns the row loss.
template <int SLOTS, bool FAST = false>
__device__ __forceinline__ float d4pg_critic_row(float (&lt)[SLOTS], float (&lo)[SLOTS], float r, float nd, float g,
                                                 int N, float v_min, float v_max, int use_ce, float* sp, float* sb,
                                                 int lane, float (&m)[SLOTS], float (&dl)[SLOTS]) {
    const float dz = (v_max - v_min) / (float)(N - 1);
    const float inv_dz = (float)(N - 1) / (v_max - v_min);
    float mx_t = -INFINITY, mx_o = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        mx_t = fmaxf(mx_t, lt[s]);
        mx_o = fmaxf(mx_o, lo[s]);
    }
    mx_t = wave_max(mx_t);
    mx_o = wave_max(mx_o);
    float sum_t = 0.f, sum_o = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        lt[s] = j < N ? __expf(lt[s] - mx_t) : 0.f;
        lo[s] = j < N ? __expf(lo[s] - mx_o) : 0.f;
        sum_t += lt[s];
        sum_o += lo[s];
    }
    sum_t = wave_sum(sum_t);
    sum_o = wave_sum(sum_o);
    const float inv_t = rcp_<FAST>(sum_t), inv_o = rcp_<FAST>(sum_o);
    const float log_sum_o = log_<FAST>(sum_o);
    // Projection as a scatter into this wave's LDS row: source atom j adds p_j (u - b_j) to bin
    // l = floor(b_j) and p_j (b_j - l) to bin l + 1 (zero weight when b_j is an integer, so the
    // clamp of l + 1 at the last bin is harmless).  LDS float adds of one wave execute in a
    // fixed lane order, so the result is reproducible; rows are private to the wave, the
    // wave-level fences only keep the compiler from reordering the LDS accesses.
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        if (j < N) sp[j] = 0.f;
    }
    lds_order<FAST>();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        if (j < N) {
            const float z = v_min + dz * (float)j;
            float tz = r + nd * g * z;
            tz = fminf(v_max, fmaxf(v_min, tz));
            const float bj = FAST ? (tz - v_min) * inv_dz : (tz - v_min) / dz;
            const float fl = floorf(bj);
            const float fr = bj - fl;
            int li = (int)fl;
            li = li < 0 ? 0 : (li > N - 1 ? N - 1 : li);
            const int ui = li + 1 > N - 1 ? N - 1 : li + 1;
            const float pj = lt[s] * inv_t;
            __hip_atomic_fetch_add(&sp[li], pj * (1.f - fr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&sp[ui], pj * fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    lds_order<FAST>();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        m[s] = i < N ? sp[i] : 0.f;
    }
    lds_order<FAST>();

    float q[SLOTS], gi[SLOTS];
    float loss = 0.f, gq = 0.f, msum = 0.f;
    const float inv_n = 1.f / (float)N;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        q[s] = lo[s] * inv_o;
        gi[s] = 0.f;
        if (i < N) {
            if (use_ce) {
                // log q from the log-sum-exp form (finite even when q underflows)
                const float logq = log_<FAST>(lo[s] > 0.f ? lo[s] : (FAST ? 1e-37f : 1e-45f)) - log_sum_o;
                loss -= m[s] * logq;
                msum += m[s];
            } else {
                const float lq = log_<FAST>(q[s]);
                const float l1 = log1m_<FAST>(q[s]);
                const bool live0 = lq > -100.f, live1 = l1 > -100.f;
                loss -= m[s] * fmaxf(lq, -100.f) + (1.f - m[s]) * fmaxf(l1, -100.f);
                float gg = 0.f;
                if (live0) gg -= FAST ? m[s] * rcp_<true>(q[s]) : m[s] / q[s];
                if (live1) gg += FAST ? (1.f - m[s]) * rcp_<true>(1.f - q[s]) : (1.f - m[s]) / (1.f - q[s]);
                gi[s] = FAST ? gg * inv_n : gg / (float)N;
                gq += gi[s] * q[s];
            }
        }
    }
    loss = wave_sum(loss);
    if (use_ce) {
        msum = wave_sum(msum);
    } else {
        loss = FAST ? loss * inv_n : loss / (float)N;
        gq = wave_sum(gq);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) dl[s] = use_ce ? (q[s] * msum - m[s]) : q[s] * (gi[s] - gq);
    return loss;
}

// Expected value of the categorical distribution and the gradient of -Q w.r.t. the logits.
//   e : raw logits of this lane's atoms (-inf beyond N); on return dl = -p (z - q)
template <int SLOTS, bool FAST = false>
__device__ __forceinline__ float d4pg_actor_row(float (&e)[SLOTS], int N, float v_min, float v_max, int lane,
                                                float (&dl)[SLOTS]) {
    const float dz = (v_max - v_min) / (float)(N - 1);
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) mx = fmaxf(mx, e[s]);
    mx = wave_max(mx);
    float sum = 0.f, qz = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        e[s] = j < N ? __expf(e[s] - mx) : 0.f;
        sum += e[s];
        qz += e[s] * (v_min + dz * (float)j);
    }
    sum = wave_sum(sum);
    qz = wave_sum(qz);
    const float inv = rcp_<FAST>(sum);
    const float qv = qz * inv;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const float z = v_min + dz * (float)(lane + 64 * s);
        dl[s] = -(e[s] * inv) * (z - qv);
    }
    return qv;
}

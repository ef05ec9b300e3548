// Per-row loss math shared by the stand-alone loss kernels (loss.hip) and the fused
// bf16 phase kernels (fused.hip).  One 64-lane wave owns one batch row; lane l holds
// atoms l, l+64, ... (SLOTS of them).
#pragma once
#include "common.h"

// Categorical projection + BCE/CE loss + logit gradient for one row.
//   lt, lo : raw target / online logits of this lane's atoms (-inf beyond N)
//   sp, sb : this wave's LDS scratch, >= N floats each
// Outputs: m (projected target), dl (d loss / d logit, unscaled by w/B), returns the row loss.
template <int SLOTS>
__device__ __forceinline__ float d4pg_critic_row(float (&lt)[SLOTS], float (&lo)[SLOTS], float r, float nd, float g,
                                                 int N, float v_min, float v_max, int use_ce, float* sp, float* sb,
                                                 int lane, float (&m)[SLOTS], float (&dl)[SLOTS]) {
    const float dz = (v_max - v_min) / (float)(N - 1);
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
    const float inv_t = 1.f / sum_t, inv_o = 1.f / sum_o;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = lane + 64 * s;
        if (j < N) {
            const float z = v_min + dz * (float)j;
            float tz = r + nd * g * z;
            tz = fminf(v_max, fmaxf(v_min, tz));
            sp[j] = lt[s] * inv_t;
            sb[j] = (tz - v_min) / dz;
        }
    }
    // LDS rows are private to this wave; wave-level execution order suffices,
    // the fences keep the compiler from reordering the LDS accesses.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // dense projection: lane owns destination atoms i = lane + 64 s
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) m[s] = 0.f;
#pragma unroll 8
    for (int j = 0; j < N; ++j) {
        const float pj = sp[j], bj = sb[j];   // LDS broadcast reads
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const float i = (float)(lane + 64 * s);
            m[s] += pj * fmaxf(0.f, 1.f - fabsf(bj - i));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    float q[SLOTS], gi[SLOTS];
    float loss = 0.f, gq = 0.f, msum = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = lane + 64 * s;
        q[s] = lo[s] * inv_o;
        gi[s] = 0.f;
        if (i < N) {
            if (use_ce) {
                // log q from the log-sum-exp form (finite even when q underflows)
                const float logq = logf(lo[s] > 0.f ? lo[s] : 1e-45f) - logf(sum_o);
                loss -= m[s] * logq;
                msum += m[s];
            } else {
                const float lq = logf(q[s]);
                const float l1 = log1pf(-q[s]);
                const bool live0 = lq > -100.f, live1 = l1 > -100.f;
                loss -= m[s] * fmaxf(lq, -100.f) + (1.f - m[s]) * fmaxf(l1, -100.f);
                float gg = 0.f;
                if (live0) gg -= m[s] / q[s];
                if (live1) gg += (1.f - m[s]) / (1.f - q[s]);
                gi[s] = gg / (float)N;
                gq += gi[s] * q[s];
            }
        }
    }
    loss = wave_sum(loss);
    if (use_ce) {
        msum = wave_sum(msum);
    } else {
        loss /= (float)N;
        gq = wave_sum(gq);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) dl[s] = use_ce ? (q[s] * msum - m[s]) : q[s] * (gi[s] - gq);
    return loss;
}

// Expected value of the categorical distribution and the gradient of -Q w.r.t. the logits.
//   e : raw logits of this lane's atoms (-inf beyond N); on return dl = -p (z - q)
template <int SLOTS>
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
    const float inv = 1.f / sum;
    const float qv = qz * inv;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const float z = v_min + dz * (float)(lane + 64 * s);
        dl[s] = -(e[s] * inv) * (z - qv);
    }
    return qv;
}

// The ant of envs/spatial.py resident on the GPU (gfx950), one thread per environment.  The simulator itself is
// csrc/ant_sim.h (shared with the host runtime); this file adds the episode bookkeeping of a device environment:
// Philox resets, time limits, episode statistics, observations for the policy and for the replay rows.
#include "common.h"
#include "ant_sim.h"

struct AntEnvArgs {
    const AntParams* P;      // device memory
    float* state;            // [E][32] position 3, quaternion 4 (w, x, y, z), hinge angles 8, velocity u 14 (v, w, q')
    int* t;                  // [E]
    float* ep_return;        // [E]
    const float* action;     // [E][ld_act]
    float* obs;              // [E][111] what the policy sees next (a fresh state where an episode ended)
    float* next_obs;         // [E][111] the observation that followed the action
    float* reward;           // [E]
    float* done;             // [E]
    float* trunc;            // [E]
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, ld_act, time_limit, init;
    int limit_terminal;
    float reward_scale;
};

namespace ant {
__device__ __forceinline__ float unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(64) void ant_step_kernel(const AntEnvArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    const AntParams* P = a.P;      // (not restrict-qualified: csrc/ant_sim.h)
    float* st = a.state + (size_t)e * STRIDE;
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
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        float rew;
        bool term;
        advance(P, pos, quat, qj, u, a.action + (size_t)e * a.ld_act, wrench, rew, term);
        t += 1;
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        float* no = a.next_obs + (size_t)e * S;
        observe(pos, quat, qj, u, wrench, no);
        a.reward[e] = rew / a.reward_scale;
        a.done[e] = (term || (limit && a.limit_terminal)) ? 1.f : 0.f;
        a.trunc[e] = (limit && !term && !a.limit_terminal) ? 1.f : 0.f;
        const float ret = a.ep_return[e] + rew;
        reset = term || limit;
        if (reset) {
            atomicAdd(&a.stats[0], (double)ret);
            atomicAdd(&a.stats[1], 1.0);
        }
        a.ep_return[e] = reset ? 0.f : ret;
    }
    if (reset) {
        float rnd[32];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            uint32_t r[4];
            philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0xA27u + blk, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
            for (int k = 0; k < 4; ++k) rnd[4 * blk + k] = unit(r[k]);
        }
        const float s = P->reset_noise;
#pragma unroll
        for (int i = 0; i < 3; ++i) pos[i] = (2.f * rnd[i] - 1.f) * s;
        pos[2] += P->z_init;
        float q0 = 1.f + (2.f * rnd[3] - 1.f) * s, q1 = (2.f * rnd[4] - 1.f) * s, q2 = (2.f * rnd[5] - 1.f) * s, q3 = (2.f * rnd[6] - 1.f) * s;
        const float inv = 1.f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        quat[0] = q0 * inv, quat[1] = q1 * inv, quat[2] = q2 * inv, quat[3] = q3 * inv;
#pragma unroll
        for (int i = 0; i < 8; ++i) qj[i] = (2.f * rnd[7 + i] - 1.f) * s + ((i & 1) ? P->ankle0[i >> 1] : 0.f);
        // 14 normal deviates x reset_noise (Box-Muller on 7 pairs)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float mag = s * sqrtf(-2.f * logf(fmaxf(rnd[15 + 2 * i], 1e-7f))), ph = 6.2831853f * rnd[16 + 2 * i];
            u[2 * i] = mag * cosf(ph);
            u[2 * i + 1] = mag * sinf(ph);
        }
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) st[i] = pos[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) st[3 + i] = quat[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[7 + i] = qj[i];
#pragma unroll
    for (int i = 0; i < N; ++i) st[15 + i] = u[i];
    a.t[e] = t;
    observe(pos, quat, qj, u, wrench, a.obs + (size_t)e * S);
}

__global__ void counter_kernel(unsigned long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1ull;
}
}  // namespace ant

// `host_params`: the host copy of *a->P (the topology check must not read device memory)
D4PG_API int d4pg_ant_env_step(const AntEnvArgs* a, const AntParams* host_params, hipStream_t stream) {
    if (a->E < 1 || a->P == nullptr || host_params == nullptr || a->state == nullptr || a->t == nullptr || a->obs == nullptr ||
        a->counter == nullptr || a->ep_return == nullptr || !(a->reward_scale > 0.f))
        return -1;
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < 8))
        return -2;
    if (host_params->n_links != ant::L || host_params->n_contacts != ant::C) return -3;
    for (int i = 0; i < ant::L; ++i)
        if (host_params->parent[i] != ant::parent_of(i)) return -3;       // not the mechanism this kernel was compiled for
    for (int c = 0; c < ant::C; ++c)
        if (host_params->c_link[c] != ant::contact_link(c)) return -3;
    if (host_params->substeps < 1 || host_params->substeps > 64 || !(host_params->dt > 0.f)) return -4;
    hipLaunchKernelGGL(ant::ant_step_kernel, dim3((a->E + 63) / 64), dim3(64), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(ant::counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}
D4PG_SIZEOF(AntParams)
D4PG_SIZEOF(AntEnvArgs)

// Planar articulated-body environments resident on the GPU (gfx950): Hopper, Walker2d, HalfCheetah and
// BipedalWalker of envs/planar.py, one thread per environment.  The simulator itself is csrc/planar_sim.h (shared
// with the host runtime); this file adds the episode bookkeeping of a device environment: Philox resets,
// time limits, episode statistics, observations for the policy and for the replay rows.
#include "common.h"
#include "planar_sim.h"

struct PlanarEnvArgs {
    const PlanarParams* P;   // device memory
    float* state;            // [E][24] q[n], qd[n] at 0 and 9; leg-contact flags at 18, 19; previous shaping at 20
    int* t;                  // [E]
    float* ep_return;        // [E]
    const float* action;     // [E][ld_act]
    float* obs;              // [E][S] what the policy sees next (a fresh state where an episode ended)
    float* next_obs;         // [E][S] the observation that followed the action
    float* reward;           // [E]
    float* done;             // [E]
    float* trunc;            // [E]
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    float* terrain;          // [200][E] ground heights of the walker (sample-major: a wavefront's 64 environments read and
                             // write one sample as one 256-byte access), redrawn at every reset; null: flat ground
    unsigned long long seed;
    int E, ld_act, time_limit, init;
    int limit_terminal, kind;        // kind: 0 Hopper, 1 Walker2d, 2 HalfCheetah, 3 BipedalWalker
    float reward_scale;
    int hilly;               // walker: 1 = the gym task's grass (random walk of the slope), 0 = all heights zero
};

namespace planar {
__device__ __forceinline__ float unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

template <class T>
__global__ __launch_bounds__(64) void planar_step_kernel(const PlanarEnvArgs a) {
    constexpr int L = T::L, n = L + 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    const PlanarParams* P = a.P;
    float* st = a.state + (size_t)e * STRIDE;
    float q[n], qd[n];
#pragma unroll
    for (int i = 0; i < n; ++i) q[i] = st[i], qd[i] = st[QD0 + i];
    float c0 = st[FLAG0], c1 = st[FLAG0 + 1], prev = st[PREV];
    int t = a.t[e];
    bool reset = a.init != 0;
    float* ter = (T::TASK == 1 && a.terrain != nullptr) ? a.terrain + e : nullptr;       // sample j at ter[j * E]
    if (!a.init) {
        float rew;
        bool term;
        advance<T>(P, q, qd, a.action + (size_t)e * a.ld_act, c0, c1, prev, rew, term, ter, a.E);
        t += 1;
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        float o[T::S];
        observe<T>(q, qd, c0, c1, o, ter, a.E);
#pragma unroll
        for (int i = 0; i < T::S; ++i) a.next_obs[(size_t)e * T::S + i] = o[i];
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
        if (T::TASK == 0) {
            const float s = P->reset_noise;
#pragma unroll
            for (int blk = 0; blk < (2 * n + 3) / 4; ++blk) {
                uint32_t r[4];
                philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x9A17u + blk, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = blk * 4 + k;
                    const float u = (2.f * unit(r[k]) - 1.f) * s;
                    if (i < n) q[i] = u;
                    else if (i < 2 * n) qd[i - n] = u;
                }
            }
            q[1] += P->q0_y;
        } else {
            using namespace biped;
            uint32_t r[4];
            philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x9A17u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
            for (int i = 0; i < n; ++i) q[i] = 0.f, qd[i] = 0.f;
            q[0] = START_X;
            q[1] = 2.f * LEG_H - LEG_DOWN + 0.02f;
            q[3] = 0.05f, q[5] = -0.05f, q[4] = -0.1f, q[6] = -0.1f;
            qd[0] = (2.f * unit(r[0]) - 1.f) * 5.f / HULL_MASS / FPS * 10.f;
            prev = 130.f * q[0] / SCALE;
            if (ter != nullptr) {        // fresh ground: a flat start pad, then a damped random walk of the slope
                uint32_t x = r[1] | 1u;      // xorshift32 seeded by the reset's Philox draw: 200 cheap uniforms
                float y = 0.f, velocity = 0.f;
                for (int i = 0; i < TERRAIN_LENGTH; ++i) {
                    x ^= x << 13, x ^= x >> 17, x ^= x << 5;
                    velocity = 0.8f * velocity + 0.01f * (y < 0.f ? 1.f : (y > 0.f ? -1.f : 0.f));
                    if (i > TERRAIN_STARTPAD && a.hilly) velocity += (2.f * unit(x) - 1.f) / SCALE;
                    y += velocity;
                    ter[(size_t)i * a.E] = y;
                }
            }
        }
        c0 = c1 = 0.f;
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < n; ++i) st[i] = q[i], st[QD0 + i] = qd[i];
    st[FLAG0] = c0, st[FLAG0 + 1] = c1, st[PREV] = prev;
    a.t[e] = t;
    if (!reset) {                    // nothing changed since next_obs was written: the policy sees the same thing
#pragma unroll
        for (int i = 0; i < T::S; ++i) a.obs[(size_t)e * T::S + i] = a.next_obs[(size_t)e * T::S + i];
        return;
    }
    float o[T::S];
    observe<T>(q, qd, c0, c1, o, ter, a.E);
#pragma unroll
    for (int i = 0; i < T::S; ++i) a.obs[(size_t)e * T::S + i] = o[i];
}

__global__ void counter_kernel(unsigned long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1ull;
}

template <class T>
int launch(const PlanarEnvArgs* a, const PlanarParams* host, hipStream_t stream) {
    if (host->n_links != T::L || host->n_contacts != T::C || host->n_act != T::A || host->parents != T::PAR ||
        host->contact_links != T::CL)
        return -3;       // the host model is not the mechanism this kernel was compiled for
    hipLaunchKernelGGL(planar_step_kernel<T>, dim3((a->E + 63) / 64), dim3(64), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}
}  // namespace planar

// `host_params`: the host copy of *a->P (the topology check must not read device memory)
D4PG_API int d4pg_planar_env_step(const PlanarEnvArgs* a, const PlanarParams* host_params, hipStream_t stream) {
    if (a->E < 1 || a->P == nullptr || host_params == nullptr || a->state == nullptr || a->t == nullptr || a->obs == nullptr ||
        a->counter == nullptr || a->ep_return == nullptr || !(a->reward_scale > 0.f))
        return -1;
    if (host_params->substeps < 1 || host_params->substeps > 64 || !(host_params->dt > 0.f)) return -4;
    const int need_act = a->kind == 0 ? 3 : (a->kind == 3 ? 4 : 6);
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < need_act))
        return -2;
    switch (a->kind) {
        case 0: return planar::launch<planar::HopperT>(a, host_params, stream);
        case 1: return planar::launch<planar::WalkerT>(a, host_params, stream);
        case 2: return planar::launch<planar::CheetahT>(a, host_params, stream);
        case 3: return planar::launch<planar::BipedT>(a, host_params, stream);
        default: return -5;
    }
}
D4PG_SIZEOF(PlanarParams)
D4PG_SIZEOF(PlanarEnvArgs)

// Device-resident rollouts and weight hand-off between processes on one GPU (gfx950).
//
//  pack_actor32_kernel   : flat actor parameters (the learner's fp32 master layout) -> the packed fp32
//                          copies the inference kernel streams (csrc/mlp32_dev.h), device to device;
//                          the source may be another process's memory opened through HIP IPC
//  pendulum_step_kernel  : E inverted pendulums (the dynamics of gym's Pendulum-v0, envs/pendulum.py)
//                          advanced one step on the device, with time-limit resets from a Philox stream
//  rollout_rows_kernel   : a [T, E] rollout segment + its n-step targets (csrc/nstep.hip) -> replay rows
//                          [s | a | R | done | gamma^k | s'] ready for replay_insert, no host hop
//  d4pg_ipc_*            : hipIpc memory handles, so that an agent process reads the learner's weights in
//                          place (what torch's share_memory() gives the reference on CUDA,
//                          models/d4pg/engine.py:129-144)
#include <vector>
#include <string.h>

#include "common.h"
#include "mlp32_dev.h"
#include "mlp16_dev.h"

struct PackMat32 {
    long long src_off;     // W[out][ld] inside the flat source (floats)
    long long bias_off;    // bias[out] inside the flat source
    long long dst_off;     // packed copy inside dst (floats)
    long long dst_bias;    // bias copy inside dst
    int out_dim, in_dim, ld, ks;
};
struct PackActorArgs32 {
    const float* src;
    float* dst;
    PackMat32 m[3];
    int wide;              // packed copies in the order of csrc/mlp16_dev.h (ks = 16-wide k-groups)
    int pad_;
};

__global__ __launch_bounds__(256) void pack_actor32_kernel(const PackActorArgs32 a) {
    const PackMat32& m = a.m[blockIdx.y];
    const long long n = (long long)m.out_dim * m.in_dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(i / m.in_dim), k = (int)(i - (long long)o * m.in_dim);
        a.dst[m.dst_off + (a.wide ? packed16_off(o, k, m.ks) : packed32_off(o, k, m.ks))] = a.src[m.src_off + (long long)o * m.ld + k];
        if (k == 0) a.dst[m.dst_bias + o] = a.src[m.bias_off + o];
    }
}
D4PG_API int d4pg_pack_actor32(const PackActorArgs32* a, hipStream_t stream) {
    if (a->src == nullptr || a->dst == nullptr) return -1;
    for (int i = 0; i < 3; ++i)
        if (a->m[i].out_dim < 1 || a->m[i].in_dim < 1 || a->m[i].ld < a->m[i].in_dim || a->m[i].ks * (a->wide ? 16 : 32) < a->m[i].in_dim) return -2;
    hipLaunchKernelGGL(pack_actor32_kernel, dim3(160, 3), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
struct PendulumArgs {
    float* theta;            // [E] state, updated in place
    float* theta_dot;        // [E]
    int* t;                  // [E] steps into the episode
    float* ep_return;        // [E] running (unscaled) episode return
    const float* action;     // [E][ld_act] torque in column 0
    float* obs;              // [E][3]  observation the policy sees NEXT (after a reset where one happened)
    float* next_obs;         // [E][3]  observation that followed the action (the transition's s')
    float* reward;           // [E]     scaled reward (reward / reward_scale)
    float* done;             // [E]     terminal flag (a pendulum never terminates: 0)
    float* trunc;            // [E]     1 where the time limit ended the episode at this step (limit_terminal == 0)
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;   // device counter: one increment per call (Philox stream position)
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, ld_act, time_limit, init;     // init: 1 = reset every environment instead of stepping
    float reward_scale;
    int limit_terminal;      // 1: the time limit is reported as done (gym's old 4-tuple API, env/env_wrapper.py:17-19)
};

__global__ __launch_bounds__(256) void pendulum_step_kernel(const PendulumArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    const float max_speed = 8.f, max_torque = 2.f, dt = 0.05f, g = 10.f, m = 1.f, l = 1.f, pi = 3.14159265358979f;
    float th = a.theta[e], thd = a.theta_dot[e];
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        const float u = fminf(max_torque, fmaxf(-max_torque, a.action[(size_t)e * a.ld_act]));
        float ang = fmodf(th + pi, 2.f * pi);
        if (ang < 0.f) ang += 2.f * pi;
        ang -= pi;
        const float cost = ang * ang + 0.1f * thd * thd + 0.001f * u * u;
        const float nthd = thd + (-3.f * g / (2.f * l) * sinf(th + pi) + 3.f / (m * l * l) * u) * dt;
        th = th + nthd * dt;
        thd = fminf(max_speed, fmaxf(-max_speed, nthd));
        t += 1;
        a.next_obs[e * 3 + 0] = cosf(th);
        a.next_obs[e * 3 + 1] = sinf(th);
        a.next_obs[e * 3 + 2] = thd;
        a.reward[e] = -cost / a.reward_scale;
        const float ret = a.ep_return[e] - cost;
        reset = a.time_limit > 0 && t >= a.time_limit;
        a.done[e] = reset && a.limit_terminal ? 1.f : 0.f;
        a.trunc[e] = reset && !a.limit_terminal ? 1.f : 0.f;
        if (reset) {
            atomicAdd(&a.stats[0], (double)ret);
            atomicAdd(&a.stats[1], 1.0);
        }
        a.ep_return[e] = reset ? 0.f : ret;
    }
    if (reset) {
        uint32_t r[4];
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x9E3779B9u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        th = ((float)(r[0] >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * pi;
        thd = (float)(r[1] >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f;
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
    a.theta[e] = th;
    a.theta_dot[e] = thd;
    a.t[e] = t;
    a.obs[e * 3 + 0] = cosf(th);
    a.obs[e * 3 + 1] = sinf(th);
    a.obs[e * 3 + 2] = thd;
}
__global__ void rollout_counter_kernel(unsigned long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}
// Paired counter slots: a sequence of an odd number of steps ends in the other slot; this brings it back, so that a
// captured sequence always starts from the slot it was captured with.
__global__ void rollout_counter_copy_kernel(unsigned long long* dst, const unsigned long long* src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src;
}
// Kernel nodes of a captured graph: what a replay launches (tests and benches count launches per environment step).
D4PG_API int d4pg_graph_kernel_nodes(hipGraph_t g) {
    size_t n = 0;
    if (g == nullptr || hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return -1;
    std::vector<hipGraphNode_t> nodes(n);
    if (n > 0 && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return -1;
    int k = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
    }
    return k;
}
D4PG_API int d4pg_counter_copy(unsigned long long* dst, const unsigned long long* src, hipStream_t stream) {
    if (dst == nullptr || src == nullptr || dst == src) return -1;
    hipLaunchKernelGGL(rollout_counter_copy_kernel, dim3(1), dim3(64), 0, stream, dst, src);
    return (int)hipGetLastError();
}
D4PG_API int d4pg_pendulum_step(const PendulumArgs* a, hipStream_t stream) {
    if (a->E < 1 || a->theta == nullptr || a->obs == nullptr || a->counter == nullptr) return -1;
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < 1))
        return -2;
    hipLaunchKernelGGL(pendulum_step_kernel, dim3((a->E + 255) / 256), dim3(256), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(rollout_counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Continuous-force cart-pole balancing (envs/synthetic.py::CartPoleContinuousVec, the stand-in with the shape of
// InvertedPendulum-v2): +1 per step, TERMINAL when the pole or the cart leaves its range, time limit on top.
struct CartPoleArgs {
    float* state;            // [E][4] x, x_dot, theta, theta_dot, updated in place
    int* t;                  // [E]
    float* ep_return;        // [E]
    const float* action;     // [E][ld_act] force in column 0, clipped to [-1, 1]
    float* obs;              // [E][4] what the policy sees next (a fresh state where an episode ended)
    float* next_obs;         // [E][4] the state that followed the action
    float* reward;           // [E]
    float* done;             // [E] 1: fell (and the time limit, with limit_terminal)
    float* trunc;            // [E] 1: time limit without limit_terminal
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, ld_act, time_limit, init;
    int limit_terminal;
    int pad_;
};
__global__ __launch_bounds__(256) void cartpole_step_kernel(const CartPoleArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    const float gravity = 9.8f, masscart = 1.f, masspole = 0.1f, length = 0.5f, force_mag = 10.f, tau = 0.02f;
    float x = a.state[e * 4 + 0], xd = a.state[e * 4 + 1], th = a.state[e * 4 + 2], thd = a.state[e * 4 + 3];
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        const float force = fminf(1.f, fmaxf(-1.f, a.action[(size_t)e * a.ld_act])) * force_mag;
        const float total = masscart + masspole, pml = masspole * length;
        const float c = cosf(th), sn = sinf(th);
        const float temp = (force + pml * thd * thd * sn) / total;
        const float th_acc = (gravity * sn - c * temp) / (length * (4.f / 3.f - masspole * c * c / total));
        const float x_acc = temp - pml * th_acc * c / total;
        x += tau * xd;
        xd += tau * x_acc;
        th += tau * thd;
        thd += tau * th_acc;
        t += 1;
        a.next_obs[e * 4 + 0] = x;
        a.next_obs[e * 4 + 1] = xd;
        a.next_obs[e * 4 + 2] = th;
        a.next_obs[e * 4 + 3] = thd;
        a.reward[e] = 1.f;
        const bool fell = fabsf(th) > 0.2f || fabsf(x) > 2.4f;
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        a.done[e] = (fell || (limit && a.limit_terminal)) ? 1.f : 0.f;
        a.trunc[e] = (limit && !fell && !a.limit_terminal) ? 1.f : 0.f;
        const float ret = a.ep_return[e] + 1.f;
        reset = fell || limit;
        if (reset) {
            atomicAdd(&a.stats[0], (double)ret);
            atomicAdd(&a.stats[1], 1.0);
        }
        a.ep_return[e] = reset ? 0.f : ret;
    }
    if (reset) {
        uint32_t r[4];
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0xC0FFEEu, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        x = ((float)(r[0] >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
        xd = ((float)(r[1] >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
        th = ((float)(r[2] >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
        thd = ((float)(r[3] >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
    a.state[e * 4 + 0] = x;
    a.state[e * 4 + 1] = xd;
    a.state[e * 4 + 2] = th;
    a.state[e * 4 + 3] = thd;
    a.t[e] = t;
    a.obs[e * 4 + 0] = x;
    a.obs[e * 4 + 1] = xd;
    a.obs[e * 4 + 2] = th;
    a.obs[e * 4 + 3] = thd;
}
D4PG_API int d4pg_cartpole_step(const CartPoleArgs* a, hipStream_t stream) {
    if (a->E < 1 || a->state == nullptr || a->t == nullptr || a->obs == nullptr || a->counter == nullptr) return -1;
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < 1))
        return -2;
    hipLaunchKernelGGL(cartpole_step_kernel, dim3((a->E + 255) / 256), dim3(256), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(rollout_counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Lunar lander with continuous engines (envs/lunar_lander.py::LunarLanderVec, the task of LunarLanderContinuous-v2
// as a planar rigid body with spring-damper feet): the same arithmetic in fp32, one thread per environment.
// Constants are those of the host class; change them together.
struct LanderArgs {
    float* state;            // [E][12] x, y, vx, vy, theta, omega, contact0, contact1, previous shaping, rest steps, 0, 0
    int* t;                  // [E]
    float* ep_return;        // [E] running UNSCALED episode return
    const float* action;     // [E][ld_act] main engine, side engines
    float* obs;              // [E][8] what the policy sees next (a fresh state where an episode ended)
    float* next_obs;         // [E][8] the observation that followed the action
    float* reward;           // [E] reward / reward_scale
    float* done;             // [E] 1: crashed, flew out, landed (and the time limit, with limit_terminal)
    float* trunc;            // [E] 1: time limit without limit_terminal
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, ld_act, time_limit, init;
    int limit_terminal;
    float reward_scale;
};
namespace lander {
constexpr float FPS = 50.f, W = 20.f, H = 400.f / 30.f, PAD_Y = H / 4.f, LEG_DOWN = 0.6f, FOOT_X = 20.f / 30.f;
constexpr float HULL_HALF_W = 17.f / 30.f, HULL_BOTTOM = -10.f / 30.f, HULL_TOP = 17.f / 30.f, HULL_TOP_HALF_W = 14.f / 30.f;
constexpr float MASS = 4.96f, INERTIA = 0.864f, GRAVITY = -10.f;
constexpr float MAIN_ACC = 13.f * (4.f / 30.f) / MASS * FPS, SIDE_ACC = 0.6f * (12.f / 30.f) / MASS * FPS;
constexpr float SIDE_ANG_ACC = 0.6f * (12.f / 30.f) * (14.f / 30.f) / INERTIA * FPS;
constexpr int SUBSTEPS = 4, REST_STEPS = 20;
constexpr float K_SPRING = 4000.f, C_DAMP = 120.f, C_FRICTION = 60.f, INIT_SPEED = 4.f, LEG_TRAVEL = 0.04f;
__device__ __forceinline__ void observe(const float* q, float c0, float c1, float* o) {
    o[0] = (q[0] - W / 2.f) / (W / 2.f);
    o[1] = (q[1] - (PAD_Y + LEG_DOWN)) / (H / 2.f);
    o[2] = q[2] * (W / 2.f) / FPS;
    o[3] = q[3] * (H / 2.f) / FPS;
    o[4] = q[4];
    o[5] = 20.f * q[5] / FPS;
    o[6] = c0;
    o[7] = c1;
}
__device__ __forceinline__ float shaping(const float* o) {
    return -100.f * sqrtf(o[0] * o[0] + o[1] * o[1]) - 100.f * sqrtf(o[2] * o[2] + o[3] * o[3]) - 100.f * fabsf(o[4]) +
           10.f * o[6] + 10.f * o[7];
}
}  // namespace lander
__global__ __launch_bounds__(256) void lander_step_kernel(const LanderArgs a) {
    using namespace lander;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    float* st = a.state + (size_t)e * 12;
    float q[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) q[i] = st[i];
    float c0 = st[6], c1 = st[7], prev = st[8], rest = st[9];
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        const float a0 = fminf(1.f, fmaxf(-1.f, a.action[(size_t)e * a.ld_act]));
        const float a1 = fminf(1.f, fmaxf(-1.f, a.action[(size_t)e * a.ld_act + 1]));
        const float main_p = a0 > 0.f ? 0.5f + 0.5f * a0 : 0.f;
        const float side = fabsf(a1) > 0.5f ? a1 : 0.f;
        const float h = 1.f / (FPS * SUBSTEPS);
        c0 = c1 = 0.f;
        bool broke = false;          // a foot pushed in beyond the leg's travel: the leg gives way
        for (int sub = 0; sub < SUBSTEPS; ++sub) {
            const float s = sinf(q[4]), c = cosf(q[4]);
            float ax = -s * MAIN_ACC * main_p + c * SIDE_ACC * side;
            float ay = c * MAIN_ACC * main_p + s * SIDE_ACC * side + GRAVITY;
            float alpha = -SIDE_ANG_ACC * side;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float fxb = k == 0 ? -FOOT_X : FOOT_X;
                const float rx = c * fxb + s * LEG_DOWN, ry = s * fxb - c * LEG_DOWN;
                const float pen = PAD_Y - (q[1] + ry);
                broke = broke || pen > LEG_TRAVEL;
                if (pen > 0.f) {
                    const float fvx = q[2] - q[5] * ry, fvy = q[3] + q[5] * rx;
                    const float fn = fmaxf(0.f, K_SPRING * pen - C_DAMP * fvy), ft = -C_FRICTION * fvx;
                    ax += ft / MASS;
                    ay += fn / MASS;
                    alpha += (rx * fn - ry * ft) / INERTIA;
                    if (k == 0) c0 = 1.f; else c1 = 1.f;
                }
            }
            q[2] += h * ax;
            q[3] += h * ay;
            q[5] += h * alpha;
            q[0] += h * q[2];
            q[1] += h * q[3];
            q[4] += h * q[5];
        }
        t += 1;
        const float s = sinf(q[4]), c = cosf(q[4]);
        const float lowest = fminf(fminf(q[1] - s * HULL_HALF_W + c * HULL_BOTTOM, q[1] + s * HULL_HALF_W + c * HULL_BOTTOM),
                                   fminf(q[1] - s * HULL_TOP_HALF_W + c * HULL_TOP, q[1] + s * HULL_TOP_HALF_W + c * HULL_TOP));
        float o[8];
        observe(q, c0, c1, o);
        const float sh = shaping(o);
        float rew = sh - prev - 0.30f * main_p - 0.03f * fabsf(side);
        prev = sh;
        const bool still = c0 > 0.f && c1 > 0.f && sqrtf(q[2] * q[2] + q[3] * q[3]) < 0.05f && fabsf(q[5]) < 0.05f;
        rest = still ? rest + 1.f : 0.f;
        const bool landed = rest >= (float)REST_STEPS;
        const bool failed = lowest < PAD_Y || broke || fabsf(o[0]) >= 1.f;
        rew = failed ? -100.f : (landed ? 100.f : rew);
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        const bool term = failed || landed;
#pragma unroll
        for (int i = 0; i < 8; ++i) a.next_obs[(size_t)e * 8 + i] = o[i];
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
        uint32_t r[4];
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x1A4DE2u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        q[0] = W / 2.f;
        q[1] = H;
        q[2] = ((float)(r[0] >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * INIT_SPEED;
        q[3] = ((float)(r[1] >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f) * INIT_SPEED;
        q[4] = q[5] = 0.f;
        c0 = c1 = rest = 0.f;
        float o[8];
        observe(q, 0.f, 0.f, o);
        prev = shaping(o);
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) st[i] = q[i];
    st[6] = c0;
    st[7] = c1;
    st[8] = prev;
    st[9] = rest;
    a.t[e] = t;
    float o[8];
    observe(q, c0, c1, o);
#pragma unroll
    for (int i = 0; i < 8; ++i) a.obs[(size_t)e * 8 + i] = o[i];
}
D4PG_API int d4pg_lander_step(const LanderArgs* a, hipStream_t stream) {
    if (a->E < 1 || a->state == nullptr || a->t == nullptr || a->obs == nullptr || a->counter == nullptr ||
        a->ep_return == nullptr || !(a->reward_scale > 0.f))
        return -1;
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < 2))
        return -2;
    hipLaunchKernelGGL(lander_step_kernel, dim3((a->E + 255) / 256), dim3(256), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(rollout_counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Shape-only environment of any observation / action size (envs/synthetic.py::SyntheticVec): a fixed random
// contraction s' = tanh(s M + a N + noise), reward -mean(s'^2) - 0.01 |a|^2, episodes end by time limit or with a
// small probability per step.  Two kernels per step: the new states (one thread per state component, reading
// the old state of its environment), then bookkeeping per environment.
struct SyntheticEnvArgs {
    float* state;            // [E][S]
    int* t;
    float* ep_return;
    const float* M;          // [S][S]
    const float* N;          // [A][S]
    const float* action;     // [E][ld_act]
    float* obs;              // [E][S]
    float* next_obs;         // [E][S]
    float* reward;
    float* done;
    float* trunc;
    float* noise;
    double* stats;
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, S, A, ld_act, time_limit, init;
    float terminal_prob, noise_std;
};
__device__ __forceinline__ float philox_normal(uint32_t a, uint32_t b, unsigned long long ctr, uint32_t stream,
                                               unsigned long long seed) {
    uint32_t r[4];
    philox4x32(a, b ^ (uint32_t)(ctr << 8), (uint32_t)(ctr >> 24), stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = fmaxf((float)(r[0] >> 8) * (1.0f / 16777216.0f), 1e-7f);
    const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}
__global__ __launch_bounds__(256) void synthetic_next_kernel(const SyntheticEnvArgs a) {
    const unsigned long long ctr = *a.counter;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)a.E * a.S) return;
    const int e = (int)(i / a.S), j = (int)(i - (long long)e * a.S);
    float acc = a.noise_std * philox_normal((uint32_t)e, (uint32_t)j, ctr, 0x51u, a.seed);
    const float* s = a.state + (size_t)e * a.S;
    for (int k = 0; k < a.S; ++k) acc += s[k] * a.M[(size_t)k * a.S + j];
    const float* u = a.action + (size_t)e * a.ld_act;
    for (int k = 0; k < a.A; ++k) acc += u[k] * a.N[(size_t)k * a.S + j];
    a.next_obs[i] = tanhf(acc);
}
__global__ __launch_bounds__(256) void synthetic_book_kernel(const SyntheticEnvArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    float* s = a.state + (size_t)e * a.S;
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        const float* ns = a.next_obs + (size_t)e * a.S;
        float sq = 0.f, ua = 0.f;
        for (int k = 0; k < a.S; ++k) {
            s[k] = ns[k];
            sq += ns[k] * ns[k];
        }
        for (int k = 0; k < a.A; ++k) ua += a.action[(size_t)e * a.ld_act + k] * a.action[(size_t)e * a.ld_act + k];
        const float rew = -sq / (float)a.S - 0.01f * ua;
        t += 1;
        uint32_t r[4];
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x7E2Du, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        const bool term = (float)(r[0] >> 8) * (1.0f / 16777216.0f) < a.terminal_prob;
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        a.reward[e] = rew;
        a.done[e] = (term || limit) ? 1.f : 0.f;          // like the host class: both read as done
        a.trunc[e] = 0.f;
        const float ret = a.ep_return[e] + rew;
        reset = term || limit;
        if (reset) {
            atomicAdd(&a.stats[0], (double)ret);
            atomicAdd(&a.stats[1], 1.0);
        }
        a.ep_return[e] = reset ? 0.f : ret;
    }
    if (reset) {
        for (int k = 0; k < a.S; ++k) s[k] = 0.5f * philox_normal((uint32_t)e, (uint32_t)k, ctr, 0xA5u, a.seed);
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
    a.t[e] = t;
    for (int k = 0; k < a.S; ++k) a.obs[(size_t)e * a.S + k] = s[k];
}
D4PG_API int d4pg_synthetic_env_step(const SyntheticEnvArgs* a, hipStream_t stream) {
    if (a->E < 1 || a->S < 1 || a->A < 1 || a->state == nullptr || a->t == nullptr || a->obs == nullptr ||
        a->counter == nullptr || a->ep_return == nullptr)
        return -1;
    if (!a->init) {
        if (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
            a->trunc == nullptr || a->stats == nullptr || a->M == nullptr || a->N == nullptr || a->ld_act < a->A)
            return -2;
        const long long n = (long long)a->E * a->S;
        hipLaunchKernelGGL(synthetic_next_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *a);
    }
    hipLaunchKernelGGL(synthetic_book_kernel, dim3((a->E + 255) / 256), dim3(256), 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(rollout_counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
struct RolloutRowsArgs {
    const float* obs;        // [T][E][S]   observation the action was chosen from
    const float* act;        // [T][E][ld_act]
    const float* next_obs;   // [T][E][S]   observation that followed (before any reset)
    const float* R;          // [T][E] n-step return          (csrc/nstep.hip)
    const float* gk;         // [T][E] gamma^k
    const int* nxt;          // [T][E] step whose next_obs is the bootstrap state
    const float* d;          // [T][E] terminal flag of that step
    float* rows;             // [T * E][row_ld]
    int T, E, S, A, ld_act, row_ld;
};
__global__ __launch_bounds__(256) void rollout_rows_kernel(const RolloutRowsArgs a) {
    const int W = 2 * a.S + a.A + 3;
    const long long total = (long long)a.T * a.E * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long te = i / W;
        const int c = (int)(i - te * W);
        const int t = (int)(te / a.E), e = (int)(te - (long long)t * a.E);
        float v;
        if (c < a.S) v = a.obs[te * a.S + c];
        else if (c < a.S + a.A) v = a.act[te * a.ld_act + (c - a.S)];
        else if (c == a.S + a.A) v = a.R[te];
        else if (c == a.S + a.A + 1) v = a.d[te];
        else if (c == a.S + a.A + 2) v = a.gk[te];
        else v = a.next_obs[((long long)a.nxt[te] * a.E + e) * a.S + (c - a.S - a.A - 3)];
        a.rows[te * a.row_ld + c] = v;
    }
}
D4PG_API int d4pg_rollout_rows(const RolloutRowsArgs* a, hipStream_t stream) {
    if (a->T < 1 || a->E < 1 || a->S < 1 || a->A < 1 || a->row_ld < 2 * a->S + a->A + 3 || a->ld_act < a->A) return -1;
    if (a->obs == nullptr || a->act == nullptr || a->next_obs == nullptr || a->R == nullptr || a->gk == nullptr ||
        a->nxt == nullptr || a->d == nullptr || a->rows == nullptr)
        return -2;
    const long long total = (long long)a->T * a->E * (2 * a->S + a->A + 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rollout_rows_kernel, dim3(blocks), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Inter-process sharing of a device buffer.  The exporting process allocates with d4pg_ipc_alloc (a
// dedicated hipMalloc allocation, so the handle names exactly this buffer) and passes the 64-byte handle
// on; any other process on the machine maps the same memory with d4pg_ipc_open.
D4PG_API int d4pg_ipc_alloc(size_t bytes, void** out, unsigned char* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    HIP_CHECK_RET(hipMalloc(out, bytes));
    HIP_CHECK_RET(hipMemset(*out, 0, bytes));
    hipIpcMemHandle_t h;
    HIP_CHECK_RET(hipIpcGetMemHandle(&h, *out));
    memcpy(handle64, &h, 64);
    return 0;
}
D4PG_API int d4pg_ipc_open(const unsigned char* handle64, void** out) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    HIP_CHECK_RET(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
D4PG_API int d4pg_ipc_close(void* p) { return (int)hipIpcCloseMemHandle(p); }
D4PG_API int d4pg_ipc_free(void* p) { return (int)hipFree(p); }
D4PG_API int d4pg_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
}

// Host memory that is not HIP's own (the explorers' POSIX shared-memory rings) made visible to the copy
// engines, so that rows go ring -> HBM with one DMA and no staging memcpy on the host.
D4PG_API int d4pg_host_register(void* p, size_t bytes) {
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();      // a refusal is an answer, not an error to leave behind for the next launch check
    return (int)e;
}
D4PG_API int d4pg_host_unregister(void* p) { return (int)hipHostUnregister(p); }
D4PG_API int d4pg_memcpy_h2d_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
}

D4PG_SIZEOF(PackMat32)
D4PG_SIZEOF(PackActorArgs32)
D4PG_SIZEOF(PendulumArgs)
D4PG_SIZEOF(RolloutRowsArgs)
D4PG_SIZEOF(CartPoleArgs)
D4PG_SIZEOF(SyntheticEnvArgs)
D4PG_SIZEOF(LanderArgs)

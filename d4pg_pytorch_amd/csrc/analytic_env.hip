// Reacher and the inverted double pendulum resident on the GPU (gfx950): envs/analytic.py::ReacherVec and
// InvertedDoublePendulumVec, the same closed-form dynamics in fp32, one thread per environment.  Constants are
// those of the host classes; change them together.
#include "common.h"

struct AnalyticEnvArgs {
    float* state;            // [E][8] reacher: q1, q2, q1', q2', target x, target y; pendulum: x, a1, a2, x', a1', a2'
    int* t;                  // [E]
    float* ep_return;        // [E]
    const float* action;     // [E][ld_act]
    float* obs;              // [E][11] what the policy sees next (a fresh state where an episode ended)
    float* next_obs;         // [E][11] the observation that followed the action
    float* reward;           // [E]
    float* done;             // [E]
    float* trunc;            // [E]
    float* noise;            // [E][ld_act] exploration-noise state, zeroed where an episode ends; or null
    double* stats;           // [2] sum of finished episode returns, number of finished episodes
    unsigned long long* counter;
    unsigned long long* counter_next;   // null: a follow-up launch increments `counter` in place; else this launch leaves counter + 1 here
    unsigned long long seed;
    int E, ld_act, time_limit, init;
    int limit_terminal, kind;        // kind: 0 Reacher, 1 InvertedDoublePendulum
    float reward_scale;
};

namespace analytic {
__device__ __forceinline__ float unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float sym(uint32_t r, float s) { return (2.f * unit(r) - 1.f) * s; }

namespace reacher {
constexpr float L1 = 0.1f, L2 = 0.11f, M1 = 0.0356f, M2 = 0.0392f, ARMATURE = 1.f, DAMPING = 1.f, GEAR = 200.f, DT = 0.02f;
constexpr int SUBSTEPS = 4;
__device__ __forceinline__ void tip(const float* s, float& x, float& y) {
    x = L1 * cosf(s[0]) + L2 * cosf(s[0] + s[1]);
    y = L1 * sinf(s[0]) + L2 * sinf(s[0] + s[1]);
}
__device__ __forceinline__ void observe(const float* s, float* o) {
    float x, y;
    tip(s, x, y);
    o[0] = cosf(s[0]), o[1] = cosf(s[1]), o[2] = sinf(s[0]), o[3] = sinf(s[1]);
    o[4] = s[4], o[5] = s[5], o[6] = s[2], o[7] = s[3];
    o[8] = x - s[4], o[9] = y - s[5], o[10] = 0.f;
}
__device__ __forceinline__ float step(float* s, const float* raw) {
    float x, y;
    tip(s, x, y);
    const float dx = x - s[4], dy = y - s[5];
    const float rew = -sqrtf(dx * dx + dy * dy) - (raw[0] * raw[0] + raw[1] * raw[1]);      // the cost uses the raw action
    const float t1 = GEAR * fminf(1.f, fmaxf(-1.f, raw[0])), t2 = GEAR * fminf(1.f, fmaxf(-1.f, raw[1]));
    constexpr float lc1 = L1 / 2, lc2 = L2 / 2, i1 = M1 * L1 * L1 / 12, i2 = M2 * L2 * L2 / 12, h = DT / SUBSTEPS;
    for (int k = 0; k < SUBSTEPS; ++k) {
        const float c2 = cosf(s[1]), s2 = sinf(s[1]);
        const float m11 = i1 + i2 + M1 * lc1 * lc1 + M2 * (L1 * L1 + lc2 * lc2 + 2 * L1 * lc2 * c2) + ARMATURE;
        const float m12 = i2 + M2 * (lc2 * lc2 + L1 * lc2 * c2), m22 = i2 + M2 * lc2 * lc2 + ARMATURE;
        const float hh = -M2 * L1 * lc2 * s2;
        const float b1 = t1 - hh * s[3] * (2 * s[2] + s[3]) - DAMPING * s[2];
        const float b2 = t2 + hh * s[2] * s[2] - DAMPING * s[3];
        const float det = m11 * m22 - m12 * m12;
        s[2] += h * (m22 * b1 - m12 * b2) / det;
        s[3] += h * (m11 * b2 - m12 * b1) / det;
        s[0] += h * s[2];
        s[1] += h * s[3];
    }
    return rew;
}
}  // namespace reacher

namespace idp {
constexpr float M0 = 1.f, M1 = 0.3f, M2 = 0.3f, L1 = 0.6f, L2 = 0.6f, G = 9.81f, FORCE = 20.f, DT = 0.05f, X_LIMIT = 1.f;
constexpr int SUBSTEPS = 10;
constexpr float d1 = M0 + M1 + M2, d2 = (M1 / 2 + M2) * L1, d3 = M2 * L2 / 2, d4 = (M1 / 3 + M2) * L1 * L1, d5 = M2 * L1 * L2 / 2,
                d6 = M2 * L2 * L2 / 3, f1 = (M1 / 2 + M2) * L1 * G, f2 = M2 * L2 / 2 * G;
__device__ __forceinline__ void observe(const float* s, float* o) {
    const float r1 = s[1], r2 = s[2] - s[1];
    o[0] = s[0], o[1] = sinf(r1), o[2] = sinf(r2), o[3] = cosf(r1), o[4] = cosf(r2);
    o[5] = fminf(10.f, fmaxf(-10.f, s[3]));
    o[6] = fminf(10.f, fmaxf(-10.f, s[4]));
    o[7] = fminf(10.f, fmaxf(-10.f, s[5] - s[4]));
    o[8] = o[9] = o[10] = 0.f;
}
__device__ __forceinline__ void step(float* s, float raw, float& rew, bool& fell) {
    const float u = fminf(1.f, fmaxf(-1.f, raw)) * FORCE;
    constexpr float h = DT / SUBSTEPS;
    for (int k = 0; k < SUBSTEPS; ++k) {
        const float c1 = cosf(s[1]), s1 = sinf(s[1]), c2 = cosf(s[2]), s2 = sinf(s[2]);
        const float c12 = cosf(s[1] - s[2]), s12 = sinf(s[1] - s[2]);
        // symmetric 3 x 3 system D x = rhs by its adjugate
        const float a = d1, b = d2 * c1, c = d3 * c2, d = d4, e = d5 * c12, f = d6;
        const float r0 = u + d2 * s1 * s[4] * s[4] + d3 * s2 * s[5] * s[5];
        const float r1 = f1 * s1 - d5 * s12 * s[5] * s[5];
        const float r2 = f2 * s2 + d5 * s12 * s[4] * s[4];
        const float A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
        const float det = a * A + b * B + c * C;
        const float D2 = a * f - c * c, E2 = b * c - a * e, F2 = a * d - b * b;
        s[3] += h * (A * r0 + B * r1 + C * r2) / det;
        s[4] += h * (B * r0 + D2 * r1 + E2 * r2) / det;
        s[5] += h * (C * r0 + E2 * r1 + F2 * r2) / det;
        s[0] += h * s[3];
        s[1] += h * s[4];
        s[2] += h * s[5];
    }
    if (fabsf(s[0]) > X_LIMIT) {             // the rail ends: the cart stops there
        s[0] = fminf(X_LIMIT, fmaxf(-X_LIMIT, s[0]));
        s[3] = 0.f;
    }
    const float x_tip = s[0] + L1 * sinf(s[1]) + L2 * sinf(s[2]), y_tip = L1 * cosf(s[1]) + L2 * cosf(s[2]);
    const float v1 = s[4], v2 = s[5] - s[4];
    rew = 10.f - (0.01f * x_tip * x_tip + (y_tip - 2.f) * (y_tip - 2.f)) - (1e-3f * v1 * v1 + 5e-3f * v2 * v2);
    fell = y_tip <= 1.f;
}
}  // namespace idp

template <int KIND>
__global__ __launch_bounds__(256) void analytic_step_kernel(const AnalyticEnvArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ctr = *a.counter;
    next_counter(a.counter_next, ctr);
    if (e >= a.E) return;
    float* st = a.state + (size_t)e * 8;
    float s[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = st[i];
    int t = a.t[e];
    bool reset = a.init != 0;
    if (!a.init) {
        float rew;
        bool term = false;
        if (KIND == 0) {
            const float raw[2] = {a.action[(size_t)e * a.ld_act], a.action[(size_t)e * a.ld_act + 1]};
            rew = reacher::step(s, raw);
        } else {
            idp::step(s, a.action[(size_t)e * a.ld_act], rew, term);
        }
        t += 1;
        const bool limit = a.time_limit > 0 && t >= a.time_limit;
        float o[11];
        if (KIND == 0) reacher::observe(s, o); else idp::observe(s, o);
#pragma unroll
        for (int i = 0; i < 11; ++i) a.next_obs[(size_t)e * 11 + i] = o[i];
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
        uint32_t r[4], r2[4];
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0xA7A1u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
        philox4x32((uint32_t)e, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0xA7A2u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r2);
        if (KIND == 0) {
            s[0] = sym(r[0], 0.1f), s[1] = sym(r[1], 0.1f), s[2] = sym(r[2], 0.005f), s[3] = sym(r[3], 0.005f);
            // a target uniform in the disk of radius 0.2: radius by the square root, angle uniform
            const float rad = 0.2f * sqrtf(unit(r2[0])), phi = 6.2831853f * unit(r2[1]);
            s[4] = rad * cosf(phi), s[5] = rad * sinf(phi);
        } else {
            s[0] = sym(r[0], 0.1f), s[1] = sym(r[1], 0.1f), s[2] = s[1] + sym(r[2], 0.1f);
            // three normal deviates (Box-Muller) x 0.1
            const float m1 = sqrtf(-2.f * logf(fmaxf(unit(r2[0]), 1e-7f))), m2 = sqrtf(-2.f * logf(fmaxf(unit(r2[2]), 1e-7f)));
            s[3] = 0.1f * m1 * cosf(6.2831853f * unit(r2[1]));
            s[4] = 0.1f * m1 * sinf(6.2831853f * unit(r2[1]));
            s[5] = 0.1f * m2 * cosf(6.2831853f * unit(r2[3]));
        }
        t = 0;
        if (a.init) a.ep_return[e] = 0.f;
        if (a.noise != nullptr)
            for (int j = 0; j < a.ld_act; ++j) a.noise[(size_t)e * a.ld_act + j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) st[i] = s[i];
    a.t[e] = t;
    float o[11];
    if (KIND == 0) reacher::observe(s, o); else idp::observe(s, o);
#pragma unroll
    for (int i = 0; i < 11; ++i) a.obs[(size_t)e * 11 + i] = o[i];
}

__global__ void counter_kernel(unsigned long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1ull;
}
}  // namespace analytic

D4PG_API int d4pg_analytic_env_step(const AnalyticEnvArgs* a, hipStream_t stream) {
    if (a->E < 1 || a->state == nullptr || a->t == nullptr || a->obs == nullptr || a->counter == nullptr ||
        a->ep_return == nullptr || !(a->reward_scale > 0.f) || a->kind < 0 || a->kind > 1)
        return -1;
    if (!a->init && (a->action == nullptr || a->next_obs == nullptr || a->reward == nullptr || a->done == nullptr ||
                     a->trunc == nullptr || a->stats == nullptr || a->ld_act < (a->kind == 0 ? 2 : 1)))
        return -2;
    const dim3 grid((a->E + 255) / 256), block(256);
    if (a->kind == 0) hipLaunchKernelGGL(analytic::analytic_step_kernel<0>, grid, block, 0, stream, *a);
    else hipLaunchKernelGGL(analytic::analytic_step_kernel<1>, grid, block, 0, stream, *a);
    if (a->counter_next == nullptr) hipLaunchKernelGGL(analytic::counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}
D4PG_SIZEOF(AnalyticEnvArgs)

// The planar articulated-body simulator of envs/planar.py in fp32, ONE source for two builds: the device kernels
// (csrc/planar_env.hip, one thread per environment) and the host runtime's explorer environments
// (csrc/planar_host.cpp, one loop iteration per environment).
//
// The mechanism is a tree of hinged links under a free-floating root, reduced coordinates
// q = (x, y, root angle, joint angles...).  Per sub-step and environment: link frames down the tree, the mass
// matrix M = sum m J'J + I Jw'Jw from the links' centre-of-mass Jacobians (a Jacobian column is non-zero only for
// the hinges above the link), bias forces from the centripetal accelerations, gravity, joint springs / friction /
// soft limits, spring-damper ground contacts with a friction cone, a Cholesky solve of the (at most 9 x 9) system,
// symplectic Euler.  The TOPOLOGY of each mechanism (parents, which link a contact point sits on) is a template
// parameter packed into integer constants: after unrolling, every index into the per-environment arrays is a
// compile-time constant, so frames, Jacobians and the mass matrix live in registers and the structural zeros of
// the Jacobians cost nothing.  The NUMBERS (masses, lengths, gains...) come from the Python model at run time
// (PlanarParams), so the three implementations cannot drift apart.
#pragma once
// Pointer parameters of the simulator functions are not restrict-qualified: with `__restrict__` on the parameter
// block hipcc 7.2 miscompiles the 3-D twin of this code at -O2 / -O3 (csrc/ant_sim.h has the evidence); nothing
// here ever writes through an alias, so the qualifier bought nothing but that exposure.
#ifndef PSIM_RESTRICT
#define PSIM_RESTRICT
#endif
#include <math.h>
#if defined(__HIPCC__)
#define PSIM_RCP(x) __frcp_rn(x)
#define PSIM_RSQRT(x) __frsqrt_rn(x)
#define PSIM_FN __device__ __forceinline__
#define PSIM_SINCOS(x, s, c) __sincosf((x), (s), (c))
#define PSIM_FINITE(x) isfinite(x)
#else
#define PSIM_RCP(x) (1.f / (x))
#define PSIM_RSQRT(x) (1.f / sqrtf(x))
#define PSIM_FN static inline
#define PSIM_SINCOS(x, s, c) sincosf((x), (s), (c))
#define PSIM_FINITE(x) std::isfinite(x)
#include <cmath>
#endif

struct PlanarParams {
    float anchor[8][2], com[8][2], mass[8], inertia[8], rest[8], lo[8], hi[8], damping[8], armature[8], stiffness[8];
    float c_local[8][2];
    float gear[8];
    float gravity, k_contact, c_contact, c_friction, mu, k_limit, c_limit;
    float dt, alive, ctrl_cost, reset_noise, q0_y, z_off, z_min, z_max, ang_max;
    int substeps, n_links, n_contacts, n_act;
    unsigned long long parents, contact_links;      // the host model's topology, packed like the templates': checked at launch
};
namespace planar {
constexpr int STRIDE = 24, QD0 = 9, FLAG0 = 18, PREV = 20;
constexpr unsigned long long pack8(int a = -1, int b = -1, int c = -1, int d = -1, int e = -1, int f = -1, int g = -1, int h = -1) {
    return (unsigned long long)(a + 1) | (unsigned long long)(b + 1) << 4 | (unsigned long long)(c + 1) << 8 |
           (unsigned long long)(d + 1) << 12 | (unsigned long long)(e + 1) << 16 | (unsigned long long)(f + 1) << 20 |
           (unsigned long long)(g + 1) << 24 | (unsigned long long)(h + 1) << 28;
}
constexpr int unpack(unsigned long long v, int i) { return (int)((v >> (4 * i)) & 15ull) - 1; }

struct HopperT {
    static constexpr int L = 4, C = 4, A = 3, TASK = 0, S = 11;
    static constexpr unsigned long long PAR = pack8(-1, 0, 1, 2), CL = pack8(3, 3, 0, 0);
};
struct WalkerT {
    static constexpr int L = 7, C = 6, A = 6, TASK = 0, S = 17;
    static constexpr unsigned long long PAR = pack8(-1, 0, 1, 2, 0, 4, 5), CL = pack8(3, 3, 6, 6, 0, 0);
};
struct CheetahT {
    static constexpr int L = 7, C = 7, A = 6, TASK = 0, S = 17;
    static constexpr unsigned long long PAR = pack8(-1, 0, 1, 2, 0, 4, 5), CL = pack8(3, 6, 2, 5, 0, 0, 0);
};
struct BipedT {
    static constexpr int L = 5, C = 6, A = 4, TASK = 1, S = 24;
    static constexpr unsigned long long PAR = pack8(-1, 0, 1, 0, 3), CL = pack8(2, 4, 0, 0, 0, 0);
};
// the walker's task constants (envs/planar.py::BipedalWalkerVec): change them together
namespace biped {
constexpr float FPS = 50.f, SCALE = 30.f, LEG_H = 34.f / 30.f, LEG_DOWN = -8.f / 30.f;
constexpr float MOTORS_TORQUE = 80.f, SPEED_HIP = 4.f, SPEED_KNEE = 6.f, LIDAR_RANGE = 160.f / 30.f;
constexpr float TRACK_END = (200 - 10) * 14.f / 30.f, START_X = 14.f / 30.f * 20.f / 2.f, SERVO_GAIN = 6.f, HULL_MASS = 5.42f;
constexpr float TERRAIN_STEP = 14.f / 30.f, INV_STEP = 30.f / 14.f;
// sine and cosine of the lidar angles 0, 0.15, ... 1.35 rad
#define LIDAR_SIN_TABLE {0.f, 0.14943813f, 0.29552021f, 0.43496553f, 0.56464247f, 0.68163876f, 0.78332691f, 0.86742323f, 0.93203909f, 0.97572336f}
#define LIDAR_COS_TABLE {1.f, 0.98877108f, 0.95533649f, 0.90044710f, 0.82533561f, 0.73168887f, 0.62160997f, 0.49757105f, 0.36235775f, 0.21900669f}
constexpr int TERRAIN_LENGTH = 200, TERRAIN_STARTPAD = 20, LIDAR_SEGMENTS = 13;
// the ground is piecewise linear between the samples terrain[j] at x = j * TERRAIN_STEP
// The stretch of ground the walker can touch during one step (feet and hull corners stay within 2.5 of the hull's
// centre), read ONCE per step: inside the sub-steps a lookup is a chain of selects on registers instead of two
// dependent memory loads on the critical path of a latency-bound thread.
constexpr int WINDOW = 12;
struct TerrainWindow {
    float h[WINDOW];
    int j0;
    bool on;
};
PSIM_FN void load_window(const float* PSIM_RESTRICT terrain, int ts, float x, TerrainWindow& w) {
    w.on = terrain != nullptr;
    int j0 = (int)floorf((x - 2.5f) * INV_STEP);
    w.j0 = j0 < 0 ? 0 : (j0 > TERRAIN_LENGTH - WINDOW ? TERRAIN_LENGTH - WINDOW : j0);
#pragma unroll
    for (int k = 0; k < WINDOW; ++k) w.h[k] = w.on ? terrain[(size_t)(w.j0 + k) * ts] : 0.f;       // ts: stride between samples
}
PSIM_FN void ground(const TerrainWindow& w, float x, float& g, float& slope) {
    int j = (int)floorf(x * INV_STEP) - w.j0;
    j = j < 0 ? 0 : (j > WINDOW - 2 ? WINDOW - 2 : j);          // beyond the window: its edge segment, extended
    float h0 = w.h[0], h1 = w.h[1];
#pragma unroll
    for (int k = 1; k < WINDOW - 1; ++k)
        if (k == j) h0 = w.h[k], h1 = w.h[k + 1];
    slope = (h1 - h0) * INV_STEP;
    g = h0 + slope * (x - (float)(w.j0 + j) * TERRAIN_STEP);
}
}  // namespace biped

template <class T>
constexpr unsigned long long ancestry() {          // bit 8 i + k: hinge k turns link i (k is i itself or one of its ancestors)
    unsigned long long m = 0ull;
    for (int i = 0; i < T::L; ++i)
        for (int k = i; k >= 0; k = unpack(T::PAR, k)) m |= 1ull << (8 * i + k);
    return m;
}
template <class T>
constexpr unsigned long long ANCESTRY = ancestry<T>();
template <class T>
constexpr bool above(int link, int k) { return (ANCESTRY<T> >> (8 * link + k)) & 1ull; }

// q'' for joint torques tau; sets the bits of `contact` for the contact points that touch
template <class T>
PSIM_FN void accelerations(const PlanarParams* PSIM_RESTRICT P, const float (&q)[T::L + 2], const float (&qd)[T::L + 2],
                                              const float (&tau)[T::L], float (&qdd)[T::L + 2], unsigned& contact,
                                              const biped::TerrainWindow& terrain) {
    constexpr int L = T::L, n = L + 2;
    float ang[L], ox[L], oy[L], w[L], sn[L], cs[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const int p = unpack(T::PAR, i);
        if (p < 0) {
            ang[i] = q[2] + P->rest[i];
            ox[i] = q[0];
            oy[i] = q[1];
            w[i] = qd[2];
        } else {
            ox[i] = ox[p] + cs[p] * P->anchor[i][0] - sn[p] * P->anchor[i][1];
            oy[i] = oy[p] + sn[p] * P->anchor[i][0] + cs[p] * P->anchor[i][1];
            ang[i] = ang[p] + P->rest[i] + q[2 + i];
            w[i] = w[p] + qd[2 + i];
        }
        PSIM_SINCOS(ang[i], &sn[i], &cs[i]);
    }
    float M[n][n], Q[n];
#pragma unroll
    for (int a = 0; a < n; ++a) {
        Q[a] = 0.f;
#pragma unroll
        for (int b = 0; b < n; ++b) M[a][b] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float m = P->mass[i], I = P->inertia[i];
        const float px = ox[i] + cs[i] * P->com[i][0] - sn[i] * P->com[i][1];
        const float py = oy[i] + sn[i] * P->com[i][0] + cs[i] * P->com[i][1];
        float jx[n], jy[n];
        jx[0] = 1.f, jy[0] = 0.f, jx[1] = 0.f, jy[1] = 1.f;
        float a0x = 0.f, a0y = 0.f, ex = px, ey = py;
#pragma unroll
        for (int k = L - 1; k >= 0; --k) {
            if (above<T>(i, k)) {
                jx[2 + k] = -(py - oy[k]);
                jy[2 + k] = px - ox[k];
                const float w2 = w[k] * w[k];
                a0x -= w2 * (ex - ox[k]);
                a0y -= w2 * (ey - oy[k]);
                ex = ox[k], ey = oy[k];
            } else {
                jx[2 + k] = 0.f, jy[2 + k] = 0.f;
            }
        }
        M[0][0] += m;
        M[1][1] += m;
        Q[0] -= m * a0x;
        Q[1] -= m * (a0y + P->gravity);
#pragma unroll
        for (int a = 2; a < n; ++a) {
            if (!above<T>(i, a - 2)) continue;
            M[a][0] += m * jx[a];
            M[a][1] += m * jy[a];
            Q[a] -= m * (jx[a] * a0x + jy[a] * (a0y + P->gravity));
#pragma unroll
            for (int b = 2; b <= a; ++b)
                if (above<T>(i, b - 2)) M[a][b] += m * (jx[a] * jx[b] + jy[a] * jy[b]) + I;
        }
        if (i > 0) {
            M[2 + i][2 + i] += P->armature[i];
            const float ja = q[2 + i], jv = qd[2 + i];
            float t = tau[i] - P->damping[i] * jv - P->stiffness[i] * ja;
            const float over = ja - P->hi[i], under = P->lo[i] - ja;
            if (over > 0.f) t -= P->k_limit * over + P->c_limit * jv;
            if (under > 0.f) t += P->k_limit * under - P->c_limit * jv;
            Q[2 + i] += t;
        }
    }
#pragma unroll
    for (int c = 0; c < T::C; ++c) {
        const int link = unpack(T::CL, c);
        const float px = ox[link] + cs[link] * P->c_local[c][0] - sn[link] * P->c_local[c][1];
        const float py = oy[link] + sn[link] * P->c_local[c][0] + cs[link] * P->c_local[c][1];
        // the ground under the point: flat at y = 0, or the walker's terrain (normal (-slope, 1) / r, tangent (1, slope) / r)
        float g = 0.f, slope = 0.f, ir = 1.f;             // ir = 1 / |(-slope, 1)|
        if (T::TASK == 1 && terrain.on) {
            biped::ground(terrain, px, g, slope);
            ir = PSIM_RSQRT(1.f + slope * slope);
        }
        const float pen = (g - py) * ir;
        if (pen > 0.f) {
            float vx = qd[0], vy = qd[1];
#pragma unroll
            for (int k = 0; k < L; ++k)
                if (above<T>(link, k)) {
                    vx -= (py - oy[k]) * qd[2 + k];
                    vy += (px - ox[k]) * qd[2 + k];
                }
            const float vn = (vy - slope * vx) * ir, vt = (vx + slope * vy) * ir;
            const float fn = fmaxf(0.f, P->k_contact * pen - P->c_contact * vn);
            const float ft = fminf(P->mu * fn, fmaxf(-P->mu * fn, -P->c_friction * vt));
            const float fx = (ft - slope * fn) * ir, fy = (fn + slope * ft) * ir;
            Q[0] += fx;
            Q[1] += fy;
#pragma unroll
            for (int k = 0; k < L; ++k)
                if (above<T>(link, k)) Q[2 + k] += -(py - oy[k]) * fx + (px - ox[k]) * fy;
            contact |= 1u << c;
        }
    }
    // Cholesky M = G G' in the lower triangle, then two triangular solves
#pragma unroll
    for (int j = 0; j < n; ++j) {
        float d = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= M[j][k] * M[j][k];
        d = sqrtf(d);
        M[j][j] = d;
        const float inv = 1.f / d;
#pragma unroll
        for (int i = j + 1; i < n; ++i) {
            float s = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= M[i][k] * M[j][k];
            M[i][j] = s * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) {
        float s = Q[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= M[i][k] * qdd[k];
        qdd[i] = s / M[i][i];
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
        float s = qdd[i];
#pragma unroll
        for (int k = i + 1; k < n; ++k) s -= M[k][i] * qdd[k];
        qdd[i] = s / M[i][i];
    }
}

template <class T>
PSIM_FN void observe(const float (&q)[T::L + 2], const float (&qd)[T::L + 2], float c0, float c1, float* PSIM_RESTRICT o,
                     const float* PSIM_RESTRICT terrain, int ts) {
    constexpr int n = T::L + 2;
    constexpr float LIDAR_SIN[10] = LIDAR_SIN_TABLE, LIDAR_COS[10] = LIDAR_COS_TABLE;
    if (T::TASK == 0) {
#pragma unroll
        for (int i = 1; i < n; ++i) o[i - 1] = q[i];
#pragma unroll
        for (int i = 0; i < n; ++i) o[n - 1 + i] = fminf(10.f, fmaxf(-10.f, qd[i]));
    } else {
        using namespace biped;
        o[0] = q[2];
        o[1] = 2.f * qd[2] / FPS;
        o[2] = 0.3f * qd[0] * (600.f / SCALE) / FPS;
        o[3] = 0.3f * qd[1] * (400.f / SCALE) / FPS;
        o[4] = q[3], o[5] = qd[3] / SPEED_HIP, o[6] = q[4] + 1.f, o[7] = qd[4] / SPEED_KNEE, o[8] = c0;
        o[9] = q[5], o[10] = qd[5] / SPEED_HIP, o[11] = q[6] + 1.f, o[12] = qd[6] / SPEED_KNEE, o[13] = c1;
        // 10 lidar rays from the hull, 0 .. 1.35 rad from straight down towards the front (world axes): the fraction of
        // the range at which each meets the ground.  The 14 heights the rays can reach are read once; the first
        // segment also covers what lies behind its left end, the last one what lies beyond the track.
        if (terrain == nullptr) {
#pragma unroll
            for (int i = 0; i < 10; ++i) o[14 + i] = fminf(1.f, fmaxf(0.f, q[1] / (LIDAR_COS[i] * LIDAR_RANGE)));
        } else {
            int j0 = (int)floorf(q[0] * INV_STEP);
            j0 = j0 < 0 ? 0 : (j0 > TERRAIN_LENGTH - 2 ? TERRAIN_LENGTH - 2 : j0);
            float hgt[LIDAR_SEGMENTS + 1];
#pragma unroll
            for (int k = 0; k <= LIDAR_SEGMENTS; ++k) {
                const int j = j0 + k;
                hgt[k] = terrain[(size_t)(j > TERRAIN_LENGTH - 1 ? TERRAIN_LENGTH - 1 : j) * ts];
            }
            const int last = TERRAIN_LENGTH - 2 - j0;       // segments k >= last are the track's final one, repeated
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float dx = LIDAR_SIN[i] * LIDAR_RANGE, dy = -LIDAR_COS[i] * LIDAR_RANGE;
                float best = 1.f;
                // ray i spans at most dx / TERRAIN_STEP segments beyond the one under the hull (a compile-time bound)
                const int reach = (int)(LIDAR_SIN[i] * LIDAR_RANGE * INV_STEP) + 2;
#pragma unroll
                for (int k = 0; k < LIDAR_SEGMENTS; ++k) {
                    if (k >= reach) continue;
                    const bool tail = k >= last;            // clamped onto the final segment of the track
                    const float h0 = tail ? terrain[(size_t)(TERRAIN_LENGTH - 2) * ts] : hgt[k];
                    const float h1 = tail ? terrain[(size_t)(TERRAIN_LENGTH - 1) * ts] : hgt[k + 1];
                    const float xj = (float)(tail ? TERRAIN_LENGTH - 2 : j0 + k) * TERRAIN_STEP;
                    const float sl = (h1 - h0) * INV_STEP;
                    const float t = (q[1] - h0 - sl * (q[0] - xj)) * PSIM_RCP(sl * dx - dy);
                    const float xh = q[0] + t * dx;
                    const bool lo_ok = (k == 0 || j0 + k == 0) || xh >= xj;
                    const bool hi_ok = tail || xh <= xj + TERRAIN_STEP;
                    if (t >= 0.f && lo_ok && hi_ok && t < best) best = t;
                }
                o[14 + i] = fminf(1.f, fmaxf(0.f, best));
            }
        }
    }
}

// One environment step without the episode bookkeeping: sub-steps, then reward and termination of the task.
// `raw` points at the T::A action components of this environment.
template <class T>
PSIM_FN void advance(const PlanarParams* PSIM_RESTRICT P, float (&q)[T::L + 2], float (&qd)[T::L + 2], const float* PSIM_RESTRICT raw,
                     float& c0, float& c1, float& prev, float& rew, bool& term, const float* PSIM_RESTRICT terrain,
                     int terrain_stride) {
    constexpr int L = T::L, n = L + 2;
    float act[T::A], raw2 = 0.f, abs_sum = 0.f;
#pragma unroll
    for (int j = 0; j < T::A; ++j) {
        const float r = raw[j];
        act[j] = fminf(1.f, fmaxf(-1.f, r));
        raw2 += act[j] * act[j];
        abs_sum += fminf(1.f, fabsf(r));
    }
    const float x0 = q[0];
    const float h = P->dt / (float)P->substeps;
    unsigned contact = 0u;
    biped::TerrainWindow window;
    window.on = false;
    if (T::TASK == 1) biped::load_window(terrain, terrain_stride, q[0], window);
    for (int sub = 0; sub < P->substeps; ++sub) {
        float tau[L], qdd[n];
        tau[0] = 0.f;
        if (T::TASK == 0) {
#pragma unroll
            for (int j = 0; j < T::A; ++j) tau[1 + j] = P->gear[j] * act[j];
        } else {
            using namespace biped;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float speed = (j & 1) ? SPEED_KNEE : SPEED_HIP;
                const float sg = act[j] > 0.f ? 1.f : (act[j] < 0.f ? -1.f : 0.f);
                const float limit = MOTORS_TORQUE * fabsf(act[j]);
                tau[1 + j] = fminf(limit, fmaxf(-limit, SERVO_GAIN * MOTORS_TORQUE * (speed * sg - qd[3 + j]) / speed));
            }
        }
        accelerations<T>(P, q, qd, tau, qdd, contact, window);
#pragma unroll
        for (int i = 0; i < n; ++i) {
            qd[i] += h * qdd[i];
            q[i] += h * qd[i];
        }
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < n; ++i) bad = bad || !PSIM_FINITE(q[i]) || !PSIM_FINITE(qd[i]);
    if (T::TASK == 0) {
        if (bad) {
#pragma unroll
            for (int i = 0; i < n; ++i) q[i] = 0.f, qd[i] = 0.f;
            q[1] = P->q0_y;
        }
        const float speed = bad ? 0.f : (q[0] - x0) / P->dt;
        rew = P->alive + speed - P->ctrl_cost * raw2;
        const float z = q[1] + P->z_off;
        term = bad || z < P->z_min || z > P->z_max || fabsf(q[2]) > P->ang_max;
    } else {
        using namespace biped;
        c0 = (contact & 1u) ? 1.f : 0.f;
        c1 = (contact & 2u) ? 1.f : 0.f;
        const float shaping = 130.f * q[0] / SCALE - 5.f * fabsf(q[2]);
        rew = shaping - prev - 0.00035f * MOTORS_TORQUE * abs_sum;
        prev = shaping;
        const bool failed = (contact >> 2) != 0u || q[0] < 0.f || bad;
        if (failed) rew = -100.f;
        term = failed || q[0] > TRACK_END;
        if (bad) {
#pragma unroll
            for (int i = 0; i < n; ++i) q[i] = 0.f, qd[i] = 0.f;
        }
    }
}
}  // namespace planar

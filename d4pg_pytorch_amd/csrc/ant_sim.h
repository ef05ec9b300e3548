// The ant of envs/spatial.py (a free-floating torso with four two-joint legs in 3-D) in fp32, ONE source for two
// builds: the device kernel (csrc/ant_env.hip, one thread per environment) and the host runtime's explorer
// environments (csrc/planar_host.cpp, one loop iteration per environment).
//
// Per sub-step and environment: link frames (quaternion root, Rodrigues rotations about the hinges), the 14 x 14
// mass matrix M = sum m Jv'Jv + Jw' (R I R') Jw from the Jacobian COLUMNS of each link's centre of mass (a link has
// at most 8 non-zero columns: 3 translations, 3 rotations of the torso, its own hinge and its parent's), bias forces
// from the accelerations at u' = 0, joint damping / soft limits, 13 contact spheres with the spring-dampers taken
// implicitly in the velocity (M + h J'CJ + h^2 J'KJ), a Cholesky solve, symplectic Euler with the quaternion
// renormalised.  The topology is compiled in (which hinges move which link are constexpr functions of the unrolled
// loop indices, so the lower triangle of M, the columns and the frames stay in registers); every NUMBER comes from
// the Python model (AntParams), so the three implementations cannot drift apart.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define ASIM_FN __device__ __forceinline__
#define ASIM_SINCOS(x, s, c) __sincosf((x), (s), (c))
#define ASIM_FINITE(x) isfinite(x)
#else
#include <cmath>
#define ASIM_FN static inline
#ifdef ASIM_NO_SINCOS
#define ASIM_SINCOS(x, s, c) (*(s) = sinf(x), *(c) = cosf(x))
#else
#define ASIM_SINCOS(x, s, c) sincosf((x), (s), (c))
#endif
#define ASIM_FINITE(x) std::isfinite(x)
#endif

struct AntParams {
    float anchor[9][3], axis[9][3], com[9][3], mass[9], inertia[9][9], lo[9], hi[9], damping[9], armature[9];
    float c_local[13][3], c_rad[13];
    float gravity, k_contact, c_contact, c_friction, mu, k_limit, c_limit;
    float dt, gear, z_init, reset_noise, ctrl_cost, contact_cost, alive, z_min, z_max, ankle0[4];
    int substeps, n_links, n_contacts, pad_;
    int parent[9], c_link[13];            // the host model's topology: checked at launch against the compiled one
};
// hipcc 7.2, gfx950, -O2 / -O3: when the ~100 loads from the parameter block are hoisted out of the sub-step loop of
// advance() (which the optimiser does whenever it can prove them invariant: `__restrict__` on the pointer, or a
// pointer that comes straight out of the kernel arguments), the kernel needs all 106 scalar registers plus spills
// for them across a body of 450 vector registers, and the 14 x 14 system comes out wrong from the FIRST sub-step on:
// entries such as M[3][1] read 0.0197 or 2.4e15 for 0.0527, and every state with ground contact ends non-finite.
// What it is not: undefined behaviour in this source (host builds at -O0 / -O3 agree bit for bit and are clean under
// ASan, UBSan and MSan; no -Wall -Wextra -Wuninitialized finding), floating-point contraction (-ffp-contract=off),
// type-based aliasing (-fno-strict-aliasing), loop unrolling (-fno-unroll-loops) or where spills go
// (-amdgpu-spill-sgpr-to-vgpr=false, -amdgpu-spill-vgpr-to-agpr=false): none of these changes the wrong result.
// accelerations() called once outside a loop is right at every level; the loop is right at -O1, and at -O3 as soon
// as the loads cannot leave it.  So the pointer passes through an empty asm at the top of every sub-step (34 scalar
// registers instead of 106, no scalar spills), the qualifier is gone, and tests/test_spatial_env.py holds the device
// build to the host simulator step by step.  Drivers, variants and their outputs: bench/exp/ant_rootcause/,
// profiles/ant_rootcause/.
#ifndef ASIM_RESTRICT
#define ASIM_RESTRICT
#endif
#ifndef ASIM_LOOP_BODY_BEGIN
#if defined(__HIPCC__)
#define ASIM_LOOP_BODY_BEGIN(P) asm volatile("" : "+s"(P));
#else
#define ASIM_LOOP_BODY_BEGIN(P)
#endif
#endif
namespace ant {
constexpr int L = 9, N = 14, C = 13, STRIDE = 32, S = 111;
struct V3 {
    float x, y, z;
};
ASIM_FN V3 v3(float x, float y, float z) { return V3{x, y, z}; }
ASIM_FN V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
ASIM_FN V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
ASIM_FN V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
ASIM_FN V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
ASIM_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ASIM_FN V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
ASIM_FN float comp(V3 a, int j) { return j == 0 ? a.x : (j == 1 ? a.y : a.z); }
struct M3 {
    float m[3][3];
};
ASIM_FN V3 mul(const M3& A, V3 v) {
    return V3{A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
              A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
ASIM_FN M3 mul(const M3& A, const M3& B) {
    M3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return o;
}
ASIM_FN V3 colv(const M3& A, int j) { return V3{A.m[0][j], A.m[1][j], A.m[2][j]}; }

// topology: link 0 the torso; leg k: hip link 1 + 2k (child of the torso), ankle link 2 + 2k (child of the hip);
// hinge j (0..7) belongs to link j + 1; velocity column 6 + j
constexpr int parent_of(int link) { return link == 0 ? -1 : ((link & 1) ? 0 : link - 1); }
constexpr bool moves(int link, int j) { return link == j + 1 || (link >= 2 && !(link & 1) && link - 1 == j + 1); }
constexpr bool active(int link, int col) { return col < 6 || moves(link, col - 6); }
constexpr int contact_link(int c) { return c == 0 ? 0 : (((c - 1) % 3 == 0) ? 1 + 2 * ((c - 1) / 3) : 2 + 2 * ((c - 1) / 3)); }
// slot of every simulated link among the published model's 14 bodies (AntVec.BODY_SLOT)
constexpr int body_slot(int link) { return link == 0 ? 1 : ((link & 1) ? 3 + 3 * ((link - 1) / 2) : 4 + 3 * ((link - 2) / 2)); }

struct Frames {
    M3 R[L];
    V3 o[L], ax[L], w[L];
};

// Jacobian column `col` (>= 3) of a point p on `link`: rotations of the torso, then the hinges above the link
ASIM_FN V3 column(const Frames& F, int link, int col, V3 p) {
    if (col < 6) {
        const V3 r = p - F.o[0];
        return col == 3 ? v3(0.f, -r.z, r.y) : (col == 4 ? v3(r.z, 0.f, -r.x) : v3(-r.y, r.x, 0.f));     // e_j x r
    }
    const int k = col - 6 + 1;
    return cross(F.ax[k], p - F.o[k]);
}

// u' for hinge torques tau; `wrench` [L][6] receives (moment about the link's centre of mass, force) of the ground
ASIM_FN void accelerations(const AntParams* ASIM_RESTRICT P, const float (&pos)[3], const float (&quat)[4],
                                              const float (&qj)[8], const float (&u)[N], const float (&tau)[8], float h,
                                              float (&ud)[N], float (&wrench)[L][6]
#ifdef ASIM_DEBUG          // bench/exp/ant_rootcause: the system before it is factorised
                                              , float* dbg = nullptr
#endif
) {
    Frames F;
    {
        const float w = quat[0], x = quat[1], y = quat[2], z = quat[3];
        F.R[0].m[0][0] = 1 - 2 * (y * y + z * z), F.R[0].m[0][1] = 2 * (x * y - w * z), F.R[0].m[0][2] = 2 * (x * z + w * y);
        F.R[0].m[1][0] = 2 * (x * y + w * z), F.R[0].m[1][1] = 1 - 2 * (x * x + z * z), F.R[0].m[1][2] = 2 * (y * z - w * x);
        F.R[0].m[2][0] = 2 * (x * z - w * y), F.R[0].m[2][1] = 2 * (y * z + w * x), F.R[0].m[2][2] = 1 - 2 * (x * x + y * y);
        F.o[0] = v3(pos[0], pos[1], pos[2]);
        F.ax[0] = v3(0.f, 0.f, 0.f);
        F.w[0] = v3(u[3], u[4], u[5]);
    }
    float M[N][N], Q[N];
#pragma unroll
    for (int a = 0; a < N; ++a) {
        Q[a] = 0.f;
#pragma unroll
        for (int b = 0; b < N; ++b) M[a][b] = 0.f;
    }
    V3 a_o[L], alpha[L], com[L];
    a_o[0] = alpha[0] = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = j + 1;
        M[6 + j][6 + j] += P->armature[i];
        const float qd = u[6 + j];
        float t = tau[j] - P->damping[i] * qd;
        const float over = qj[j] - P->hi[i], under = P->lo[i] - qj[j];
        if (over > 0.f) t -= P->k_limit * over + P->c_limit * qd;
        if (under > 0.f) t += P->k_limit * under - P->c_limit * qd;
        Q[6 + j] += t;
    }
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
    // the torso (g = 0), then one leg after the other: a leg's frames, Jacobian columns and contact points are dead
    // before the next leg starts, which shortens the live ranges of everything but M, Q and the torso's frame.
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int first = g == 0 ? 0 : 2 * g - 1, last = g == 0 ? 0 : 2 * g;
        const int c_first = g == 0 ? 0 : 3 * g - 2, c_last = g == 0 ? 0 : 3 * g;
#pragma unroll
        for (int i = first; i <= last; ++i) {
            if (i == 0) continue;
            const int p = parent_of(i);
            const V3 a = ld3(P->axis[i]);
            float sn, cs;
            ASIM_SINCOS(qj[i - 1], &sn, &cs);
            // Rodrigues: I + sin K + (1 - cos) K^2, K = skew(a), K^2 = a a' - I for a unit axis
            const float oc = 1.f - cs;
            M3 rot;
            rot.m[0][0] = 1.f + oc * (a.x * a.x - 1.f), rot.m[0][1] = -sn * a.z + oc * a.x * a.y, rot.m[0][2] = sn * a.y + oc * a.x * a.z;
            rot.m[1][0] = sn * a.z + oc * a.x * a.y, rot.m[1][1] = 1.f + oc * (a.y * a.y - 1.f), rot.m[1][2] = -sn * a.x + oc * a.y * a.z;
            rot.m[2][0] = -sn * a.y + oc * a.x * a.z, rot.m[2][1] = sn * a.x + oc * a.y * a.z, rot.m[2][2] = 1.f + oc * (a.z * a.z - 1.f);
            F.R[i] = mul(F.R[p], rot);
            F.o[i] = F.o[p] + mul(F.R[p], ld3(P->anchor[i]));
            F.ax[i] = mul(F.R[p], a);
            F.w[i] = F.w[p] + u[6 + i - 1] * F.ax[i];
        }
#pragma unroll
        for (int i = first; i <= last; ++i) {
            const int p = parent_of(i);
            if (i > 0) {
                const V3 r = F.o[i] - F.o[p];
                a_o[i] = a_o[p] + cross(alpha[p], r) + cross(F.w[p], cross(F.w[p], r));
                alpha[i] = alpha[p] + u[6 + i - 1] * cross(F.w[p], F.ax[i]);
            }
            const float m = P->mass[i];
            com[i] = F.o[i] + mul(F.R[i], ld3(P->com[i]));
            const V3 rc = com[i] - F.o[i];
            const V3 a_c = a_o[i] + cross(alpha[i], rc) + cross(F.w[i], cross(F.w[i], rc));
            const V3 f_lin = m * (v3(0.f, 0.f, -P->gravity) - a_c);
            M3 I0, Iw;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) I0.m[r][c] = P->inertia[i][3 * r + c];
            {
                const M3 RI = mul(F.R[i], I0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Iw.m[r][c] = RI.m[r][0] * F.R[i].m[c][0] + RI.m[r][1] * F.R[i].m[c][1] + RI.m[r][2] * F.R[i].m[c][2];
            }
            const V3 t_rot = -1.f * (mul(Iw, alpha[i]) + cross(F.w[i], mul(Iw, F.w[i])));
            V3 jv[N], jw[N], ijw[N];
#pragma unroll
            for (int c = 3; c < N; ++c) {
                if (!active(i, c)) continue;
                jv[c] = column(F, i, c, com[i]);
                jw[c] = c < 6 ? v3(c == 3 ? 1.f : 0.f, c == 4 ? 1.f : 0.f, c == 5 ? 1.f : 0.f) : F.ax[c - 6 + 1];
                ijw[c] = c < 6 ? colv(Iw, c - 3) : mul(Iw, jw[c]);
            }
            M[0][0] += m, M[1][1] += m, M[2][2] += m;
            Q[0] += f_lin.x, Q[1] += f_lin.y, Q[2] += f_lin.z;
#pragma unroll
            for (int c = 3; c < N; ++c) {
                if (!active(i, c)) continue;
                M[c][0] += m * jv[c].x, M[c][1] += m * jv[c].y, M[c][2] += m * jv[c].z;
                Q[c] += dot(jv[c], f_lin) + dot(jw[c], t_rot);
#pragma unroll
                for (int d = 3; d <= c; ++d)
                    if (active(i, d)) M[c][d] += m * dot(jv[c], jv[d]) + dot(jw[c], ijw[d]);
            }
        }
#pragma unroll
        for (int c = c_first; c <= c_last; ++c) {
            const int link = contact_link(c);
            V3 p = F.o[link] + mul(F.R[link], ld3(P->c_local[c]));
            p.z -= P->c_rad[c];                                     // the sphere's lowest point
            if (p.z < 0.f) {
                V3 col[N];
                V3 v = v3(u[0], u[1], u[2]);
#pragma unroll
                for (int k = 3; k < N; ++k) {
                    if (!active(link, k)) continue;
                    col[k] = column(F, link, k, p);
                    v = v + u[k] * col[k];
                }
                float fn = -P->k_contact * p.z - P->c_contact * v.z;
                const bool push = fn > 0.f;
                fn = push ? fn : 0.f;
                float fx = -P->c_friction * v.x, fy = -P->c_friction * v.y;
                const float norm = sqrtf(fx * fx + fy * fy), cap = P->mu * fn;
                const bool slide = norm > cap;
                if (slide) {
                    const float s = cap / fmaxf(norm, 1e-12f);
                    fx *= s, fy *= s;
                }
                if (!push) fx = fy = 0.f;
                const V3 f = v3(fx, fy, fn);
                Q[0] += f.x, Q[1] += f.y, Q[2] += f.z;
                const float gxy = (push && !slide) ? h * P->c_friction : 0.f;
                const float gz = push ? h * P->c_contact + h * h * P->k_contact : 0.f;
                M[0][0] += gxy, M[1][1] += gxy, M[2][2] += gz;
#pragma unroll
                for (int k = 3; k < N; ++k) {
                    if (!active(link, k)) continue;
                    Q[k] += dot(col[k], f);
                    const V3 g = v3(gxy * col[k].x, gxy * col[k].y, gz * col[k].z);
                    M[k][0] += g.x, M[k][1] += g.y, M[k][2] += g.z;
#pragma unroll
                    for (int d = 3; d <= k; ++d)
                        if (active(link, d)) M[k][d] += dot(g, col[d]);
                }
                const V3 mom = cross(p - com[link], f);
                wrench[link][0] += mom.x, wrench[link][1] += mom.y, wrench[link][2] += mom.z;
                wrench[link][3] += f.x, wrench[link][4] += f.y, wrench[link][5] += f.z;
            }
        }
    }
#ifdef ASIM_DEBUG
    if (dbg != nullptr) {
        int n = 0;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) dbg[n++] = M[a][b];
#pragma unroll
        for (int a = 0; a < N; ++a) dbg[n++] = Q[a];
    }
#endif
    // Cholesky M = G G' in the lower triangle, then two triangular solves
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float d = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= M[j][k] * M[j][k];
        d = sqrtf(d);
        M[j][j] = d;
        const float inv = 1.f / d;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            float s = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= M[i][k] * M[j][k];
            M[i][j] = s * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = Q[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= M[i][k] * ud[k];
        ud[i] = s / M[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        float s = ud[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= M[k][i] * ud[k];
        ud[i] = s / M[i][i];
    }
}

ASIM_FN void observe(const float (&pos)[3], const float (&quat)[4], const float (&qj)[8], const float (&u)[N],
                                        const float (&wrench)[L][6], float* o) {
    o[0] = pos[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[1 + i] = quat[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[5 + i] = qj[i];
    o[13] = u[0], o[14] = u[1], o[15] = u[2];
    {
        const float w = quat[0], x = quat[1], y = quat[2], z = quat[3];
        // R' w: the angular velocity in torso axes
        const float r00 = 1 - 2 * (y * y + z * z), r01 = 2 * (x * y - w * z), r02 = 2 * (x * z + w * y);
        const float r10 = 2 * (x * y + w * z), r11 = 1 - 2 * (x * x + z * z), r12 = 2 * (y * z - w * x);
        const float r20 = 2 * (x * z - w * y), r21 = 2 * (y * z + w * x), r22 = 1 - 2 * (x * x + y * y);
        o[16] = r00 * u[3] + r10 * u[4] + r20 * u[5];
        o[17] = r01 * u[3] + r11 * u[4] + r21 * u[5];
        o[18] = r02 * u[3] + r12 * u[4] + r22 * u[5];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) o[19 + i] = u[6 + i];
#pragma unroll
    for (int i = 0; i < 84; ++i) o[27 + i] = 0.f;
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) o[27 + 6 * body_slot(i) + k] = fminf(1.f, fmaxf(-1.f, wrench[i][k]));
}

// One environment step without the episode bookkeeping: sub-steps, then reward and termination of the task.
// `raw` points at the 8 action components of this environment.
ASIM_FN void advance(const AntParams* ASIM_RESTRICT P_in, float (&pos)[3], float (&quat)[4], float (&qj)[8], float (&u)[N],
                     const float* ASIM_RESTRICT raw, float (&wrench)[L][6], float& rew, bool& term
#ifdef ASIM_DEBUG
                     , float* dbg = nullptr
#endif
) {
    const AntParams* P = P_in;
    float tau[8], act2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float c = fminf(1.f, fmaxf(-1.f, raw[j]));
        tau[j] = P->gear * c;
        act2 += c * c;
    }
    const float x0 = pos[0];
    const float h = P->dt / (float)P->substeps;
    for (int sub = 0; sub < P->substeps; ++sub) {
        float ud[N];
        ASIM_LOOP_BODY_BEGIN(P)
#ifdef ASIM_DEBUG
        accelerations(P, pos, quat, qj, u, tau, h, ud, wrench, sub == 0 ? dbg : nullptr);
        if (sub == 0 && dbg != nullptr)
            for (int i = 0; i < N; ++i) dbg[119 + i] = ud[i];
#else
        accelerations(P, pos, quat, qj, u, tau, h, ud, wrench);
#endif
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] += h * ud[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pos[i] += h * u[i];
        const float wx = u[3], wy = u[4], wz = u[5];
        const float q0 = quat[0], q1 = quat[1], q2 = quat[2], q3 = quat[3];
        float n0 = q0 + 0.5f * h * (-(wx * q1 + wy * q2 + wz * q3));
        float n1 = q1 + 0.5f * h * (q0 * wx + wy * q3 - wz * q2);
        float n2 = q2 + 0.5f * h * (q0 * wy + wz * q1 - wx * q3);
        float n3 = q3 + 0.5f * h * (q0 * wz + wx * q2 - wy * q1);
        const float inv = 1.f / sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
        quat[0] = n0 * inv, quat[1] = n1 * inv, quat[2] = n2 * inv, quat[3] = n3 * inv;
#pragma unroll
        for (int i = 0; i < 8; ++i) qj[i] += h * u[6 + i];
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) bad = bad || !ASIM_FINITE(pos[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bad = bad || !ASIM_FINITE(quat[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) bad = bad || !ASIM_FINITE(qj[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) bad = bad || !ASIM_FINITE(u[i]);
    if (bad) {                                   // a blown-up state never reaches the policy
        pos[0] = pos[1] = 0.f, pos[2] = P->z_init;
        quat[0] = 1.f, quat[1] = quat[2] = quat[3] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) qj[i] = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) u[i] = 0.f;
#pragma unroll
        for (int i = 0; i < L; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
    }
    const float speed = bad ? 0.f : (pos[0] - x0) / P->dt;
    float w2 = 0.f;
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float c = fminf(1.f, fmaxf(-1.f, wrench[i][k]));
            w2 += c * c;
        }
    rew = speed + P->alive - P->ctrl_cost * act2 - P->contact_cost * w2;
    term = pos[2] < P->z_min || pos[2] > P->z_max || bad;
}
}  // namespace ant

// Host-side explorer environments on the planar articulated-body simulator (csrc/planar_sim.h, the source the
// device kernels are built from): Hopper, Walker2d, HalfCheetah, BipedalWalker advanced by one environment step,
// a plain loop over the environments of one explorer process.  Episode bookkeeping (resets, time limits) stays in
// Python (envs/native_planar.py): it runs once per episode, this runs once per step.
#include <stddef.h>
#include <stdint.h>

#include "planar_sim.h"

namespace {
template <class T>
int step_all(const PlanarParams* P, int E, float* state, const float* action, int ld_act, float* next_obs, float* reward,
             uint8_t* term_out, const float* terrain) {
    using namespace planar;
    constexpr int n = T::L + 2;
    if (P->n_links != T::L || P->n_contacts != T::C || P->n_act != T::A || P->parents != T::PAR || P->contact_links != T::CL)
        return -3;       // not the mechanism this instantiation was compiled for
    for (int e = 0; e < E; ++e) {
        float* st = state + (size_t)e * STRIDE;
        float q[n], qd[n];
        for (int i = 0; i < n; ++i) q[i] = st[i], qd[i] = st[QD0 + i];
        float c0 = st[FLAG0], c1 = st[FLAG0 + 1], prev = st[PREV], rew;
        bool term;
        const float* ter = (T::TASK == 1 && terrain != nullptr) ? terrain + (size_t)e * biped::TERRAIN_LENGTH : nullptr;
        advance<T>(P, q, qd, action + (size_t)e * ld_act, c0, c1, prev, rew, term, ter, 1);
        for (int i = 0; i < n; ++i) st[i] = q[i], st[QD0 + i] = qd[i];
        st[FLAG0] = c0, st[FLAG0 + 1] = c1, st[PREV] = prev;
        observe<T>(q, qd, c0, c1, next_obs + (size_t)e * T::S, ter, 1);
        reward[e] = rew;
        term_out[e] = term ? 1 : 0;
    }
    return 0;
}
}  // namespace

// kind: 0 Hopper, 1 Walker2d, 2 HalfCheetah, 3 BipedalWalker.  state [E][24] (q at 0, q' at 9, leg-contact flags at
// 18 / 19, previous shaping at 20), action [E][ld_act], next_obs [E][S], reward [E], term [E] (terminal state of the
// task, time limits excluded), terrain [E][200] ground heights of the walker (null: flat).
extern "C" __attribute__((visibility("default"))) int d4pg_planar_host_step(const PlanarParams* P, int kind, int E, float* state,
                                                                            const float* action, int ld_act, float* next_obs,
                                                                            float* reward, uint8_t* term, const float* terrain) {
    if (P == nullptr || E < 1 || state == nullptr || action == nullptr || next_obs == nullptr || reward == nullptr || term == nullptr)
        return -1;
    if (P->substeps < 1 || P->substeps > 64 || !(P->dt > 0.f)) return -4;
    switch (kind) {
        case 0: return ld_act < 3 ? -2 : step_all<planar::HopperT>(P, E, state, action, ld_act, next_obs, reward, term, terrain);
        case 1: return ld_act < 6 ? -2 : step_all<planar::WalkerT>(P, E, state, action, ld_act, next_obs, reward, term, terrain);
        case 2: return ld_act < 6 ? -2 : step_all<planar::CheetahT>(P, E, state, action, ld_act, next_obs, reward, term, terrain);
        case 3: return ld_act < 4 ? -2 : step_all<planar::BipedT>(P, E, state, action, ld_act, next_obs, reward, term, terrain);
        default: return -5;
    }
}
extern "C" __attribute__((visibility("default"))) int d4pg_sizeof_host_PlanarParams() { return (int)sizeof(PlanarParams); }

// ---------------------------------------------------------------------------------------------------
// The ant (csrc/ant_sim.h): state [E][32] = position 3, quaternion 4, hinge angles 8, velocity 14; next_obs [E][111].
#include "ant_sim.h"

extern "C" __attribute__((visibility("default"))) int d4pg_ant_host_step(const AntParams* P, int E, float* state, const float* action,
                                                                         int ld_act, float* next_obs, float* reward, uint8_t* term_out) {
    using namespace ant;
    if (P == nullptr || E < 1 || state == nullptr || action == nullptr || next_obs == nullptr || reward == nullptr ||
        term_out == nullptr || ld_act < 8)
        return -1;
    if (P->n_links != L || P->n_contacts != C) return -3;
    for (int i = 0; i < L; ++i)
        if (P->parent[i] != parent_of(i)) return -3;       // not the mechanism this code was compiled for
    for (int c = 0; c < C; ++c)
        if (P->c_link[c] != contact_link(c)) return -3;
    if (P->substeps < 1 || P->substeps > 64 || !(P->dt > 0.f)) return -4;
    for (int e = 0; e < E; ++e) {
        float* st = state + (size_t)e * STRIDE;
        float pos[3], quat[4], qj[8], u[N], wrench[L][6], rew;
        bool term;
        for (int i = 0; i < 3; ++i) pos[i] = st[i];
        for (int i = 0; i < 4; ++i) quat[i] = st[3 + i];
        for (int i = 0; i < 8; ++i) qj[i] = st[7 + i];
        for (int i = 0; i < N; ++i) u[i] = st[15 + i];
        for (int i = 0; i < L; ++i)
            for (int k = 0; k < 6; ++k) wrench[i][k] = 0.f;
        advance(P, pos, quat, qj, u, action + (size_t)e * ld_act, wrench, rew, term);
        for (int i = 0; i < 3; ++i) st[i] = pos[i];
        for (int i = 0; i < 4; ++i) st[3 + i] = quat[i];
        for (int i = 0; i < 8; ++i) st[7 + i] = qj[i];
        for (int i = 0; i < N; ++i) st[15 + i] = u[i];
        observe(pos, quat, qj, u, wrench, next_obs + (size_t)e * S);
        reward[e] = rew;
        term_out[e] = term ? 1 : 0;
    }
    return 0;
}
extern "C" __attribute__((visibility("default"))) int d4pg_sizeof_host_AntParams() { return (int)sizeof(AntParams); }

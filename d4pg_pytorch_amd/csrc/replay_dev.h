// Device-side pieces of the HBM replay's priority tree that more than one translation unit
// needs (csrc/replay.hip, csrc/fused.hip).
#pragma once
#include "common.h"

constexpr int MAXL = 6;

struct TreeDesc {
    float* leaf;            // [n[0]] priority^alpha, 0 for empty slots
    double* sum[MAXL + 1];  // sum[l], l = 1..L : [n[l]] fp64 sums of 64 children
    float* mn[MAXL + 1];    // mn[l],  l = 1..L : [n[l]] min over non-empty leaves below (+inf if none)
    long long n[MAXL + 1];  // entries per level, each padded to a multiple of 64; n[L] <= 64
    int L;                  // top level index (0: the leaves themselves fit one node)
    int pad_;
};

struct ReplayDesc {
    float* rows;            // [capacity, row_ld]
    long long capacity;
    int S, A, row_ld;
    int pad_;
};

__device__ __forceinline__ double tree_load_sum(const TreeDesc& t, int level, long long idx) {
    if (idx >= t.n[level]) return 0.0;
    return level == 0 ? (double)t.leaf[idx] : t.sum[level][idx];
}
__device__ __forceinline__ float tree_load_min(const TreeDesc& t, int level, long long idx) {
    if (idx >= t.n[level]) return INFINITY;
    if (level == 0) {
        const float v = t.leaf[idx];
        return v > 0.f ? v : INFINITY;
    }
    return t.mn[level][idx];
}

__device__ __forceinline__ double wave_scan_incl_d(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// ---------------------------------------------------------------------------
// Tree update: leaf[idx[i]] = priority[i]^alpha, then recompute every touched ancestor level by level
// (recomputed from the children, so the fp64 sums never drift).
//
// `parts` workgroups share the work by OWNERSHIP of subtrees: the node of level L-1 above an item
// (idx >> 6(L-1)) belongs to workgroup (node % parts), and so does everything below it.  A workgroup
// therefore only ever reads what it wrote itself (or what no one writes): leaf writes and levels
// 1 .. L-1 need no synchronisation between workgroups, only __syncthreads() between the levels.  The top
// level (at most 64 nodes) is recomputed by whichever workgroup finishes last (ticket counter behind a
// __threadfence(), children read at device scope): nobody waits for anybody.
struct TreeUpdateArgs {
    TreeDesc t;
    const long long* idx;   // [n]
    const float* prio;      // [n] raw priorities
    DevState* st;
    unsigned int* ticket;   // parts > 1: zero-initialised counter, left at zero again
    int n;
    float alpha;
    int parts;              // workgroups that call tree_update_body (1 .. 64); > 1 needs t.L >= 2
    int pad_;
};

// LDS the body needs to hold the items a workgroup owns (worst case all n of them): a counter, then per item its
// slot (8 bytes) and its priority^alpha (4 bytes).
__host__ __device__ __forceinline__ int tree_update_lds_bytes(int n) { return 16 + ((n + 1) & ~1) * 12; }

// One node of level l >= 1 recomputed from its 64 children (contiguous: 256 B of leaves, or 512 B of fp64 sums
// + 256 B of minima), fetched with 16-byte loads that are all in flight together.  `coherent`: the children
// were written by other workgroups of this launch (other CUs), read them past this CU's vector cache.
__device__ __forceinline__ void tree_recompute_node(const TreeDesc& t, int l, long long g, bool coherent) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    double s = 0.0;
    float m = INFINITY;
    if (l == 1) {
        const f32x4* c = reinterpret_cast<const f32x4*>(t.leaf + g * 64);
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = c[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s += (double)v[k][e];
                m = fminf(m, v[k][e] > 0.f ? v[k][e] : INFINITY);
            }
        }
    } else if (coherent) {
        const double* cs = t.sum[l - 1] + g * 64;
        const float* cm = t.mn[l - 1] + g * 64;
        for (int k = 0; k < 64; ++k) {
            s += __hip_atomic_load(cs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = fminf(m, __hip_atomic_load(cm + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    } else {
        const f64x2* cs = reinterpret_cast<const f64x2*>(t.sum[l - 1] + g * 64);
        const f32x4* cm = reinterpret_cast<const f32x4*>(t.mn[l - 1] + g * 64);
        f32x4 mv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) mv[k] = cm[k];
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const f64x2 v = cs[k];
            s += v[0];
            s += v[1];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) m = fminf(fminf(m, fminf(mv[k][0], mv[k][1])), fminf(mv[k][2], mv[k][3]));
    }
    t.sum[l][g] = s;
    t.mn[l][g] = m;
}

// NWAVES = waves of the calling workgroup (all of them must call this); part = 0 .. a.parts - 1;
// lds = tree_update_lds_bytes(a.n) bytes of LDS, 8-byte aligned (lds_bytes = what is there: with less, the items
// are filtered again at every level instead of once).  Returns true (to every thread) in the workgroup that
// leaves the tree complete: the only one, or the last to finish of several.
//
// The workgroup first COMPACTS the items it owns into LDS (slot, priority^alpha): the loads of the index list
// are independent and pipeline, and afterwards item j of the list belongs to thread j % threads, so a level
// costs one memory round trip however the owned items were spread over the batch (filtering the batch again
// at every level made a thread with three owned items pay three dependent round trips per level, and every
// thread one load of its item's index per level on top: 70 us for 4096 items in 8 workgroups, 10M rows).
template <int NWAVES>
__device__ __forceinline__ bool tree_update_body(const TreeUpdateArgs& a, int part, void* lds, int lds_bytes) {
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int nthreads = 64 * NWAVES;
    const int W = a.parts > 1 ? a.parts : 1;
    const int L = a.t.L;
    const int top_shift = 6 * (L > 1 ? L - 1 : 0);
    auto mine = [&](long long node, int level) {          // level <= L-1
        return W == 1 || (int)((node >> (top_shift - 6 * level)) % W) == part;
    };
    int* cnt = reinterpret_cast<int*>(lds);               // [0] items owned, [1] last-arriver flag
    long long* l_slot = reinterpret_cast<long long*>((char*)lds + 16);
    float* l_leaf = reinterpret_cast<float*>((char*)lds + 16 + (size_t)((a.n + 1) & ~1) * 8);
    const bool listed = lds_bytes >= tree_update_lds_bytes(a.n);
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    float local_max = 0.f;
    // four items per thread and trip: all eight loads are issued before the first is used (one memory round
    // trip per trip; an item's priority is loaded whether the workgroup owns it or not)
    for (int i0 = tid; i0 < a.n; i0 += 4 * nthreads) {
        long long sl[4];
        float pr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * nthreads;
            sl[k] = i < a.n ? a.idx[i] : -1;
            pr[k] = i < a.n ? a.prio[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (sl[k] < 0 || !mine(sl[k], 0)) continue;
            float p = pr[k];
            if (!(p > 0.f) || p > 3.0e38f) p = 1e-4f;   // NaN / inf / non-positive never enter the tree
            local_max = fmaxf(local_max, p);
            const float leaf = powf(p, a.alpha);
            if (listed) {
                const int pos = atomicAdd(&cnt[0], 1);
                l_slot[pos] = sl[k];
                l_leaf[pos] = leaf;
            } else {
                a.t.leaf[sl[k]] = leaf;
            }
        }
    }
    local_max = wave_max(local_max);
    // positive floats order like their bit patterns
    if (lane == 0 && local_max > 0.f)
        atomicMax(reinterpret_cast<unsigned int*>(&a.st->max_priority), __float_as_uint(local_max));
    __syncthreads();
    const int m = listed ? cnt[0] : a.n;
    if (listed)
        for (int j = tid; j < m; j += nthreads) a.t.leaf[l_slot[j]] = l_leaf[j];
    __syncthreads();
    // One THREAD per touched parent (duplicate parents are recomputed to the same value).  A level with far
    // fewer nodes than the batch has items is recomputed whole (the top level: 4 nodes for 1M rows).
    for (int l = 1; l <= L; ++l) {
        const long long nodes = a.t.n[l - 1] >> 6;       // nodes of this level that have children (levels are padded)
        const bool top = l == L && W > 1;
        if (top) {
            // everything this workgroup wrote is visible device-wide before its ticket is
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                const unsigned int t = atomicAdd(a.ticket, 1u);
                cnt[1] = t == (unsigned int)(W - 1);
                if (cnt[1]) *a.ticket = 0u;            // ready for the next update (stream order)
            }
            __syncthreads();
            if (!cnt[1]) return false;
            __threadfence();
        }
        const bool whole = top || nodes * 8 <= (long long)a.n;
        const int tasks = whole ? (int)nodes : m;
        for (int i = tid; i < tasks; i += nthreads) {
            long long g;
            if (whole) {
                g = i;
                if (!top && !mine(g, l)) continue;
            } else if (listed) {
                g = l_slot[i] >> (6 * l);
            } else {
                g = a.idx[i] >> (6 * l);
                if (!mine(g, l)) continue;
            }
            tree_recompute_node(a.t, l, g, top);
        }
        __syncthreads();
    }
    return true;
}

// ---------------------------------------------------------------------------
// Sampling: one wave draws one row.
struct SampleArgs {
    ReplayDesc rp;
    TreeDesc t;
    DevState* st;
    float* sa;        // [B, ld_x]  cols [0,S) = state, [S,S+A) = action
    float* s_pi;      // [B, ld_x]  cols [0,S) = state          (action columns are produced by the actor)
    float* s2_a2;     // [B, ld_x]  cols [0,S) = next state     (action columns are produced by the target actor)
    float* reward;    // [B]
    float* done;      // [B]
    float* gamma;     // [B]
    float* weights;   // [B]
    long long* idx;   // [B]
    int B, ld_x;
    int prioritized;
    int do_tick;
    float beta1, beta2;           // Adam betas for the tick
    float per_b0, per_b1;         // importance-exponent schedule
    long long total_steps;
};

// Draw the replay slot of sample `b` of `B` (all 64 lanes of the wave take part and get the
// same result): uniform, or stratified proportional through the fan-out-64 tree with the
// importance weight (p_min / p_leaf)^beta.  `step` selects the RNG stream position and the
// point on the beta schedule.
__device__ __forceinline__ long long draw_slot(const TreeDesc& t, int prioritized, int b, int B, long long step,
                                               long long size, unsigned long long seed, float per_b0, float per_b1,
                                               long long total_steps, int lane, float& w) {
    uint32_t rnd[4];
    philox4x32((uint32_t)b, (uint32_t)step, (uint32_t)(step >> 32), 0x5EEDu, (uint32_t)seed, (uint32_t)(seed >> 32),
               rnd);
    const double u = u01_from_bits(rnd[0], rnd[1]);
    long long slot;
    w = 1.f;
    if (!prioritized) {
        slot = (long long)(u * (double)size);
        if (slot >= size) slot = size - 1;
        return slot;
    }
    // root totals from the top level
    const double top = tree_load_sum(t, t.L, lane);
    const double total = wave_sum_d(top);
    const float pmin = wave_min(tree_load_min(t, t.L, lane));
    double target = ((double)b + u) * (total / (double)B);
    long long g = 0;
    float pleaf = 0.f;
    for (int l = t.L; l >= 0; --l) {
        const long long c = g * 64 + lane;
        const double v = tree_load_sum(t, l, c);
        const double incl = wave_scan_incl_d(v, lane);
        const unsigned long long live = __ballot(v > 0.0);
        const unsigned long long hit = __ballot(v > 0.0 && incl > target);
        int pick;
        if (hit) pick = __ffsll((long long)hit) - 1;
        else if (live) pick = 63 - __clzll((long long)live);   // rounding pushed target past the end
        else pick = 0;
        const double excl = __shfl(incl - v, pick, 64);
        target -= excl;
        if (target < 0.0) target = 0.0;
        g = g * 64 + pick;
        if (l == 0) pleaf = (float)__shfl(v, pick, 64);
    }
    slot = g < size ? g : size - 1;
    // beta schedule evaluated locally (per_beta in DevState belongs to the tick)
    double frac = (double)step / (double)(total_steps > 0 ? total_steps : 1);
    frac = frac > 1.0 ? 1.0 : frac;
    const float beta = (float)(per_b0 + frac * (per_b1 - per_b0));
    w = (pleaf > 0.f && pmin < INFINITY) ? powf(pmin / pleaf, beta) : 1.f;
    return slot;
}


// Draw sample `b` of the batch consumed by the update at stream position `step` and gather its row
// into the learner's input buffers (all 64 lanes of one wave).
__device__ __forceinline__ void sample_row(const SampleArgs& a, int b, long long step, long long size, int lane) {
    float w;
    const long long slot = draw_slot(a.t, a.prioritized, b, a.B, step, size, a.st->seed, a.per_b0, a.per_b1,
                                     a.total_steps, lane, w);
    const float* row = a.rp.rows + slot * a.rp.row_ld;
    const int S = a.rp.S, A = a.rp.A;
    const size_t ob = (size_t)b * a.ld_x;
    for (int c = lane; c < S; c += 64) {
        const float s = row[c];
        a.sa[ob + c] = s;
        a.s_pi[ob + c] = s;
        a.s2_a2[ob + c] = row[S + A + 3 + c];
    }
    for (int c = lane; c < A; c += 64) a.sa[ob + S + c] = row[S + c];
    if (lane == 0) {
        a.reward[b] = row[S + A];
        a.done[b] = row[S + A + 1];
        a.gamma[b] = row[S + A + 2];
        a.weights[b] = w;
        a.idx[b] = slot;
    }
}

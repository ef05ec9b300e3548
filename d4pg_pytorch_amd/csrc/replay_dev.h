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
// Tree update: leaf[idx[i]] = priority[i]^alpha, then recompute every touched
// ancestor level by level.  One workgroup (16 waves); __syncthreads() between
// levels orders the global writes of one level before the reads of the next.
struct TreeUpdateArgs {
    TreeDesc t;
    const long long* idx;   // [n]
    const float* prio;      // [n] raw priorities
    DevState* st;
    int n;
    float alpha;
};

// NWAVES = waves of the calling workgroup (all of them must call this)
template <int NWAVES>
__device__ __forceinline__ void tree_update_body(const TreeUpdateArgs& a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float local_max = 0.f;
    for (int i = tid; i < a.n; i += 64 * NWAVES) {
        float p = a.prio[i];
        if (!(p > 0.f) || p > 3.0e38f) p = 1e-4f;   // NaN / inf / non-positive never enter the tree
        a.t.leaf[a.idx[i]] = powf(p, a.alpha);
        local_max = fmaxf(local_max, p);
    }
    local_max = wave_max(local_max);
    // positive floats order like their bit patterns
    if (lane == 0 && local_max > 0.f)
        atomicMax(reinterpret_cast<unsigned int*>(&a.st->max_priority), __float_as_uint(local_max));
    __syncthreads();
    // Per level every wave handles TU touched parents at a time: all index loads, then all
    // 64-wide child loads are issued before the first reduction, so a level costs about two
    // memory round trips instead of two per parent.
    // A level with far fewer nodes than the batch has items is recomputed whole (the top level:
    // 4 nodes for a 1M-row replay) instead of once per item.
    constexpr int TU = 8;
    for (int l = 1; l <= a.t.L; ++l) {
        const bool whole = a.t.n[l] * 8 <= (long long)a.n;
        const int tasks = whole ? (int)a.t.n[l] : a.n;
        for (int base = 0; base < tasks; base += NWAVES * TU) {
            long long g[TU];
            bool act[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int i = base + wave + NWAVES * u;
                act[u] = i < tasks;
                g[u] = 0;
                if (act[u]) {
                    if (whole) {
                        g[u] = i;
                    } else {
                        g[u] = a.idx[i] >> (6 * l);
                        if (i > 0 && (a.idx[i - 1] >> (6 * l)) == g[u]) act[u] = false;   // adjacent duplicate parent
                    }
                }
            }
            double sv[TU];
            float mv[TU];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                sv[u] = 0.0;
                mv[u] = INFINITY;
                if (act[u]) {
                    const long long c = g[u] * 64 + lane;
                    sv[u] = tree_load_sum(a.t, l - 1, c);
                    mv[u] = tree_load_min(a.t, l - 1, c);
                }
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                if (act[u]) {
                    const double s = wave_sum_d(sv[u]);
                    const float m = wave_min(mv[u]);
                    if (lane == 0) {
                        a.t.sum[l][g[u]] = s;
                        a.t.mn[l][g[u]] = m;
                    }
                }
            }
        }
        __syncthreads();
    }
}


// N-step return assembly for batched (vectorised) rollouts on the GPU (gfx950).
//
// Input is a rollout segment laid out [T, E] (time major, E environments).  For
// every (t, e) the kernel produces the n-step transition the reference's agent
// assembles with a deque on the host (models/agent.py:89-123):
//   R     = sum_{i<k} gamma^i r[t+i]        k = steps until min(n, episode end, segment end)
//   gk    = gamma^k                         (per-sample bootstrap discount)
//   nxt   = t+k-1                           (row whose next-observation is the bootstrap state)
//   d     = done[t+k-1]                     (time-limit truncation does not set it)
// One thread per (t, e); consecutive threads walk consecutive e, so every load is coalesced.
#include "common.h"

struct NstepArgs {
    const float* reward;   // [T, E]
    const float* done;     // [T, E]
    const float* trunc;    // [T, E] or null
    float* R;              // [T, E]
    float* gk;             // [T, E]
    int* nxt;              // [T, E]
    float* d;              // [T, E]
    int T, E, n;
    float gamma;
};

__global__ __launch_bounds__(256) void nstep_kernel(const NstepArgs a) {
    const long long total = (long long)a.T * a.E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / a.E), e = (int)(i - (long long)t * a.E);
        float acc = 0.f, g = 1.f;
        int last = t;
        for (int k = 0; k < a.n && t + k < a.T; ++k) {
            const size_t o = (size_t)(t + k) * a.E + e;
            acc += g * a.reward[o];
            g *= a.gamma;
            last = t + k;
            const bool end = a.done[o] > 0.5f || (a.trunc != nullptr && a.trunc[o] > 0.5f);
            if (end) break;
        }
        a.R[i] = acc;
        a.gk[i] = g;
        a.nxt[i] = last;
        a.d[i] = a.done[(size_t)last * a.E + e] > 0.5f ? 1.f : 0.f;
    }
}

D4PG_API int d4pg_nstep(const NstepArgs* a, hipStream_t stream) {
    if (a->T < 1 || a->E < 1 || a->n < 1) return -1;
    const long long total = (long long)a->T * a->E;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(nstep_kernel, dim3(blocks), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

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

// ---------------------------------------------------------------------------
// Ornstein-Uhlenbeck exploration noise for batched GPU rollouts (utils/utils.py:9-34):
//   x <- x + theta (mu - x) + sigma N(0, 1);  out = clip(action + x, low, high)
// One thread per (env, action dim); normals from Philox-4x32 (Box-Muller), keyed by
// (seed, call counter) so replays of a captured graph still advance the stream via the
// device-resident counter.
struct OuArgs {
    float* x;                 // [E*A] noise state, updated in place
    const float* action;      // [E*A]
    float* out;               // [E*A]
    const float* sigma;       // [E] per-env sigma (schedule evaluated by the caller) or null
    unsigned long long* counter;   // device counter, incremented once per call
    unsigned long long seed;
    int n;                    // E*A
    int A;
    float theta, mu, sigma_const, low, high;
};

__global__ __launch_bounds__(256) void ou_noise_kernel(const OuArgs a) {
    const unsigned long long ctr = *a.counter;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) {
        const float sg = a.sigma != nullptr ? a.sigma[i / a.A] : a.sigma_const;
        const float x = ou_advance(a.x[i], (uint32_t)i, ctr, a.seed, a.theta, a.mu, sg);
        a.x[i] = x;
        a.out[i] = ou_apply(a.action[i], x, a.low, a.high);
    }
}
__global__ void ou_counter_kernel(unsigned long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

D4PG_API int d4pg_ou_noise(const OuArgs* a, hipStream_t stream) {
    if (a->n < 1 || a->A < 1) return -1;
    hipLaunchKernelGGL(ou_noise_kernel, dim3((a->n + 255) / 256), dim3(256), 0, stream, *a);
    hipLaunchKernelGGL(ou_counter_kernel, dim3(1), dim3(64), 0, stream, a->counter);
    return (int)hipGetLastError();
}

// The same noise step with a counter that somebody else advances (a device environment's step counter, which its
// own kernel increments once per environment step): one small launch less per rollout step.
D4PG_API int d4pg_ou_noise_shared_counter(const OuArgs* a, hipStream_t stream) {
    if (a->n < 1 || a->A < 1 || a->counter == nullptr) return -1;
    hipLaunchKernelGGL(ou_noise_kernel, dim3((a->n + 255) / 256), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

D4PG_SIZEOF(NstepArgs)
D4PG_SIZEOF(OuArgs)

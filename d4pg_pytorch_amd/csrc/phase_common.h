// Declarations shared by the row-local phase kernels of both precisions (csrc/fused.hip: bf16 MFMA
// operands; csrc/fused32.hip: fp32): where the batch rows come from, the problem dimensions, small LDS
// helpers, and the optimiser that the weight-gradient kernels apply to the tile they have reduced.
#pragma once
#include "common.h"
#include "replay_dev.h"

constexpr int AHEAD_STEP = 0, AHEAD_ROWS = 1;      // DevState::pad_ slots of the tag of a critic forward pass made ahead

// tanh through one v_exp_f32: 1 - 2 / (e^{2x} + 1); exact at +-inf, |error| < 2e-7
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }

// Batch rows read straight out of the gathered chunk buffers of csrc/replay.hip (packed rows
// [s | a | r | done | gamma^k | s' | weight | slot]), so that no per-update kernel has to
// copy them into separate input matrices.  Which chunk buffer and which batch of the chunk
// follow from the device step counter, exactly as in batch_unpack_kernel.
struct PackedIn {
    const float* in0;        // gathered chunk, even chunks  [world][K][Bl][row_ld + 4]
    const float* in1;        // gathered chunk, odd chunks
    DevState* st;
    long long step0;         // device step counter at the first batch of chunk 0
    int K, Bl, row_ld;
    int enabled;             // 0: the plain input arrays of the phase arguments are used
    int do_tick;             // the critic phase performs the per-update tick (block 0, thread 0)
    float beta1, beta2, per_b0, per_b1;
    int pad_;
    long long total_steps;
};

// Where batch row b of some input lives:  base + (b / Bl) * rank_stride + (b % Bl) * stride.
struct RowSrc {
    const float* base;
    long long stride, rank_stride;
    int Bl;
};
__device__ __forceinline__ RowSrc plain_src(const float* base, long long stride, int B) {
    return RowSrc{base, stride, 0, B > 0 ? B : 1};
}
// delta = 1: the batch of the NEXT update
__device__ __forceinline__ RowSrc packed_src(const PackedIn& p, int col0, int delta = 0) {
    const long long s = p.st->step_done - p.step0 + delta;
    // 32-bit division while the count fits (a 64-bit one is a ~150-instruction routine at the head of
    // the dependent chain of every phase kernel)
    int j, odd;
    if (((unsigned long long)s >> 32) == 0) {
        const unsigned q = (unsigned)s / (unsigned)p.K;
        j = (int)((unsigned)s - q * (unsigned)p.K);
        odd = (int)(q & 1u);
    } else {
        j = (int)(s % p.K);
        odd = (int)((s / p.K) & 1);
    }
    const float* chunk = odd ? p.in1 : p.in0;
    const long long ldp = p.row_ld + 4;
    return RowSrc{chunk + (long long)j * p.Bl * ldp + col0, ldp, (long long)p.K * p.Bl * ldp, p.Bl};
}
__device__ __forceinline__ const float* row_of(const RowSrc& r, int b) {
    const int q = b / r.Bl;
    return r.base + q * r.rank_stride + (b - q * r.Bl) * r.stride;
}

struct PhaseDims {
    int B;            // batch rows (multiple of 16*MT)
    int Bp;           // batch rows rounded up to a multiple of 32 (K of the weight-gradient GEMMs, stash_off)
    int S, A, H, NO;  // state, action, hidden, critic outputs (atoms or 1)
    int ldx_in;       // leading dimension of the fp32 input matrices (floats)
    int ldx;          // LDS leading dim of the input tile   = pad32(S + A) + LDS_PAD
    int ldh;          // LDS leading dim of the hidden tiles  = pad32(H) + LDS_PAD
    int ldd;          // LDS leading dim of the dlogits tile  = pad32(NO) + LDS_PAD
    int lda_;         // LDS leading dim of the d-action tile = pad32(A) + LDS_PAD
    int nop;          // fp32 logits row stride in LDS        = pad16(NO)
    int distributional;
    int use_ce;
    float v_min, v_max;
    float fixed_gamma;   // < 0: per-sample gamma
    float inv_B;         // 1 / global batch
    float min_value, max_value;   // scalar-critic target clamp
};

// zero `bytes` (multiple of 16) of LDS
__device__ __forceinline__ void zero_block(void* buf, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16)
        *reinterpret_cast<f32x4*>((char*)buf + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}
// flat copy of `bytes` (multiple of 16, both sides 16-byte aligned)
__device__ __forceinline__ void copy16(void* dst, const void* src, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16)
        *reinterpret_cast<f32x4*>((char*)dst + i) = *reinterpret_cast<const f32x4*>((const char*)src + i);
}
// flat copy like copy16 with up to four 16-byte loads of a thread in flight: a 16-row tile of a 400-wide layer (26 KB)
// in one round trip to memory for a workgroup of 512 threads instead of four
__device__ __forceinline__ void copy16x4(void* dst, const void* src, int bytes) {
    const int stride = blockDim.x * 16;
    for (int i0 = threadIdx.x * 16; i0 < bytes; i0 += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * stride;
            v[k] = i < bytes ? *reinterpret_cast<const f32x4*>((const char*)src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * stride;
            if (i < bytes) *reinterpret_cast<f32x4*>((char*)dst + i) = v[k];
        }
    }
}

static int check_packed(const PackedIn& p, const PhaseDims& d) {
    if (p.do_tick && p.st == nullptr) return -12;
    if (!p.enabled) return 0;
    if (p.in0 == nullptr || p.in1 == nullptr || p.st == nullptr) return -12;
    if (p.K < 1 || p.Bl < 1 || d.B % p.Bl || p.row_ld < 2 * d.S + d.A + 3) return -13;
    return 0;
}
template <typename K, typename Args>
static int launch_phase(K kernel, const Args& a, int grid, int nw, int lds, hipStream_t stream) {
    if (lds > 160 * 1024) return -8;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * nw), lds, stream, a);
    return (int)hipGetLastError();
}

// Optimiser state for the fused variant: when `p` is non-null every workgroup applies
// Adam + Polyak + the bf16 shadow refresh to the 32x32 tile of W it has just reduced
// (and the tn == 0 workgroups to the bias rows), so no separate optimiser launch is
// needed.  Only valid without data parallelism (the gradient must not need an all-reduce).
struct WgradAdam {
    float* p;            // flat online parameters (whole buffer)
    float* m;
    float* v;
    float* t;            // flat target parameters
    void* p_shadow;      // packed copies of the online / target weights in the precision of the pipeline
    void* t_shadow;
    DevState* st;
    float* metrics;              // finishing launch only
    const float* per_loss;
    const float* weights;
    const float* q_val;
    int B;
    float inv_B;
    float lr, beta1, beta2, eps, tau;
    int finish_step;
};

// optimiser state of one parameter, fetched at kernel entry (ahead of the stash loads and the
// MFMA loop, so its memory latency overlaps theirs)
struct AdamState {
    float m, v, p, t;
};
__device__ __forceinline__ AdamState adam_fetch(const WgradAdam& o, long long idx) {
    return AdamState{o.m[idx], o.v[idx], o.p[idx], o.t[idx]};
}
__device__ __forceinline__ float adam_apply(const WgradAdam& o, long long idx, const AdamState& s, float g,
                                            float step_size, float inv_sqrt_bc2, float& t_out) {
    float m = s.m, v = s.v, p = s.p, t = s.t;
    adam_polyak_math(m, v, p, t, g, o.beta1, o.beta2, o.eps, o.tau, step_size, inv_sqrt_bc2);
    o.m[idx] = m;
    o.v[idx] = v;
    o.p[idx] = p;
    o.t[idx] = t;
    t_out = t;
    return p;
}
__device__ __forceinline__ float adam_elem(const WgradAdam& o, long long idx, float g, float step_size,
                                           float inv_sqrt_bc2, float& t_out) {
    float m = o.m[idx], v = o.v[idx], p = o.p[idx], t = o.t[idx];
    adam_polyak_math(m, v, p, t, g, o.beta1, o.beta2, o.eps, o.tau, step_size, inv_sqrt_bc2);
    o.m[idx] = m;
    o.v[idx] = v;
    o.p[idx] = p;
    o.t[idx] = t;
    t_out = t;
    return p;
}

// End of an update: fold its losses into the metric sums and publish the step (step_done is read
// only by the first kernel of the next update).  Run by ONE EXTRA workgroup of the finishing launch,
// from its start: in a workgroup that also owns a tile this is ~1 us at the very end of the kernel.
template <int NWV>
__device__ __forceinline__ void wgrad_finish_role(const WgradAdam& o, float* cs) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // fold this update's losses into the metric sums and publish the step
    // (step_done is read only by the first kernel of the next update)
    __syncthreads();
    float vl = 0.f, pl = 0.f;
    for (int b = threadIdx.x; b < o.B; b += blockDim.x) {
        vl += o.per_loss[b] * (o.weights ? o.weights[b] : 1.f);
        pl -= o.q_val[b];
    }
    vl = wave_sum(vl);
    pl = wave_sum(pl);
    if (lane == 0) {
        cs[wave] = vl;
        cs[NWV + wave] = pl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (o.metrics != nullptr) {
            float svl = 0.f, spl = 0.f;
            for (int w = 0; w < NWV; ++w) {
                svl += cs[w];
                spl += cs[NWV + w];
            }
            o.metrics[MET_VALUE_LOSS] += svl * o.inv_B;
            o.metrics[MET_POLICY_LOSS] += spl * o.inv_B;
            o.metrics[MET_COUNT] += 1.f;
        }
        o.st->step_done = o.st->step_cur;
    }
}


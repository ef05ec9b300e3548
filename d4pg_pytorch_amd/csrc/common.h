// Shared declarations for the gfx950 (MI355X / CDNA4) learner kernels.
//
// Everything here is written for 64-lane wavefronts and the f32-input MFMA
// (v_mfma_f32_32x32x2_f32).  There is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define D4PG_API extern "C" __attribute__((visibility("default")))
// Size of an argument struct as compiled here; ops/hip.py compares it with its ctypes mirror at load time.
#define D4PG_SIZEOF(T) D4PG_API int d4pg_sizeof_##T() { return (int)sizeof(T); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// Device-resident learner state.  Kernel arguments are frozen when the update
// is captured into a HIP graph, so everything that changes from one update to
// the next (step counters, bias corrections, replay cursor, RNG stream) lives
// in this struct in device memory and is advanced by the kernels themselves.
//
// Race-free by construction: every field is written by exactly one thread of a
// kernel in which no thread reads that field.
//   step_done  : read by the first kernel of an update, written by the last.
//   step_cur   : written by the first kernel (= step_done + 1), read by the rest.
//   bc1, bc2   : Adam bias corrections for step_cur, written by the first kernel.
// ---------------------------------------------------------------------------
struct DevState {
    long long step_done;      // completed updates
    long long step_cur;       // 1-based index of the update in flight
    float bc1;                // 1 - beta1^step_cur
    float bc2;                // 1 - beta2^step_cur
    float per_beta;           // importance exponent used for the batch in flight
    float max_priority;       // running max of raw priorities (PER), >= 1
    long long replay_size;    // valid rows in the replay shard
    long long replay_head;    // next row to overwrite
    unsigned long long seed;  // RNG seed
    long long inserted_total; // rows ever inserted
    long long pad_[2];
};

// Device metric accumulator layout (floats).
enum { MET_VALUE_LOSS = 0, MET_POLICY_LOSS = 1, MET_COUNT = 2, MET_Q_MEAN = 3 };

// Offset (bf16 elements) of element (n, k) of a B operand in the packed "fragment order"
// used by the fused bf16 pipeline: [n / 16][k / 32][(k % 32) / 8][n % 16][k % 8], i.e. the
// 64 lanes x 8 elements that one wave feeds to v_mfma_f32_16x16x32_bf16 for output tile
// n / 16 and K-step k / 32 are one contiguous 1 KB block.  ks = K / 32 (K padded to 32).
__host__ __device__ __forceinline__ long long packed_off(int n, int k, int ks) {
    return (((((long long)(n >> 4) * ks + (k >> 5)) * 4 + ((k & 31) >> 3)) * 16 + (n & 15)) << 3) + (k & 7);
}

// Wave-wide reductions on the DPP lane-permute path of the VALU (quad_perm, row_ror):
// four VALU ops reduce each row of 16 lanes, three more combine the four rows through
// v_readlane.  (__shfl_xor compiles to ds_bpermute_b32: six dependent LDS-pipe round
// trips per reduction.)  Every lane returns the same value.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xb1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4e>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);   // row_ror 4
    v += dpp_mov<0x128>(v);   // row_ror 8
    return (row_lane(v, 0) + row_lane(v, 16)) + (row_lane(v, 32) + row_lane(v, 48));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_lane_d(double v, int l) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_mov_d<0xb1>(v);
    v += dpp_mov_d<0x4e>(v);
    v += dpp_mov_d<0x124>(v);
    v += dpp_mov_d<0x128>(v);
    return (row_lane_d(v, 0) + row_lane_d(v, 16)) + (row_lane_d(v, 32) + row_lane_d(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xb1>(v));
    v = fmaxf(v, dpp_mov<0x4e>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return fmaxf(fmaxf(row_lane(v, 0), row_lane(v, 16)), fmaxf(row_lane(v, 32), row_lane(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_mov<0xb1>(v));
    v = fminf(v, dpp_mov<0x4e>(v));
    v = fminf(v, dpp_mov<0x124>(v));
    v = fminf(v, dpp_mov<0x128>(v));
    return fminf(fminf(row_lane(v, 0), row_lane(v, 16)), fminf(row_lane(v, 32), row_lane(v, 48)));
}

// Philox-4x32-10 counter-based generator (one call = 4 x 32 random bits).
// Adam (torch.optim.Adam's arithmetic: models/d4pg/d4pg.py:55-56 of the reference) and the Polyak average of the
// target (d4pg.py:126-138) for one parameter.  ONE definition for the optimiser fused into the weight-gradient kernels
// and for the stand-alone kernel of the gradient-exchange mode, contraction off: the two round alike, so a world of
// one reproduces the fused path bit for bit.
__device__ __forceinline__ void adam_polyak_math(float& m, float& v, float& p, float& t, float g, float beta1, float beta2,
                                                 float eps, float tau, float step_size, float inv_sqrt_bc2) {
#pragma clang fp contract(off)
    m = beta1 * m + (1.f - beta1) * g;
    v = beta2 * v + (1.f - beta2) * g * g;
    p = p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    t = (1.f - tau) * t + tau * p;
}

// The step counter of a device environment (the Philox stream position of its resets and of the exploration noise) can
// be a PAIR of slots: a step reads one and one thread leaves counter + 1 in the other, which the kernels of the next
// step read.  No launch of its own, and no kernel reads a slot that a kernel of the same step writes.
__device__ __forceinline__ void next_counter(unsigned long long* next, unsigned long long ctr) {
    if (next != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *next = ctr + 1ull;
}

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One Ornstein-Uhlenbeck step of element i at stream position ctr (utils/utils.py:24-28 of the reference):
// x <- x + theta (mu - x) + sigma N(0, 1), N from Philox-4x32 + Box-Muller.  Shared by the stand-alone noise kernel
// (csrc/nstep.hip) and the policy kernel's epilogue (csrc/fused32.hip); contraction is off so that both round alike.
__device__ __forceinline__ float ou_advance(float x, uint32_t i, unsigned long long ctr, unsigned long long seed,
                                            float theta, float mu, float sigma) {
#pragma clang fp contract(off)
    uint32_t r[4];
    philox4x32(i, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0x0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = fmaxf((float)(r[0] >> 8) * (1.0f / 16777216.0f), 1e-7f);
    const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
    return x + theta * (mu - x) + sigma * nrm;
}
__device__ __forceinline__ float ou_apply(float action, float x, float low, float high) {
#pragma clang fp contract(off)
    return fminf(high, fmaxf(low, action + x));
}

// uniform double in [0,1) from 53 random bits
__device__ __forceinline__ double u01_from_bits(uint32_t hi, uint32_t lo) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)x * (1.0 / 9007199254740992.0);
}

// Advance the step counter / Adam bias corrections (single thread).  Reads
// step_done, writes step_cur, bc1, bc2, per_beta: the same contract as the
// tick embedded in the replay-sampling kernel.
static __device__ __forceinline__ void devstate_tick(DevState* st, float beta1, float beta2, float per_b0, float per_b1,
                              long long total_steps) {
    const long long t = st->step_done + 1;
    st->step_cur = t;
    st->bc1 = (float)(1.0 - pow((double)beta1, (double)t));
    st->bc2 = (float)(1.0 - pow((double)beta2, (double)t));
    double frac = (double)st->step_done / (double)(total_steps > 0 ? total_steps : 1);
    frac = frac < 0.0 ? 0.0 : (frac > 1.0 ? 1.0 : frac);
    st->per_beta = (float)(per_b0 + frac * (per_b1 - per_b0));
}


#define HIP_CHECK_RET(expr)                                   \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return (int)e_;                 \
    } while (0)

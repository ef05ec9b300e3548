// Fused, row-local bf16 pipeline of the learner update (gfx950).
//
// Every op of the update except the weight gradients and the optimiser is local to a
// batch row: MLP forwards, softmax, the categorical projection, the loss and the
// activation gradients (dX) never mix rows.  So instead of one launch per layer
// (23 dependent launches, csrc/gemm.hip) a workgroup takes a tile of 16*MT rows
// through WHOLE chains of layers with the activations resident in LDS:
//
//   critic_phase : target actor(s') -> target critic(s',a') -> online critic(s,a)
//                  -> softmax / projection / loss / dlogits -> dX through critic L3, L2
//   actor_phase  : actor(s) -> critic(s,pi(s)) -> expected Q / dlogits
//                  -> dX through critic L3, L2, L1(action columns) -> tanh' -> dX through actor L3, L2
//   wgrad        : all dW / db of one network (the only reduction over the batch) as ONE
//                  grouped launch on transposed bf16 stashes written by the phase kernels
//
// An update is 7 launches: sample, critic_phase, wgrad, adam, actor_phase, wgrad, adam.
//
// Arithmetic: bf16 MFMA (v_mfma_f32_16x16x32_bf16 in the phases, 32x32x16 in wgrad)
// with fp32 accumulation; master weights, biases, logits, losses, gradients and Adam
// state stay fp32.  Weights are read from bf16 "shadow" copies kept K-contiguous for
// each use (forward: W[o][i]; backward: the transpose), refreshed by the optimiser kernel.
// The kernel is bound by streaming those shadows from L2 (~1.4 MB per workgroup and
// phase), not by MFMA issue, and its time is independent of the batch size until
// B / (16*MT) exceeds the 256 CUs.
#include <cstddef>
#include "common.h"
#include "loss_dev.h"
#include "replay_dev.h"
#include "phase_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// This file is compiled five times into five objects (build.py) to keep the build short: the phase
// kernels are instantiated per hidden width (KSH = 13 / 4 / 8 K-steps, 0 = generic), and each width is
// one translation unit (-DD4PG_KSH_PART=<KSH>) that holds nothing but those kernels and their launch
// functions; the main unit (-DD4PG_KSH_PART=-1) holds the C API, the checks and the weight-gradient
// kernel and reaches the others through `extern template`.  Without the macro everything is in one unit
// (diagnostic and experiment builds).
#ifndef D4PG_KSH_PART
#define D4PG_KSH_PART 99
#endif
#define D4PG_MAIN_UNIT (D4PG_KSH_PART == 99 || D4PG_KSH_PART == -1)

// Diagnostic build (-DD4PG_STAMPS): workgroup 0 records the 100 MHz wall clock at every layer
// seam of the phase kernels; read back with d4pg_read_stamps (bench/stamps.py).
#ifdef D4PG_STAMPS
__device__ unsigned long long g_stamps[2][32];
#define STAMP(k, i)                                                                  \
    do {                                                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[k][i] = wall_clock64();    \
    } while (0)
#define STAMP_SYNC() __syncthreads()
#if D4PG_KSH_PART != 99
#error "the stamp build keeps everything in one translation unit (one g_stamps symbol)"
#endif
D4PG_API int d4pg_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define STAMP(k, i)
#define STAMP_SYNC()
#endif

// Waves per workgroup of the phase kernels.  Every wave keeps one weight tile (13 KB at H = 400) in
// flight, and a layer is bound by the latency of those loads: measured per update at batch 256,
// 8 waves 66.2 us, 12 waves 63.6 us, 16 waves 64.2 us (16 waves = 4 per SIMD leave 128 registers per
// wave, which the two prefetch slots of the layer chain overflow; 12 = 3 per SIMD leave 168; 13 would
// balance the 25 tiles of a 400-wide layer over two rounds but is capped at 128 registers too: 57.5 us).
// The kernels are instantiated for 12 and for 8 waves; the host picks 8 once the grid has more
// workgroups than the chip has CUs (batch 4096: 12 waves 107.8 us, 8 waves 93.5 us per update; capping
// the 8-wave variant at 128 registers so that two workgroups fit on a CU costs 14 spills: 97.8 us).
#ifndef D4PG_NW_FEW
#define D4PG_NW_FEW 12
#endif
constexpr int NW_FEW = D4PG_NW_FEW;     // grid <= 256 workgroups
constexpr int NW_MANY = 8;
constexpr int LDS_PAD = 8;     // bf16 elements appended to every LDS row (16 B: breaks the bank stride)

struct LayerW {
    const __bf16* W;     // packed shadow (common.h, packed_off): [ntiles][ks][64 lanes][8], zero padded
    const float* bias;   // fp32 [N]; a zero vector for bias-free (backward) layers, never null
    int ks;              // K / 32 (K-steps of one MFMA)
    int ntiles;          // output tiles of 16 columns
    int N;               // true number of outputs
    int pad_;
};

// ---------------------------------------------------------------------------------------
// One layer on a row tile held in LDS:  epi(row, col, sum_k A[row][k] * W[col][k]).
// A : LDS bf16 [16*MT][lda] (zero padded to K).  Wave w owns output tiles w, w+NW, ...
// The B fragment of (tile, k-step) is ONE contiguous 1 KB block of the packed shadow, so a
// wave-wide 16-byte load touches 8 whole cache lines (row-major weights touch 16 half
// lines per load and stream at half the rate: 5.1 vs 2.45 us per 400x416 layer and CU).
// C/D map of 16x16x32: col = lane & 15, row = 4 * (lane >> 4) + reg.
//
// Latency hiding across the layer chain: the weights of layer i+1 do not depend on the
// activations of layer i, so every layer function takes a `pre` callable that it invokes
// as soon as it has issued its own last weight load; `pre` issues the first loads of the
// NEXT layer (into registers owned by the kernel body), which then land while this
// layer's last MFMAs, its epilogue and the workgroup barrier run.
// ---------------------------------------------------------------------------------------
template <int KS>
struct TileB {
    bf16x8 b[KS];
    float bias;
};

// The bias of a tile is fetched TOGETHER with (ahead of) that tile's B fragments: vmcnt
// retires in issue order, so a bias load issued in the epilogue would sit behind the next
// tile's prefetch and drain it before every epilogue.
// Returns the RAW load (column clamped into range): any arithmetic on the value here would
// make the wave wait for the load right after issuing it.  use_bias() masks the padding
// columns when the epilogue consumes it.
__device__ __forceinline__ float load_bias(const LayerW& L, int nt, int c) {
    const int col = nt * 16 + c;
    const int cc = col < L.N ? col : 0;
    return L.bias[cc];      // never null: bias-free layers point at a zero vector
}
__device__ __forceinline__ float use_bias(const LayerW& L, int nt, int c, float raw) {
    return (nt * 16 + c) < L.N ? raw : 0.f;
}

// Returns a FRESH value: the registers of a tile that does not exist stay undefined (and are
// never consumed) instead of carrying the previous tile's contents, which would keep a whole
// register set alive across every conditional prefetch.
template <int KS>
__device__ __forceinline__ TileB<KS> fetch_tile(const LayerW& L, int nt, int lane) {
    TileB<KS> t;
    if (nt < L.ntiles) {   // wave-uniform
        t.bias = load_bias(L, nt, lane & 15);
        const __bf16* p = L.W + ((size_t)nt * KS * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) t.b[ks] = *reinterpret_cast<const bf16x8*>(p + ks * 512);
    }
    return t;
}

// Hidden-K layer, K-steps known at compile time (L.ks == KS).  `t` already holds the wave's
// first tile (issued by the previous layer's `pre`) and is refilled in place with the wave's
// further tiles right after the MFMAs that consumed it, ahead of the epilogue.  The eight
// waves of the workgroup keep ~100 KB of weight loads in flight between them, which is what
// one CU needs to stream from L2 at its full rate.
// `pre` is called at ONE place, before the tile loop: prefetch registers that are defined
// on several control-flow paths get merged with register copies, and a copy of a register
// with a load in flight waits for that load.
template <int MT, int KS, typename F>
__device__ __forceinline__ void ct_epilogue(const f32x4 (&acc)[MT], float bias, int nt, int lane, F&& epi) {
    const int col = nt * 16 + (lane & 15);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 v = acc[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias;
        epi(mt * 16 + 4 * (lane >> 4), col, v);      // 4 consecutive rows of one column
    }
}

template <int NW, int MT, int KS, typename F, typename P>
__device__ __forceinline__ void layer_ct(TileB<KS>& t, const __bf16* A, int lda, const LayerW& L, F&& epi, P&& pre) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kg = lane >> 4;
    pre();
    int nt = wave;
    if (nt >= L.ntiles) return;
    for (;;) {
        // two accumulators per row tile (even / odd K-steps): a dependent MFMA waits out the whole
        // pipeline of the one before it
        f32x4 acc[MT], acc1[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + c) * lda + ks * 32 + 8 * kg);
                if (ks & 1)
                    acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, t.b[ks], acc1[mt], 0, 0, 0);
                else
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, t.b[ks], acc[mt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] += acc1[mt];
        const float bias = use_bias(L, nt, c, t.bias);
        const int nn = nt + NW;
        if (nn < L.ntiles) {
            t = fetch_tile<KS>(L, nn, lane);
            ct_epilogue<MT, KS>(acc, bias, nt, lane, epi);
            nt = nn;
        } else {
            ct_epilogue<MT, KS>(acc, bias, nt, lane, epi);
            break;
        }
    }
}

// Generic layer (any K-steps, any tile count): no register staging beyond one k-step,
// no prefetch.  Used for hidden sizes without a compile-time variant and for narrow-K
// layers that exceed the register block of layer_rt.
template <int NW, int MT, typename F>
__device__ __forceinline__ void layer_gen(const __bf16* A, int lda, const LayerW& L, int first_tile, F&& epi) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, kg = lane >> 4;
    for (int nt = first_tile; nt < L.ntiles; nt += NW) {
        const __bf16* p = L.W + ((size_t)nt * L.ks * 64 + lane) * 8;
        const float bias = load_bias(L, nt, c);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ks = 0; ks < L.ks; ++ks) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(p + ks * 512);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + c) * lda + ks * 32 + 8 * kg);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[mt], 0, 0, 0);
            }
        }
        const int col = nt * 16 + c;
        const float bv = use_bias(L, nt, c, bias);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = acc[mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv;
            epi(mt * 16 + 4 * kg, col, v);
        }
    }
}

// Narrow-K layers (inputs, logits, actions: K <= 128): the wave's whole share of the
// layer (up to RT_T tiles x RT_K k-steps) is loaded in one burst, normally by the previous
// layer's `pre`, so the layer itself starts on its MFMAs.
constexpr int rt_tiles(int nw) { return (32 + nw - 1) / nw; }   // tiles per wave held in registers: covers >= 512 output columns
template <int NW, int RT_K>
struct RtB {
    static constexpr int RT_T = rt_tiles(NW);
    bf16x8 b[RT_T][RT_K];
    float bias[RT_T];
};

template <int NW, int RT_K>
__device__ __forceinline__ RtB<NW, RT_K> fetch_rt(const LayerW& L, int wave, int lane) {
    constexpr int RT_T = rt_tiles(NW);
    RtB<NW, RT_K> r;              // fresh value, see fetch_tile
    if (L.ks > RT_K) return r;    // layer_rt falls back to layer_gen
#pragma unroll
    for (int t = 0; t < RT_T; ++t) {
        const int nt = wave + t * NW;
        if (nt < L.ntiles) {
            r.bias[t] = load_bias(L, nt, lane & 15);
            const __bf16* p = L.W + ((size_t)nt * L.ks * 64 + lane) * 8;
#pragma unroll
            for (int ks = 0; ks < RT_K; ++ks)
                if (ks < L.ks) r.b[t][ks] = *reinterpret_cast<const bf16x8*>(p + ks * 512);
        }
    }
    return r;
}

template <int NW, int MT, int RT_K, typename F, typename P>
__device__ __forceinline__ void layer_rt(const RtB<NW, RT_K>& r, const __bf16* A, int lda, const LayerW& L, F&& epi, P&& pre) {
    constexpr int RT_T = rt_tiles(NW);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kg = lane >> 4;
    if (L.ks > RT_K) {
        pre();
        layer_gen<NW, MT>(A, lda, L, wave, epi);
        return;
    }
    // all MFMAs of the wave's register block first: the wait for `r` must come BEFORE the
    // prefetch is issued (the compiler cannot count loads behind the runtime predicates
    // and waits for everything in flight)
    f32x4 acc[RT_T][MT];
#pragma unroll
    for (int t = 0; t < RT_T; ++t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave + t * NW < L.ntiles) {
#pragma unroll
            for (int ks = 0; ks < RT_K; ++ks) {
                if (ks < L.ks) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + c) * lda + ks * 32 + 8 * kg);
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, r.b[t][ks], acc[t][mt], 0, 0, 0);
                    }
                }
            }
        }
    }
    float bv[RT_T];
#pragma unroll
    for (int t = 0; t < RT_T; ++t) bv[t] = use_bias(L, wave + t * NW, c, r.bias[t]);
    pre();
#pragma unroll
    for (int t = 0; t < RT_T; ++t) {
        const int nt = wave + t * NW;
        if (nt < L.ntiles) {
            const int col = nt * 16 + c;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v = acc[t][mt];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) v[rr] += bv[t];
                epi(mt * 16 + 4 * kg, col, v);
            }
        }
    }
    // more than RT_T tiles per wave (hidden sizes above 16 * RT_T * NW): stream the rest
    layer_gen<NW, MT>(A, lda, L, wave + RT_T * NW, epi);
}

// Per-kernel prefetch state + dispatch.  Two register sets (slots) alternate along the chain
// of hidden-K layers: the layer running out of slot X prefetches its successor into slot
// 1 - X (or into the narrow-K block).  Hidden-K layers run layer_ct when the kernel was
// instantiated for that K (KSH > 0), else the generic path.
template <int NW, int MT, int KSH, int KSN>
struct Chain {
    TileB<(KSH > 0 ? KSH : 1)> t0, t1;   // first tile of the next hidden-K layer (slot 0 / 1)
    RtB<NW, KSN> nr;                     // the next narrow-K layer (up to KSN K-steps)
    int wave, lane;
    __device__ __forceinline__ Chain() {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_h(const LayerW& L) {
        if constexpr (KSH > 0) {
            if constexpr (SLOT == 0)
                t0 = fetch_tile<KSH>(L, wave, lane);
            else
                t1 = fetch_tile<KSH>(L, wave, lane);
        }
    }
    __device__ __forceinline__ void prefetch_n(const LayerW& L) { nr = fetch_rt<NW, KSN>(L, wave, lane); }
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void hidden(const __bf16* A, int lda, const LayerW& L, F&& epi, P&& pre) {
        if constexpr (KSH > 0) {
            if constexpr (SLOT == 0)
                layer_ct<NW, MT, KSH>(t0, A, lda, L, epi, pre);
            else
                layer_ct<NW, MT, KSH>(t1, A, lda, L, epi, pre);
        } else {
            pre();
            layer_gen<NW, MT>(A, lda, L, wave, epi);
        }
    }
    template <typename F, typename P>
    __device__ __forceinline__ void narrow(const __bf16* A, int lda, const LayerW& L, F&& epi, P&& pre) {
        layer_rt<NW, MT, KSN>(nr, A, lda, L, epi, pre);
    }
};

// ---- epilogue helpers -------------------------------------------------------------------
// 4 consecutive rows (row0..row0+3) of one column -> LDS column write
__device__ __forceinline__ void put_lds4(__bf16* buf, int ld, int row0, int col, const f32x4& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(row0 + r) * ld + col] = (__bf16)v[r];
}
// Activation / gradient stashes for the weight-gradient kernel: element (feature col, batch row b)
// of a stash lives where the 32x32x16 MFMA of that kernel wants it, i.e. in operand-fragment
// order [col / 32][b / 16][lane = 32 * ((b % 16) / 8) + col % 32][b % 8], so that one wave-wide
// 16-byte load there is a contiguous 1 KB (a [feature][batch] matrix makes it 32 pieces of
// 32 bytes in 32 rows).  Kb = batch rows rounded up to 32; features are padded to 32.
__device__ __forceinline__ size_t stash_off(int col, int b, int Kb) {
    return ((size_t)(col >> 5) * (Kb >> 4) + (b >> 4)) * 512 + ((((b >> 3) & 1) << 5) + (col & 31)) * 8 + (b & 7);
}
// 4 consecutive batch rows (b0 % 4 == 0) of one feature: 8 contiguous bytes of the stash
__device__ __forceinline__ void put_stash4(__bf16* XT, int Kb, int col, int b0, const f32x4& v) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
    *reinterpret_cast<bf16x4*>(XT + stash_off(col, b0, Kb)) = o;
}

// The online critic's forward pass on the batch of update u + 1 needs nothing but that batch and
// the critic weights as update u leaves them, and those are final once the critic optimiser of
// update u has run.  When the batches are drawn ahead (PackedIn), B / M extra workgroups of the
// actor phase launch of update u make that pass (critic_ahead_role) and leave its row tiles in
// global memory; the critic phase of update u + 1 then starts from them and its dependent chain is
// three layers shorter.  A tag in DevState names the update and the rows the tiles belong to, so
// a critic phase that finds anything else (first update, another batch source, weights loaded by
// the host in between) simply makes the pass itself.
struct CriticAhead {
    int mode;        // 0: off; 1: only while the next batch is in the same chunk buffer; 2: always
    int pad_;
    __bf16* h1;      // [B][ldh] relu(layer 1) row tiles (LDS images)
    __bf16* h2;      // [B][ldh]
    float* lg;       // [B][nop] logits
};

// The online actor's forward pass a = pi(s) does not depend on the critic update, so it does
// not belong on the critical path critic phase -> critic optimiser -> actor phase: it runs in
// B / M EXTRA workgroups of the critic phase launch (on compute units that would idle) and
// hands its row tiles to the actor phase through global memory.
struct ActorFwd {
    LayerW af[3];          // online actor, forward
    const float* s;        // [B, ldx_in] state (action columns ignored) unless the batch is packed
    __bf16* sT;            // transposed stashes for the actor's weight gradients, [rows16][Bp]
    __bf16* h1T;
    __bf16* h2T;
    __bf16* g_xin;         // [B][ldx]  s | pi(s) | 0   (LDS tile images, row-major, read back by the actor phase)
    __bf16* g_h1;          // [B][ldh]
    __bf16* g_h2;          // [B][ldh]
    float* g_pi;           // [B][16 * ceil(A / 16)]  pi(s) in fp32 (for tanh')
};

struct CriticPhaseArgs {
    PhaseDims d;
    const float* sa;       // [B, ldx_in]  state | action
    const float* s2;       // [B, ldx_in]  next state (action columns ignored)
    const float* reward;
    const float* done;
    const float* gamma;
    const float* weights;  // may be null
    PackedIn pk;           // pk.enabled: rows come from the packed chunk buffers instead of the six arrays above
    LayerW ta[3];          // target actor, forward
    LayerW tc[3];          // target critic, forward
    LayerW cf[3];          // online critic, forward
    LayerW cb3, cb2;       // online critic, backward (transposed shadows) of layers 3 and 2
    // transposed bf16 stashes for wgrad, each [rows16][Bp]
    __bf16* saT;
    __bf16* h1T;
    __bf16* h2T;
    __bf16* dlT;
    __bf16* dh2T;
    __bf16* dh1T;
    float* per_loss;
    float* priority;
    ActorFwd fwd;
    CriticAhead ahead;     // tiles made by the previous update's actor phase launch (consumed when the tag matches)
};

// zero the K-padding tail of every row of an M-row bf16 LDS tile: columns [c0 & ~7, ld)
// (ld is a multiple of 8, so every chunk is an aligned 16-byte store; columns c0 & ~7 .. c0
// are rewritten by the producing layer's epilogue afterwards)
__device__ __forceinline__ void zero_tail(__bf16* buf, int ld, int M, int c0) {
    const int a0 = c0 & ~7, chunks = (ld - a0) >> 3;
    for (int i = threadIdx.x; i < M * chunks; i += blockDim.x) {
        const int r = i / chunks, c = i - r * chunks;
        *reinterpret_cast<f32x4*>(buf + r * ld + a0 + 8 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Load `ncols` fp32 columns of M batch rows into a bf16 LDS tile whose remaining columns (up to
// ld_dst) are written as zeros, so the tile needs no separate clearing pass; optionally
// also into a transposed global stash.
__device__ __forceinline__ void load_input(const RowSrc& src, int row0, int M, int ncols, __bf16* dst, int ld_dst,
                                           __bf16* stashT, int ldt) {
    for (int i = threadIdx.x; i < M * ld_dst; i += blockDim.x) {
        const int r = i / ld_dst, c = i - r * ld_dst;
        float v = 0.f;
        if (c < ncols) {
            v = row_of(src, row0 + r)[c];
            if (stashT != nullptr) stashT[stash_off(c, row0 + r, ldt)] = (__bf16)v;
        }
        dst[r * ld_dst + c] = (__bf16)v;
    }
}

// L2 warmer role.  The bf16 weights change every update, so the first touch of every weight line
// by an XCD misses its L2 and is served by the Infinity Cache at ~2.5x the latency; a phase is a chain
// of loads whose latency is what it costs.  Extra workgroups (slice `slice` of `nslices` on each XCD;
// consecutive workgroup ids go round the 8 XCDs) stream the layers in the order the phase uses them,
// a few microseconds ahead of the workgroups that compute, which then hit in L2.
// (One call per layer, layers passed by reference: an array of layer pointers indexed in a loop
// would force the kernel arguments out of scalar registers into memory for the whole kernel.)
__device__ __forceinline__ void warm_layer(const LayerW& L, int slice, int nslices, unsigned& acc) {
    const int chunks = L.ntiles * L.ks * 64;                 // 16-byte chunks of the packed shadow
    const int per = (chunks + nslices - 1) / nslices;
    const int lo = slice * per, hi = min(chunks, lo + per);
    const u32x4* p = reinterpret_cast<const u32x4*>(L.W);
#pragma unroll 4
    for (int i = lo + (int)threadIdx.x; i < hi; i += blockDim.x) {
        const u32x4 v = p[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
}

// Loss of the rows of one tile, one wave per row (SL atom slots per lane: 1 for <= 64 atoms).
template <int NW, int SL, int M>
__device__ __forceinline__ void critic_rows(const CriticPhaseArgs& a, int row0, int lane, int wave, const float* lg_t,
                                            const float* lg_o, __bf16* dlg, float* sp, float* sb, const float* rs) {
    const PhaseDims& d = a.d;
#pragma unroll
    for (int r = wave; r < M; r += NW) {      // M / NW rows per wave, unrolled: independent chains interleave
        const int b = row0 + r;
        const float rew = rs[r], nd = rs[M + r], g = rs[2 * M + r], scale = rs[3 * M + r];
        if (d.distributional) {
            float lt[SL], lo[SL], m[SL], dl[SL];
            // consecutive rows of a wave alternate between two scratch rows so that they can overlap
            float* scratch = ((((r - wave) / NW) & 1) ? sb : sp) + wave * ((d.NO + 63) & ~63);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                lt[s] = j < d.NO ? lg_t[r * d.nop + j] : -INFINITY;
                lo[s] = j < d.NO ? lg_o[r * d.nop + j] : -INFINITY;
            }
            if (r == 0) STAMP(0, 16);
            const float loss = d4pg_critic_row<SL, true>(lt, lo, rew, nd, g, d.NO, d.v_min, d.v_max,
                                                   d.use_ce, scratch, scratch, lane, m, dl);
            if (r == 0) STAMP(0, 17);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                if (j < d.NO) {
                    const __bf16 v = (__bf16)(dl[s] * scale);
                    dlg[r * d.ldd + j] = v;
                    a.dlT[stash_off(j, b, d.Bp)] = v;
                }
            }
            if (lane == 0) {
                a.per_loss[b] = loss * (scale / d.inv_B);      // weighted: the finishing role sums it as is
                a.priority[b] = fabsf(loss) + 1e-4f;
            }
            if (r == 0) STAMP(0, 18);
            if (r == NW) STAMP(0, 19);
        } else if (lane == 0) {
            float y = rew + nd * g * lg_t[r * d.nop];
            y = fminf(d.max_value, fmaxf(d.min_value, y));
            const float diff = lg_o[r * d.nop] - y;
            const float l = diff * diff;
            const __bf16 v = (__bf16)(2.f * diff * scale);
            dlg[r * d.ldd] = v;
            a.dlT[stash_off(0, b, d.Bp)] = v;
            a.per_loss[b] = l * (scale / d.inv_B);
            a.priority[b] = fabsf(l) + 1e-4f;
        }
    }
}

template <int NW, int MT, int KSH, int KSN>
__device__ __forceinline__ void actor_forward_role(const CriticPhaseArgs& a, int row0, char* smem) {
    constexpr int M = 16 * MT;
    const PhaseDims& d = a.d;
    const ActorFwd& f = a.fwd;
    Chain<NW, MT, KSH, KSN> ch;
    ch.prefetch_n(f.af[0]);
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    const int ldpi = 16 * ((d.A + 15) / 16);
    __bf16* xin = (__bf16*)carve(M * d.ldx * 2);    // [s | pi(s)]
    __bf16* bufA = (__bf16*)carve(M * d.ldh * 2);   // h1
    __bf16* bufB = (__bf16*)carve(M * d.ldh * 2);   // h2
    float* pi = (float*)carve(M * ldpi * 4);
    zero_tail(bufA, d.ldh, M, d.H);
    zero_tail(bufB, d.ldh, M, d.H);
    zero_block(pi, M * ldpi * 4);
    load_input(a.pk.enabled ? packed_src(a.pk, 0) : plain_src(f.s, d.ldx_in, d.B), row0, M, d.S, xin, d.ldx, f.sT,
               d.Bp);
    __syncthreads();
    ch.narrow(xin, d.ldx, f.af[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufA, d.ldh, r0, col, v);
        put_stash4(f.h1T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<0>(f.af[1]); });
    __syncthreads();
    copy16(f.g_h1 + (size_t)row0 * d.ldh, bufA, M * d.ldh * 2);
    ch.template hidden<0>(bufA, d.ldh, f.af[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufB, d.ldh, r0, col, v);
        put_stash4(f.h2T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<1>(f.af[2]); });
    __syncthreads();
    copy16(f.g_h2 + (size_t)row0 * d.ldh, bufB, M * d.ldh * 2);
    ch.template hidden<1>(bufB, d.ldh, f.af[2], [=](int r0, int col, f32x4 v) {
        if (col < d.A) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fast_tanh(v[r]);
                pi[(r0 + r) * ldpi + col] = v[r];
            }
            put_lds4(xin, d.ldx, r0, d.S + col, v);
        }
    }, [] {});
    __syncthreads();
    copy16(f.g_xin + (size_t)row0 * d.ldx, xin, M * d.ldx * 2);
    copy16(f.g_pi + (size_t)row0 * ldpi, pi, M * ldpi * 4);
}

// Everything after the logits of both critics: loss rows, then the backward pass through critic
// layers 3 and 2.  A function called once per branch of the kernel (and not code after the branches
// join): registers holding prefetched weights must not be merged where two control-flow paths meet.
template <int NW, int MT, int KSH, int KSN>
__device__ __forceinline__ void critic_tail(const CriticPhaseArgs& a, Chain<NW, MT, KSH, KSN>& ch, int row0, int lane,
                                            int wave, const float* lg_t, const float* lg_o, __bf16* dlg, float* sp,
                                            float* sb, const float* rs, __bf16* bufC, __bf16* bufD) {
    constexpr int M = 16 * MT;
    const PhaseDims& d = a.d;
    __syncthreads();
    STAMP(0, 9);
    // ---- loss: one wave per row ----
    if (d.distributional && d.NO <= 64)
        critic_rows<NW, 1, M>(a, row0, lane, wave, lg_t, lg_o, dlg, sp, sb, rs);
    else
        critic_rows<NW, 4, M>(a, row0, lane, wave, lg_t, lg_o, dlg, sp, sb, rs);
    __syncthreads();
    STAMP(0, 10);
    // ---- backward through critic layers 3 and 2 (activation gradients only) ----
    ch.narrow(dlg, d.ldd, a.cb3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufD[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufD, d.ldh, r0, col, v);       // in place: each element is read and written by one lane
        put_stash4(a.dh2T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<0>(a.cb2); });
    __syncthreads();
    STAMP(0, 11);
    ch.template hidden<0>(bufD, d.ldh, a.cb2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufC[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_stash4(a.dh1T, d.Bp, col, row0 + r0, v);
    }, [] {});
    STAMP_SYNC();
    STAMP(0, 12);
}

template <int NW, int MT, int KSH, int KSN>
__global__ __launch_bounds__(64 * NW) void critic_phase_kernel(const CriticPhaseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 16 * MT;
    const PhaseDims& d = a.d;
    const int row0 = blockIdx.x * M;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row0 >= 2 * d.B) {     // beyond both halves: L2 warmers
        const int sl = (blockIdx.x - 2 * (d.B / M)) >> 3, nsl = (gridDim.x - 2 * (d.B / M) + 7) >> 3;
        unsigned acc = 0;
        warm_layer(a.ta[0], sl, nsl, acc);
        warm_layer(a.fwd.af[0], sl, nsl, acc);
        warm_layer(a.ta[1], sl, nsl, acc);
        warm_layer(a.fwd.af[1], sl, nsl, acc);
        warm_layer(a.ta[2], sl, nsl, acc);
        warm_layer(a.fwd.af[2], sl, nsl, acc);
        warm_layer(a.tc[0], sl, nsl, acc);
        warm_layer(a.tc[1], sl, nsl, acc);
        warm_layer(a.tc[2], sl, nsl, acc);
        warm_layer(a.cf[0], sl, nsl, acc);
        warm_layer(a.cf[1], sl, nsl, acc);
        warm_layer(a.cf[2], sl, nsl, acc);
        warm_layer(a.cb3, sl, nsl, acc);
        warm_layer(a.cb2, sl, nsl, acc);
        asm volatile("" ::"v"(acc));      // the loads must be issued and returned; their values are not used
        return;
    }
    if (row0 >= d.B) {         // the second half of the grid: actor forward
        // The per-update tick (step counter, Adam bias corrections: two double-precision pow() on one
        // thread, microseconds) runs in a workgroup that finishes long before the critic role does.
        // It reads step_done (like packed_src) and writes step_cur / bc1 / bc2 / per_beta, which
        // nothing in this launch reads.
        if (a.pk.do_tick && row0 == d.B && threadIdx.x == 0)
            devstate_tick(a.pk.st, a.pk.beta1, a.pk.beta2, a.pk.per_b0, a.pk.per_b1, a.pk.total_steps);
        actor_forward_role<NW, MT, KSH, KSN>(a, row0 - d.B, smem);
        return;
    }
    STAMP(0, 31);
    Chain<NW, MT, KSH, KSN> ch;
    ch.prefetch_n(a.ta[0]);
    // ---- carve LDS (every offset a multiple of 16 bytes) ----
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    __bf16* xin = (__bf16*)carve(M * d.ldx * 2);    // [s | a]
    __bf16* xin2 = (__bf16*)carve(M * d.ldx * 2);   // [s' | a']
    __bf16* bufA = (__bf16*)carve(M * d.ldh * 2);
    __bf16* bufB = (__bf16*)carve(M * d.ldh * 2);
    __bf16* bufC = (__bf16*)carve(M * d.ldh * 2);   // h1 of the online critic (mask for the backward)
    __bf16* bufD = (__bf16*)carve(M * d.ldh * 2);   // h2 of the online critic, then dh2 in place
    __bf16* dlg = (__bf16*)carve(M * d.ldd * 2);
    float* lg_t = (float*)carve(M * d.nop * 4);
    float* lg_o = (float*)carve(M * d.nop * 4);
    float* sp = (float*)carve(NW * ((d.NO + 63) & ~63) * 4);      // per-wave projection rows (two sets)
    float* sb = (float*)carve(NW * ((d.NO + 63) & ~63) * 4);
    float* rs = (float*)carve(4 * M * 4);           // reward | 1 - done | gamma | weight / B
    // per-row scalars of the loss: fetched now (registers), parked in LDS behind the inputs
    const int SA = d.S + d.A;
    const RowSrc src_sa = a.pk.enabled ? packed_src(a.pk, 0) : plain_src(a.sa, d.ldx_in, d.B);
    const RowSrc src_s2 = a.pk.enabled ? packed_src(a.pk, SA + 3) : plain_src(a.s2, d.ldx_in, d.B);
    float rs_r = 0.f, rs_nd = 0.f, rs_g = 0.f, rs_w = 0.f;
    if (threadIdx.x < M) {
        const int b = row0 + threadIdx.x;
        if (a.pk.enabled) {
            const float* row = row_of(src_sa, b);
            rs_r = row[SA];
            rs_nd = 1.f - row[SA + 1];
            rs_g = d.fixed_gamma >= 0.f ? d.fixed_gamma : row[SA + 2];
            rs_w = row[a.pk.row_ld] * d.inv_B;
        } else {
            rs_r = a.reward[b];
            rs_nd = 1.f - a.done[b];
            rs_g = d.fixed_gamma >= 0.f ? d.fixed_gamma : a.gamma[b];
            rs_w = (a.weights ? a.weights[b] : 1.f) * d.inv_B;
        }
    }
    // K-padding columns are never written by an epilogue: clear them once (the matching
    // weight columns are zero, but LDS garbage could be NaN)
    zero_block(dlg, M * d.ldd * 2);
    zero_tail(bufA, d.ldh, M, d.H);
    zero_tail(bufB, d.ldh, M, d.H);
    // Was the online critic's forward pass on exactly these rows made ahead, after the last change
    // of the critic weights?  (wave-uniform; see CriticAhead)
    bool ahead_ok = false;
    if (a.ahead.mode != 0 && a.pk.enabled)
        ahead_ok = a.pk.st->pad_[AHEAD_STEP] == a.pk.st->step_done + 1 &&
                   a.pk.st->pad_[AHEAD_ROWS] == (long long)(uintptr_t)src_sa.base;
    STAMP(0, 0);
    if (ahead_ok) {
        copy16(bufC, a.ahead.h1 + (size_t)row0 * d.ldh, M * d.ldh * 2);
        copy16(bufD, a.ahead.h2 + (size_t)row0 * d.ldh, M * d.ldh * 2);
        copy16(lg_o, a.ahead.lg + (size_t)row0 * d.nop, M * d.nop * 4);
    } else {
        zero_tail(bufC, d.ldh, M, d.H);
        zero_tail(bufD, d.ldh, M, d.H);
        load_input(src_sa, row0, M, SA, xin, d.ldx, a.saT, d.Bp);
    }
    load_input(src_s2, row0, M, d.S, xin2, d.ldx, nullptr, 0);
    if (threadIdx.x < M) {
        rs[threadIdx.x] = rs_r;
        rs[M + threadIdx.x] = rs_nd;
        rs[2 * M + threadIdx.x] = rs_g;
        rs[3 * M + threadIdx.x] = rs_w;
    }
    __syncthreads();
    STAMP(0, 1);

    auto relu_to = [&](__bf16* buf) {
        return [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(buf, d.ldh, r0, col, v);
        };
    };
    // ---- target actor: s' -> a' (written into the action columns of xin2) ----
    ch.narrow(xin2, d.ldx, a.ta[0], relu_to(bufA), [&] { ch.template prefetch_h<0>(a.ta[1]); });
    __syncthreads();
    STAMP(0, 2);
    ch.template hidden<0>(bufA, d.ldh, a.ta[1], relu_to(bufB), [&] { ch.template prefetch_h<1>(a.ta[2]); });
    __syncthreads();
    STAMP(0, 3);
    ch.template hidden<1>(bufB, d.ldh, a.ta[2], [=](int r0, int col, f32x4 v) {
        if (col < d.A) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fast_tanh(v[r]);
            put_lds4(xin2, d.ldx, r0, d.S + col, v);
        }
    }, [&] { ch.prefetch_n(a.tc[0]); });
    __syncthreads();
    STAMP(0, 4);
    // ---- target critic: (s', a') -> logits_t ----
    ch.narrow(xin2, d.ldx, a.tc[0], relu_to(bufA), [&] { ch.template prefetch_h<0>(a.tc[1]); });
    __syncthreads();
    STAMP(0, 5);
    ch.template hidden<0>(bufA, d.ldh, a.tc[1], relu_to(bufB), [&] { ch.template prefetch_h<1>(a.tc[2]); });
    __syncthreads();
    STAMP(0, 6);
    auto to_lg_t = [=](int r0, int col, f32x4 v) {
        if (col < d.nop) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lg_t[(r0 + r) * d.nop + col] = v[r];
        }
    };
    // With the online critic's tiles made ahead, its three layers are skipped.  The first of them is
    // prefetched by the layer before; skipping is expressed as a layer without tiles, with the fields
    // selected one by one: a POINTER into the kernel arguments that depends on a run-time condition
    // makes the compiler keep a copy of all the arguments in scratch memory (so does a second call
    // site of the code after the branch, which is why that code is not duplicated into the branches).
    LayerW after_tc = a.cf[0];
    after_tc.ntiles = ahead_ok ? 0 : a.cf[0].ntiles;
    ch.template hidden<1>(bufB, d.ldh, a.tc[2], to_lg_t, [&] { ch.prefetch_n(after_tc); });
    if (!ahead_ok) {
        // ---- online critic: (s, a) -> logits, activations stashed (transposed) for wgrad ----
        ch.narrow(xin, d.ldx, a.cf[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(bufC, d.ldh, r0, col, v);
            put_stash4(a.h1T, d.Bp, col, row0 + r0, v);
        }, [&] { ch.template prefetch_h<0>(a.cf[1]); });
        __syncthreads();
        STAMP(0, 7);
        ch.template hidden<0>(bufC, d.ldh, a.cf[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(bufD, d.ldh, r0, col, v);
            put_stash4(a.h2T, d.Bp, col, row0 + r0, v);
        }, [&] { ch.template prefetch_h<1>(a.cf[2]); });
        __syncthreads();
        STAMP(0, 8);
        ch.template hidden<1>(bufD, d.ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
            if (col < d.nop) {
#pragma unroll
                for (int r = 0; r < 4; ++r) lg_o[(r0 + r) * d.nop + col] = v[r];
            }
        }, [] {});
    }
    // No prefetched registers are live where the two paths join: the first backward layer is fetched
    // here and lands while the loss rows are computed.
    ch.prefetch_n(a.cb3);
    critic_tail<NW, MT, KSH, KSN>(a, ch, row0, lane, wave, lg_t, lg_o, dlg, sp, sb, rs, bufC, bufD);
}

// ---------------------------------------------------------------------------------------
struct ActorPhaseArgs {
    PhaseDims d;
    // row tiles of the actor forward pass, produced by the extra workgroups of the critic phase
    const __bf16* g_xin;   // [B][ldx]  s | pi(s) | 0
    const __bf16* g_h1;    // [B][ldh]
    const __bf16* g_h2;    // [B][ldh]
    const float* g_pi;     // [B][16 * ceil(A / 16)]
    LayerW cf[3];          // online critic (already updated), forward
    LayerW cb3, cb2, cb1;  // critic backward: layers 3, 2 and the action columns of layer 1
    LayerW ab3, ab2;       // actor backward: layers 3 and 2
    __bf16* daT;           // transposed stashes [rows16][Bp]
    __bf16* dh2T;
    __bf16* dh1T;
    float* q_val;          // [B]
    // Priority write-back of the prioritised replay, run by ONE extra workgroup of this launch
    // (block B / M) while the others do the actor phase: it only needs the priorities the
    // critic phase wrote, and must be finished before the next update samples.  Forking it
    // onto a second stream inside the graph instead costs ~10 us per cross-queue dependency.
    TreeUpdateArgs tree;   // tree.n == 0: none
    // The online critic's forward pass for the NEXT update, in B / M more workgroups (CriticAhead)
    PackedIn pk;           // where the batches are (ahead.mode != 0 needs pk.enabled)
    __bf16* c_saT;         // the critic's input / activation stashes for its weight gradients
    __bf16* c_h1T;
    __bf16* c_h2T;
    CriticAhead ahead;
};

template <int NW, int MT, int KSH, int KSN>
__device__ __forceinline__ void critic_ahead_role(const ActorPhaseArgs& a, int row0, char* smem) {
    constexpr int M = 16 * MT;
    const PhaseDims& d = a.d;
    const long long s_next = a.pk.st->step_done - a.pk.step0 + 1;
    // mode 1: the next chunk buffer may still be receiving its rows (all-gather on another stream)
    if (a.ahead.mode == 1 && s_next % a.pk.K == 0) return;
    Chain<NW, MT, KSH, KSN> ch;
    ch.prefetch_n(a.cf[0]);
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    __bf16* xin = (__bf16*)carve(M * d.ldx * 2);    // [s | a]
    __bf16* bufC = (__bf16*)carve(M * d.ldh * 2);   // h1
    __bf16* bufD = (__bf16*)carve(M * d.ldh * 2);   // h2
    float* lg = (float*)carve(M * d.nop * 4);
    zero_tail(bufC, d.ldh, M, d.H);
    zero_tail(bufD, d.ldh, M, d.H);
    const RowSrc src = packed_src(a.pk, 0, 1);
    load_input(src, row0, M, d.S + d.A, xin, d.ldx, a.c_saT, d.Bp);
    __syncthreads();
    ch.narrow(xin, d.ldx, a.cf[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufC, d.ldh, r0, col, v);
        put_stash4(a.c_h1T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<0>(a.cf[1]); });
    __syncthreads();
    copy16(a.ahead.h1 + (size_t)row0 * d.ldh, bufC, M * d.ldh * 2);
    ch.template hidden<0>(bufC, d.ldh, a.cf[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufD, d.ldh, r0, col, v);
        put_stash4(a.c_h2T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<1>(a.cf[2]); });
    __syncthreads();
    copy16(a.ahead.h2 + (size_t)row0 * d.ldh, bufD, M * d.ldh * 2);
    ch.template hidden<1>(bufD, d.ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
        if (col < d.nop) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[(r0 + r) * d.nop + col] = v[r];
        }
    }, [] {});
    __syncthreads();
    copy16(a.ahead.lg + (size_t)row0 * d.nop, lg, M * d.nop * 4);
    // the tag: which update and which rows these tiles are for (read by the next launch)
    if (row0 == 0 && threadIdx.x == 0) {
        a.pk.st->pad_[AHEAD_STEP] = a.pk.st->step_done + 2;
        a.pk.st->pad_[AHEAD_ROWS] = (long long)(uintptr_t)src.base;
    }
}

template <int NW, int SL, int M>
__device__ __forceinline__ void actor_rows(const ActorPhaseArgs& a, int row0, int lane, int wave, const float* lg_o,
                                           __bf16* dlg) {
    const PhaseDims& d = a.d;
    for (int r = wave; r < M; r += NW) {
        const int b = row0 + r;
        if (d.distributional) {
            float e[SL], dl[SL];
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                e[s] = j < d.NO ? lg_o[r * d.nop + j] : -INFINITY;
            }
            const float qv = d4pg_actor_row<SL, true>(e, d.NO, d.v_min, d.v_max, lane, dl);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                if (j < d.NO) dlg[r * d.ldd + j] = (__bf16)(dl[s] * d.inv_B);
            }
            if (lane == 0) a.q_val[b] = qv;
        } else if (lane == 0) {
            dlg[r * d.ldd] = (__bf16)(-d.inv_B);
            a.q_val[b] = lg_o[r * d.nop];
        }
    }
}

template <int NW, int MT, int KSH, int KSN>
__global__ __launch_bounds__(64 * NW) void actor_phase_kernel(const ActorPhaseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 16 * MT;
    const PhaseDims& d = a.d;
    const int row0 = blockIdx.x * M;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_tree = a.tree.n > 0 ? (a.tree.parts > 1 ? a.tree.parts : 1) : 0;
    const int n_work = d.B / M + n_tree + (a.ahead.mode != 0 ? d.B / M : 0);
    if ((int)blockIdx.x >= n_work) {                // L2 warmers
        const int sl = (blockIdx.x - n_work) >> 3, nsl = (gridDim.x - n_work + 7) >> 3;
        unsigned acc = 0;
        warm_layer(a.cf[0], sl, nsl, acc);
        warm_layer(a.cf[1], sl, nsl, acc);
        warm_layer(a.cf[2], sl, nsl, acc);
        warm_layer(a.cb3, sl, nsl, acc);
        warm_layer(a.cb2, sl, nsl, acc);
        warm_layer(a.cb1, sl, nsl, acc);
        warm_layer(a.ab3, sl, nsl, acc);
        warm_layer(a.ab2, sl, nsl, acc);
        asm volatile("" ::"v"(acc));
        return;
    }
    if ((int)blockIdx.x >= d.B / M + n_tree) {      // the online critic's forward pass of the next update
        critic_ahead_role<NW, MT, KSH, KSN>(a, (blockIdx.x - (d.B / M + n_tree)) * M, smem);
        return;
    }
    if (row0 >= d.B) {         // the extra workgroup
        // The tree descriptor is read IN PLACE from the kernel-argument segment: its level arrays are indexed at
        // run time, and through the by-value parameter that makes the compiler copy the arguments to scratch
        // memory (and every phase role of this kernel pays for the registers and the copy).
        const TreeUpdateArgs& tree = *(const TreeUpdateArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                              offsetof(ActorPhaseArgs, tree));
        tree_update_body<NW>(tree, (int)blockIdx.x - d.B / M, smem, 4 * M * d.ldh * 2);      // (the hidden tiles alone)
        return;
    }
    STAMP(1, 31);
    Chain<NW, MT, KSH, KSN> ch;
    ch.prefetch_n(a.cf[0]);
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    const int ldpi = 16 * ((d.A + 15) / 16);
    __bf16* xin = (__bf16*)carve(M * d.ldx * 2);    // [s | pi(s)]
    __bf16* bufA = (__bf16*)carve(M * d.ldh * 2);   // h1 actor
    __bf16* bufB = (__bf16*)carve(M * d.ldh * 2);   // h2 actor, then dh2 actor in place
    __bf16* bufC = (__bf16*)carve(M * d.ldh * 2);   // h1 critic, then dh1 critic in place
    __bf16* bufD = (__bf16*)carve(M * d.ldh * 2);   // h2 critic, then dh2 critic in place
    __bf16* dlg = (__bf16*)carve(M * d.ldd * 2);
    __bf16* dab = (__bf16*)carve(M * d.lda_ * 2);
    float* lg_o = (float*)carve(M * d.nop * 4);
    float* pi = (float*)carve(M * ldpi * 4);
    // ---- the actor forward pass of this tile was made alongside the critic phase ----
    copy16(xin, a.g_xin + (size_t)row0 * d.ldx, M * d.ldx * 2);
    copy16(pi, a.g_pi + (size_t)row0 * ldpi, M * ldpi * 4);
    copy16(bufA, a.g_h1 + (size_t)row0 * d.ldh, M * d.ldh * 2);
    copy16(bufB, a.g_h2 + (size_t)row0 * d.ldh, M * d.ldh * 2);
    zero_block(dlg, M * (d.ldd + d.lda_) * 2);      // dlg and dab are adjacent
    zero_tail(bufC, d.ldh, M, d.H);
    zero_tail(bufD, d.ldh, M, d.H);
    __syncthreads();
    STAMP(1, 0);
    // ---- critic forward at (s, pi(s)) ----
    auto relu_to = [&](__bf16* buf) {
        return [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(buf, d.ldh, r0, col, v);
        };
    };
    ch.narrow(xin, d.ldx, a.cf[0], relu_to(bufC), [&] { ch.template prefetch_h<0>(a.cf[1]); });
    __syncthreads();
    STAMP(1, 1);
    ch.template hidden<0>(bufC, d.ldh, a.cf[1], relu_to(bufD), [&] { ch.template prefetch_h<1>(a.cf[2]); });
    __syncthreads();
    STAMP(1, 2);
    ch.template hidden<1>(bufD, d.ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
        if (col < d.nop) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lg_o[(r0 + r) * d.nop + col] = v[r];
        }
    }, [&] { ch.prefetch_n(a.cb3); });
    __syncthreads();
    STAMP(1, 3);
    // ---- actor loss: -mean Q ----
    if (d.distributional && d.NO <= 64)
        actor_rows<NW, 1, M>(a, row0, lane, wave, lg_o, dlg);
    else
        actor_rows<NW, 4, M>(a, row0, lane, wave, lg_o, dlg);
    __syncthreads();
    STAMP(1, 4);
    // ---- critic backward (dX only) down to the action ----
    ch.narrow(dlg, d.ldd, a.cb3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufD[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufD, d.ldh, r0, col, v);
    }, [&] { ch.template prefetch_h<0>(a.cb2); });
    __syncthreads();
    STAMP(1, 5);
    ch.template hidden<0>(bufD, d.ldh, a.cb2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufC[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufC, d.ldh, r0, col, v);
    }, [&] { ch.template prefetch_h<1>(a.cb1); });
    __syncthreads();
    STAMP(1, 6);
    ch.template hidden<1>(bufC, d.ldh, a.cb1, [=](int r0, int col, f32x4 v) {
        if (col < d.A) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = pi[(r0 + r) * ldpi + col];
                v[r] *= (1.f - y * y);
            }
            put_lds4(dab, d.lda_, r0, col, v);
            put_stash4(a.daT, d.Bp, col, row0 + r0, v);
        }
    }, [&] { ch.prefetch_n(a.ab3); });
    __syncthreads();
    STAMP(1, 7);
    // ---- actor backward ----
    ch.narrow(dab, d.lda_, a.ab3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufB[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufB, d.ldh, r0, col, v);
        put_stash4(a.dh2T, d.Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<0>(a.ab2); });
    __syncthreads();
    STAMP(1, 8);
    ch.template hidden<0>(bufB, d.ldh, a.ab2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)bufA[(r0 + r) * d.ldh + col] > 0.f ? v[r] : 0.f;
        put_stash4(a.dh1T, d.Bp, col, row0 + r0, v);
    }, [] {});
    STAMP_SYNC();
    STAMP(1, 9);
}

// ---------------------------------------------------------------------------------------
// host launchers
static int phase_lds_bytes_critic(const PhaseDims& d, int M, int nw) {
    auto al = [](int b) { return (b + 15) & ~15; };
    return 2 * al(M * d.ldx * 2) + 4 * al(M * d.ldh * 2) + al(M * d.ldd * 2) + 2 * al(M * d.nop * 4) +
           2 * al(nw * ((d.NO + 63) & ~63) * 4) + al(4 * M * 4);
}
static int phase_lds_bytes_actor(const PhaseDims& d, int M, int) {
    auto al = [](int b) { return (b + 15) & ~15; };
    return al(M * d.ldx * 2) + 4 * al(M * d.ldh * 2) + al(M * d.ldd * 2) + al(M * d.lda_ * 2) + al(M * d.nop * 4) +
           al(M * 16 * ((d.A + 15) / 16) * 4);
}
// waves per workgroup for a batch of B rows in tiles of M: see NW_FEW
static int phase_waves(const PhaseDims& d, int mt) { return (mt == 1 && d.B / 16 <= 128) ? NW_FEW : NW_MANY; }

static int check_dims(const PhaseDims& d, int M) {
    if (d.B < M || d.B % M) return -1;
    if (d.Bp < d.B || (d.Bp & 31)) return -2;
    if ((d.ldx & 7) || (d.ldh & 7) || (d.ldd & 7) || (d.lda_ & 7)) return -3;
    if (d.NO < 1 || d.NO > 256 || d.nop < d.NO || (d.nop & 15)) return -4;
    if (d.ldx < d.S + d.A || d.ldx_in < d.S + d.A) return -5;
    return 0;
}
static int max_k32(const PhaseDims& d) {
    int k = d.ldh;
    if (d.ldx > k) k = d.ldx;
    if (d.ldd > k) k = d.ldd;
    if (d.lda_ > k) k = d.lda_;
    return k - LDS_PAD;
}
static int check_layer(const LayerW& L, int k32_max) {
    if (L.W == nullptr || L.bias == nullptr || L.ntiles < 1 || L.ks < 1 || L.ks * 32 > k32_max) return -6;
    if (((uintptr_t)L.W) & 15) return -7;
    return 0;
}
static int max3i(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
// hidden-K layers run the compile-time path: their K-steps must be the kernel's
static int check_hidden(const LayerW& L, int ksh) { return L.ks == ksh ? 0 : -10; }

// L2 warmer workgroups per phase launch (warm_l2_role): a multiple of 8, one slice per XCD each
// Only while the grid leaves CUs idle: behind a grid that fills the chip the warmers are dispatched last
// and just lengthen the launch (batch 4096: 97.8 -> 115.6 us).
static int phase_waves(const PhaseDims& d, int mt);
static int n_warmers(const PhaseDims& d, int mt) {
    if (phase_waves(d, mt) != NW_FEW) return 0;
    static const int n = [] {
        const char* e = getenv("D4PG_L2_WARMERS");
        // measured per update (batch 256 / 2048 / 256 with PER): 0: 63.5 / 72.1 / 72.6 us, 8: 63.3 / 76.3 / 73.0,
        // 16: 60.5 / 70.5 / 69.5, 32: 59.6 / 69.9 / 69.2, 64: 59.7 / 69.7 / 69.5
        const int v = e ? atoi(e) : 32;
        return v < 0 ? 0 : (v > 256 ? 256 : (v + 7) & ~7);
    }();
    return n;
}

// All variants of one phase kernel for ONE hidden width (KSN: register block of the narrow-K layers, 2
// covers inputs / logits up to 64 wide).  Explicitly instantiated in the translation unit of that width.
#define D4PG_PHASE_LAUNCHER(NAME, KERNEL, ARGS)                                                              \
    template <int KSH>                                                                                        \
    int NAME(const ARGS* a, int mt, int nw, int ksn, int grid, int lds, hipStream_t stream) {                 \
        if (nw == NW_FEW) {                                                                                   \
            if (ksn <= 2) return launch_phase(KERNEL<NW_FEW, 1, KSH, 2>, *a, grid, NW_FEW, lds, stream);      \
            return launch_phase(KERNEL<NW_FEW, 1, KSH, 4>, *a, grid, NW_FEW, lds, stream);                    \
        }                                                                                                     \
        if (mt == 1) {                                                                                        \
            if (ksn <= 2) return launch_phase(KERNEL<NW_MANY, 1, KSH, 2>, *a, grid, NW_MANY, lds, stream);    \
            return launch_phase(KERNEL<NW_MANY, 1, KSH, 4>, *a, grid, NW_MANY, lds, stream);                  \
        }                                                                                                     \
        if (ksn <= 2) return launch_phase(KERNEL<NW_MANY, 2, KSH, 2>, *a, grid, NW_MANY, lds, stream);        \
        return launch_phase(KERNEL<NW_MANY, 2, KSH, 4>, *a, grid, NW_MANY, lds, stream);                      \
    }
D4PG_PHASE_LAUNCHER(launch_critic_ksh, critic_phase_kernel, CriticPhaseArgs)
D4PG_PHASE_LAUNCHER(launch_actor_ksh, actor_phase_kernel, ActorPhaseArgs)

#define D4PG_PHASE_INSTANCES(PREFIX, KSHV)                                                                      \
    PREFIX template int launch_critic_ksh<KSHV>(const CriticPhaseArgs*, int, int, int, int, int, hipStream_t);  \
    PREFIX template int launch_actor_ksh<KSHV>(const ActorPhaseArgs*, int, int, int, int, int, hipStream_t);
#if D4PG_KSH_PART == -1          // main unit: the widths live in their own objects
D4PG_PHASE_INSTANCES(extern, 13)
D4PG_PHASE_INSTANCES(extern, 4)
D4PG_PHASE_INSTANCES(extern, 8)
D4PG_PHASE_INSTANCES(extern, 0)
#elif D4PG_KSH_PART != 99        // the unit of one width
D4PG_PHASE_INSTANCES(, D4PG_KSH_PART)
#endif

#if D4PG_MAIN_UNIT
#define DISPATCH_PHASE(LAUNCHER, LDSFN)                                                     \
    const PhaseDims& d = a->d;                                                              \
    const int M = 16 * mt;                                                                  \
    if (mt != 1 && mt != 2) return -9;                                                      \
    int rc = check_dims(d, M);                                                              \
    if (rc) return rc;                                                                      \
    const int nw = phase_waves(d, mt);                                                      \
    const int lds = LDSFN(d, M, nw);                                                        \
    const int grid = d.B / M + extra_blocks;                                                \
    const int ksh = (d.ldh - LDS_PAD) / 32;                                                 \
    if (ksh == 13) return LAUNCHER<13>(a, mt, nw, ksn, grid, lds, stream);                  \
    if (ksh == 4) return LAUNCHER<4>(a, mt, nw, ksn, grid, lds, stream);                    \
    if (ksh == 8) return LAUNCHER<8>(a, mt, nw, ksn, grid, lds, stream);                    \
    return LAUNCHER<0>(a, mt, nw, ksn, grid, lds, stream);

D4PG_API int d4pg_critic_phase(const CriticPhaseArgs* a, int mt, hipStream_t stream) {
    const int kmax = max_k32(a->d);
    const int ksh_ = (a->d.ldh - LDS_PAD) / 32;
    for (int i = 0; i < 3; ++i)
        if (check_layer(a->ta[i], kmax) || check_layer(a->tc[i], kmax) || check_layer(a->cf[i], kmax)) return -6;
    if (check_layer(a->cb3, kmax) || check_layer(a->cb2, kmax)) return -6;
    for (int i = 1; i < 3; ++i)
        if (check_hidden(a->ta[i], ksh_) || check_hidden(a->tc[i], ksh_) || check_hidden(a->cf[i], ksh_)) return -10;
    if (check_hidden(a->cb2, ksh_)) return -10;
    for (int i = 0; i < 3; ++i)
        if (check_layer(a->fwd.af[i], kmax)) return -6;
    if (check_hidden(a->fwd.af[1], ksh_) || check_hidden(a->fwd.af[2], ksh_)) return -10;
    if (a->fwd.g_xin == nullptr || a->fwd.g_h1 == nullptr || a->fwd.g_h2 == nullptr || a->fwd.g_pi == nullptr ||
        a->fwd.sT == nullptr || a->fwd.h1T == nullptr || a->fwd.h2T == nullptr)
        return -14;
    const int ksn = max3i(a->cb3.ks, a->ta[0].ks, max3i(a->tc[0].ks, a->cf[0].ks, a->fwd.af[0].ks));
    // the actor forward pass rides along, then the L2 warmers
    const int extra_blocks = a->d.B / (mt == 2 ? 32 : 16) + n_warmers(a->d, mt);
    if (int prc = check_packed(a->pk, a->d)) return prc;
    if (a->ahead.mode != 0 && (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr)) return -15;
    DISPATCH_PHASE(launch_critic_ksh, phase_lds_bytes_critic)
}

D4PG_API int d4pg_actor_phase(const ActorPhaseArgs* a, int mt, hipStream_t stream) {
    const int kmax = max_k32(a->d);
    const int ksh_ = (a->d.ldh - LDS_PAD) / 32;
    for (int i = 0; i < 3; ++i)
        if (check_layer(a->cf[i], kmax)) return -6;
    if (check_layer(a->cb3, kmax) || check_layer(a->cb2, kmax) || check_layer(a->cb1, kmax) ||
        check_layer(a->ab3, kmax) || check_layer(a->ab2, kmax))
        return -6;
    for (int i = 1; i < 3; ++i)
        if (check_hidden(a->cf[i], ksh_)) return -10;
    if (a->g_xin == nullptr || a->g_h1 == nullptr || a->g_h2 == nullptr || a->g_pi == nullptr) return -14;
    if (check_hidden(a->cb2, ksh_) || check_hidden(a->cb1, ksh_) || check_hidden(a->ab2, ksh_)) return -10;
    const int ksn = max3i(a->cf[0].ks, a->cb3.ks, a->ab3.ks);
    if (a->tree.n < 0 || (a->tree.n > 0 && (a->tree.idx == nullptr || a->tree.prio == nullptr ||
                                            a->tree.t.L < 0 || a->tree.t.L > MAXL || a->tree.parts > 64 ||
                                            (a->tree.parts > 1 && (a->tree.t.L < 2 || a->tree.ticket == nullptr)))))
        return -11;
    if (a->ahead.mode != 0) {
        if (a->ahead.mode < 0 || a->ahead.mode > 2 || !a->pk.enabled) return -15;
        if (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr || a->c_saT == nullptr ||
            a->c_h1T == nullptr || a->c_h2T == nullptr)
            return -15;
        if (int prc = check_packed(a->pk, a->d)) return prc;
    }
    const int n_ahead = a->ahead.mode != 0 ? a->d.B / (mt == 2 ? 32 : 16) : 0;
    const int extra_blocks = (a->tree.n > 0 ? (a->tree.parts > 1 ? a->tree.parts : 1) : 0) + n_ahead + n_warmers(a->d, mt);
    DISPATCH_PHASE(launch_actor_ksh, phase_lds_bytes_actor)
}

D4PG_API int d4pg_phase_lds_bytes(const PhaseDims* d, int mt, int actor) {
    const int nw = phase_waves(*d, mt);
    return actor ? phase_lds_bytes_actor(*d, 16 * mt, nw) : phase_lds_bytes_critic(*d, 16 * mt, nw);
}

// ---------------------------------------------------------------------------------------
// Weight gradients:  dW[o][i] = sum_b dYT[o][b] * XT[i][b],  db[o] = sum_b dYT[o][b]
// Grouped (all three layers of a network in one launch), 32x32 tiles on
// v_mfma_f32_32x32x16_bf16, four waves split the batch (K) and combine through LDS.
// ---------------------------------------------------------------------------------------
struct WgradDesc {
    const __bf16* AT;   // dY stash, rowsA features x Bp batch rows in stash_off order
    const __bf16* BT;   // X stash,  rowsB features x Bp batch rows
    float* C;           // dW fp32 [M][ldc]
    float* rowsum;      // db fp32 [M] or null
    int M, N;           // true output rows / columns
    int rowsA, rowsB;   // allocated features of the stashes (multiples of 32)
    int ldc;
    int tiles_m, tiles_n, tile_base;
    // ---- fused optimiser (single-GPU path): where this matrix lives in the flat buffers ----
    long long w_off, b_off;        // offsets (floats) of W and bias in the flat parameter buffer
    long long sf_off, sb_off;      // forward / transposed shadow offsets (bf16 elements), -1: none
    int sf_ld, sb_ld, sb_c0, sb_c1;
};
constexpr int WG_MAX = 3;

struct WgradGroup {
    WgradDesc p[WG_MAX];
    int count, total_tiles;
    int Bp;             // K: batch rows rounded up to a multiple of 32
    int apply_only;     // gradient-exchange mode: the (all-reduced) gradients are in C / rowsum already, every tile only
                        // applies them: Adam, Polyak, shadows, with the tile-shaped stores of the fused optimiser
    WgradAdam opt;
    // Prioritised replay: the batch of the NEXT update is drawn by extra workgroups of the finishing
    // launch (one wave per row), where it sees the priorities this update has just written back (the
    // tree update rides in the actor phase launch before) and costs nothing on the critical path; a
    // sampling kernel at the start of every update is 6.8 us.  In device memory, not in the kernel
    // arguments: the tree descent indexes the level arrays of the descriptor at run time.
    const SampleArgs* smp;     // null: none
    int smp_rows;              // = smp->B (host copy: smp points to device memory)
    int tree_parts;            // > 0: that many more workgroups (the last of the grid) write the update's priorities
    const TreeUpdateArgs* tree;   // into the tree (the critic's launch: the chip has idle CUs there, and the tree is
                                  // complete before the actor phase starts); in device memory like smp
};

// NWV waves split the batch (K), then all of them finish the tile (sum of the NWV partial
// tiles, optimiser, shadows).  More waves = more stash loads in flight per CU: the loop is
// bound by the latency of those loads (the stashes were just written by other CUs), so NWV
// grows with the batch.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void wgrad_kernel(const WgradGroup g) {
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];
    float* red = wg_smem;                      // [NWV][16 accumulator registers][64 lanes]
    float* cs = wg_smem + NWV * 16 * 64;       // [NWV][32] partial row sums
    const int total = g.total_tiles;
    const int id = blockIdx.x;
    if (g.tree_parts > 0 && id >= (int)gridDim.x - g.tree_parts) {      // the last workgroups: priorities into the tree
        tree_update_body<NWV>(*g.tree, id - ((int)gridDim.x - g.tree_parts), wg_smem, NWV * (16 * 64 + 32) * 4);
        return;
    }
    if (id == total) {          // the extra workgroup of a finishing launch
        wgrad_finish_role<NWV>(g.opt, cs);
        return;
    }
    if (id > total) {           // the batch of the next update (SampleArgs): one wave per row
        const SampleArgs& sa = *g.smp;
        const int b = (id - total - 1) * NWV + (int)(threadIdx.x >> 6);
        const long long size = sa.st->replay_size;
        // stream position of the next update = updates completed once this one is: step_cur (step_done
        // is being advanced by the finish role of this very launch)
        if (b < sa.B && size > 0) sample_row(sa, b, sa.st->step_cur, size, threadIdx.x & 63);
        return;
    }
    STAMP(1, 20);
    const int xcd = id & 7, local = id >> 3;
    const int q = total >> 3, rem = total & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    int p = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX; ++i)
        if (i < g.count && tile >= g.p[i].tile_base) p = i;
    const WgradDesc& d = g.p[p];
    const int t = tile - d.tile_base;
    const int tm = t / d.tiles_n, tn = t - tm * d.tiles_n;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m = tm * 32 + r, n = tn * 32 + r;
    const bool m_ok = m < d.M, n_ok = n < d.N;
    // operand fragments of (tile, k-step) are contiguous 1 KB blocks of the stashes (stash_off)
    const int ksteps = g.Bp >> 4;              // k-steps of 16
    const __bf16* arow = d.AT + ((size_t)tm * ksteps * 64 + lane) * 8;
    const __bf16* brow = d.BT + ((size_t)tn * ksteps * 64 + lane) * 8;
    // Finishing the tile (sum of the NWV partial tiles, optimiser, shadows) is spread over ALL waves:
    // wave w owns accumulator registers w * EPT .. w * EPT + EPT - 1 of every lane, i.e. EPT rows of
    // column pn (register q of a 32x32 accumulator is row 8 * (q / 4) + 4 * h + q % 4).
    constexpr int EPT = 16 / NWV;
    const WgradAdam& o = g.opt;
    const bool fuse = o.p != nullptr;
    const int pn = tn * 32 + r;
    auto row_of_reg = [&](int q) { return tm * 32 + 8 * (q >> 2) + 4 * h + (q & 3); };
    AdamState ast[EPT];
    float step_size = 0.f, inv_sqrt_bc2 = 0.f;
    if (fuse) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int pm = row_of_reg(wave * EPT + i);
            if (pm < d.M && pn < d.N) ast[i] = adam_fetch(o, d.w_off + (long long)pm * d.ldc + pn);
        }
        step_size = o.st->bc1;
        inv_sqrt_bc2 = o.st->bc2;
    }
    const int per = (ksteps + NWV - 1) / NWV;
    const int k0 = wave * per, k1 = min(ksteps, k0 + per);
    STAMP(1, 21);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float asum = 0.f;
    const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    // the bias gradient (row sums of dY) is only needed from the first tile column
    const bool want_rowsum = d.rowsum != nullptr && tn == 0;
    const bool apply_only = g.apply_only != 0;      // (uniform over the launch)
#pragma unroll 8
    for (int ks = apply_only ? k1 : k0; ks < k1; ++ks) {
        bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + ks * 512);
        bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + ks * 512);
        av = m_ok ? av : zero;
        bv = n_ok ? bv : zero;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        if (want_rowsum) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asum += (float)av[j];
        }
    }
    STAMP(1, 22);
    if (!apply_only) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
        if (want_rowsum) {
            asum += __shfl_xor(asum, 32, 64);
            if (lane < 32) cs[wave * 32 + lane] = asum;
        }
        __syncthreads();
    }
    STAMP(1, 23);
    {
        if (fuse) {
            step_size = o.lr / step_size;
            inv_sqrt_bc2 = 1.f / sqrtf(inv_sqrt_bc2);
        }
        const int cn = pn;
        float pnew[EPT];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int reg = wave * EPT + i;
            float gv = 0.f;
            const int cm = row_of_reg(reg);
            if (!apply_only) {
#pragma unroll
                for (int w = 0; w < NWV; ++w) gv += red[(w * 16 + reg) * 64 + lane];
            } else if (cm < d.M && cn < d.N) {
                gv = d.C[(size_t)cm * d.ldc + cn];
            }
            pnew[i] = 0.f;
            if (cm < d.M && cn < d.N) {
                // the gradient itself goes to memory only when something outside reads it (the all-reduce of
                // the gradient-exchange mode); with the optimiser fused below nothing does
                if (!fuse) d.C[(size_t)cm * d.ldc + cn] = gv;
                if (fuse) {
                    float tv;
                    const float pv = adam_apply(o, d.w_off + (long long)cm * d.ldc + cn, ast[i], gv, step_size,
                                                inv_sqrt_bc2, tv);
                    pnew[i] = pv;
                    if (d.sf_off >= 0) {
                        const long long so = d.sf_off + packed_off(cm, cn, d.sf_ld >> 5);
                        ((__bf16*)o.p_shadow)[so] = (__bf16)pv;
                        ((__bf16*)o.t_shadow)[so] = (__bf16)tv;
                    }
                }
            }
        }
        if (fuse && d.sb_off >= 0 && cn >= d.sb_c0 && cn < d.sb_c1 && cn < d.N) {
            // transposed shadow: output rows are the k of a packed row
            if constexpr (EPT == 4) {
                // 4 consecutive output rows (multiple of 4) of one input column: one 8-byte store
                const int cm0 = row_of_reg(wave * EPT);
                __bf16* dst = (__bf16*)o.p_shadow + d.sb_off + packed_off(cn - d.sb_c0, cm0, d.sb_ld >> 5);
                if (cm0 + 3 < d.M) {
                    bf16x4 pk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk[i] = (__bf16)pnew[i];
                    *reinterpret_cast<bf16x4*>(dst) = pk;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (cm0 + i < d.M) dst[i] = (__bf16)pnew[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int cm = row_of_reg(wave * EPT + i);
                    if (cm < d.M)
                        ((__bf16*)o.p_shadow)[d.sb_off + packed_off(cn - d.sb_c0, cm, d.sb_ld >> 5)] = (__bf16)pnew[i];
                }
            }
        }
        if (d.rowsum != nullptr && tn == 0 && threadIdx.x < 32) {
            const int cm = tm * 32 + threadIdx.x;
            if (cm < d.M) {
                float gb = 0.f;
                if (apply_only) {
                    gb = d.rowsum[cm];
                } else {
#pragma unroll
                    for (int w = 0; w < NWV; ++w) gb += cs[w * 32 + threadIdx.x];
                }
                if (!fuse) d.rowsum[cm] = gb;
                if (fuse) {
                    float tv;
                    adam_elem(o, d.b_off + cm, gb, step_size, inv_sqrt_bc2, tv);
                }
            }
        }
        STAMP(1, 24);
    }
}

template <typename K>
static int launch_wgrad(K kernel, int nwv, int grid, const WgradGroup& g, hipStream_t stream) {
    const int lds = nwv * (16 * 64 + 32) * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
    }
    int extra = (g.opt.p != nullptr && g.opt.finish_step) ? 1 : 0;       // wgrad_finish_role
    if (g.smp != nullptr) {
        if (!extra || g.smp_rows < 1) return -5;          // only behind the finish workgroup
        extra += (g.smp_rows + nwv - 1) / nwv;
    }
    if (g.tree_parts < 0 || g.tree_parts > 64 || (g.tree_parts > 0 && g.tree == nullptr)) return -6;
    extra += g.tree_parts;
    hipLaunchKernelGGL(kernel, dim3(grid + extra), dim3(64 * nwv), lds, stream, g);
    return (int)hipGetLastError();
}

D4PG_API int d4pg_wgrad(WgradGroup* g, hipStream_t stream) {
    if (g->count < 1 || g->count > WG_MAX || g->Bp < 32 || (g->Bp & 31)) return -1;
    int base = 0;
    for (int i = 0; i < g->count; ++i) {
        WgradDesc& d = g->p[i];
        if (d.M < 1 || d.N < 1 || d.ldc < d.N) return -2;
        // the stashes hold whole 32-feature blocks (stash_off): every tile row must exist
        if ((d.rowsA & 31) || (d.rowsB & 31) || d.rowsA < d.M || d.rowsB < d.N) return -2;
        if (d.AT == nullptr || d.BT == nullptr || d.C == nullptr) return -3;
        if ((((uintptr_t)d.AT) | ((uintptr_t)d.BT)) & 15) return -4;
        d.tiles_m = (d.M + 31) / 32;
        d.tiles_n = (d.N + 31) / 32;
        d.tile_base = base;
        base += d.tiles_m * d.tiles_n;
    }
    g->total_tiles = base;
    if (g->apply_only && g->opt.p == nullptr) return -7;
    for (int i = 0; g->apply_only && i < g->count; ++i)
        if (g->p[i].rowsum == nullptr) return -7;
    // waves per workgroup (they split the batch): one k-step of 16 batch rows per wave up to 16 waves;
    // measured on MI355X (us per update, batch 256 / 1024 / 2048): 4 waves 68.0 / 79.2 / 91.8,
    // 8 waves 66.5 / 73.5 / 79.3, 16 waves 66.2 / 70.8 / 74.8.  D4PG_WGRAD_WAVES overrides (tuning).
    int nwv = g->Bp >= 256 ? 16 : (g->Bp >= 128 ? 8 : 4);
    static const int forced_waves = [] { const char* e = getenv("D4PG_WGRAD_WAVES"); return e ? atoi(e) : 0; }();      // (A/B runs)
    if (forced_waves) nwv = forced_waves;
    if (nwv == 16) return launch_wgrad(wgrad_kernel<16>, 16, base, *g, stream);
    if (nwv == 8) return launch_wgrad(wgrad_kernel<8>, 8, base, *g, stream);
    return launch_wgrad(wgrad_kernel<4>, 4, base, *g, stream);
}
D4PG_SIZEOF(PackedIn)
D4PG_SIZEOF(PhaseDims)
D4PG_SIZEOF(WgradAdam)
D4PG_SIZEOF(LayerW)
D4PG_SIZEOF(CriticAhead)
D4PG_SIZEOF(ActorFwd)
D4PG_SIZEOF(CriticPhaseArgs)
D4PG_SIZEOF(ActorPhaseArgs)
D4PG_SIZEOF(WgradDesc)
D4PG_SIZEOF(WgradGroup)
#endif   // D4PG_MAIN_UNIT

// Fused, row-local learner update in reference precision (fp32 operands, fp32 accumulation; gfx950).
//
// The same decomposition as csrc/fused.hip — everything but the weight gradients is local to a batch
// row, so a workgroup takes a tile of rows through WHOLE chains of layers with the activations in LDS,
// and an update is four launches (critic phase, weight gradients + optimiser, actor phase, weight
// gradients + optimiser) — but every product is an exact f32 multiply-add on the matrix cores:
//
//   phases : v_mfma_f32_4x4x1_16b_f32 on tiles of 4 * MT rows (csrc/mlp32_dev.h; one instance per hidden-width
//            class, KSH > 0) or v_mfma_f32_16x16x4_f32 on tiles of 8 / 16 rows (csrc/mlp16_dev.h; widths at run
//            time, KSH = 0; KSN = 1: its lean build, two workgroups per CU for grids of two workgroups per CU
//            and more); weights streamed from fp32 copies packed in the fragment order of the path;
//            activations, logits, losses, dX in fp32 in LDS
//   wgrad  : v_mfma_f32_32x32x2_f32 on fp32 stashes written by the phases in operand-fragment order,
//            Adam + Polyak + refresh of the packed copies fused into the tile epilogue
//
// The reference computes in fp32 throughout (models/d4pg/d4pg.py:73-77 `.float()`,
// models/d4pg/networks.py:23-25 fp32 nn.Linear); this pipeline tracks an eager fp32 PyTorch update to
// rounding (tests/test_hip_f32.py).  A layer costs what it takes to stream its fp32 weights through the
// one CU that owns the row tile (a CU's vector cache delivers ~130 GB/s: 5.3 us for 400 x 400, twice the
// bf16 copies of csrc/fused.hip); the MFMA work of a 4-row tile (2.3 us) hides under it.
#include "common.h"
#include "loss_dev.h"
#include "replay_dev.h"
#include "phase_common.h"
#include "mlp32_dev.h"
#include "mlp16_dev.h"

// Compiled once per hidden width (build.py, -DD4PG_KSH_PART=<KSH>), like csrc/fused.hip: the unit of one
// width holds the phase kernels for that width, the main unit (-1) the C API and the weight gradients.
#ifndef D4PG_KSH_PART
#define D4PG_KSH_PART 99
#endif
#define D4PG_MAIN_UNIT (D4PG_KSH_PART == 99 || D4PG_KSH_PART == -1)

// Diagnostic build (-DD4PG_STAMPS, single translation unit): workgroup 0 records the 100 MHz wall clock at
// every layer seam of the phase kernels; read back with d4pg_read_stamps32 (bench/stamps.py --dtype fp32).
#ifdef D4PG_STAMPS
__device__ unsigned long long g_stamps32[2][32];
#define STAMP(k, i)                                                                    \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps32[k][i] = wall_clock64();    \
    } while (0)
// earliest start / latest end over ALL workgroups of a role (slots 20.. of the kernel's row)
#define STAMP_SPAN(k, i, is_end)                                                                       \
    do {                                                                                               \
        if (threadIdx.x == 0) {                                                                        \
            if (is_end) atomicMax(&g_stamps32[k][i], wall_clock64());                                  \
            else atomicMin(&g_stamps32[k][i], wall_clock64());                                         \
        }                                                                                              \
    } while (0)
#if D4PG_KSH_PART != 99
#error "the stamp build keeps everything in one translation unit (one g_stamps32 symbol)"
#endif
D4PG_API int d4pg_read_stamps32(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps32), sizeof(g_stamps32));
}
D4PG_API int d4pg_reset_stamps32() {      // re-arm the running minima (slots 20, 22) and maxima
    unsigned long long h[2][32];
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 32; ++i) h[k][i] = (i == 20 || i == 22) ? ~0ull : 0ull;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps32), h, sizeof(h));
}
#else
#define STAMP(k, i)
#define STAMP_SPAN(k, i, is_end)
#endif

// The argument struct of a kernel, read in place from the kernel-argument segment.  A by-value struct parameter
// is an addressable local object for the compiler; whether it manages to turn every access back into a scalar
// load from the argument segment depends on the size of the kernel (variants of these kernels have ended up
// copying all their arguments to scratch memory after small changes elsewhere, which also makes every buffer
// descriptor built from them divergent).  Reading through the segment pointer leaves nothing to copy.  The
// struct is the kernel's only parameter, so it sits at offset 0 of the segment.
template <typename Args>
__device__ __forceinline__ const Args& kernel_args() {
    return *(const Args*)__builtin_amdgcn_kernarg_segment_ptr();      // (C-style cast: drops the address space)
}

constexpr int NW32 = 8;          // waves per phase workgroup: one unit (column tile or K-slice) each
constexpr int LDS_PAD32 = 4;     // floats appended to every LDS row: the 4 rows of one wide read hit disjoint banks

// chunks per K-slice of a single-tile layer whose K is the hidden width (KSH chunks dealt to NW32 waves)
constexpr int split_nch(int ksh) { return (ksh + NW32 - 1) / NW32; }

// Prefetch state along the chain of layers: two register sets alternate, the layer running out of
// slot X has the first chunk of its successor fetched into slot 1 - X.
template <int MT, int KSH, int KSN>
struct Chain32 {
    Pre32 t0, t1;
    float* red;
    int wave, lane;
    __device__ __forceinline__ Chain32(float* red_) : red(red_) {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    }
    // prefetch of a layer by kind (h: hidden x hidden, s: K-sliced output layer, n: narrow K): the first
    // chunks of the wave's unit into slot SLOT
    template <int SLOT>
    __device__ __forceinline__ void prefetch_h(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_first<KSH>(L, wave, lane);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_s(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_first<split_nch(KSH)>(L, wave, lane);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_n(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_first_rt(L, L.kg, wave, lane);
    }
    // the same for a layer that may be skipped (`run` false: nothing is loaded, the registers stay dead)
    template <int SLOT>
    __device__ __forceinline__ void prefetch_n_if(const LayerW32& L, bool run) {
        (SLOT == 0 ? t0 : t1) = fetch_first_rt(L, run ? L.kg : 0, wave, lane);
    }
    // K = hidden width, N = hidden width: KSH chunks per column tile
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void hidden(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre, int lst = -1) {
        layer32<NW32, MT, KSH, false>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre, lst);
    }
    // K = hidden width, N <= 64 (actions, atoms): K-slices of one column tile
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void split(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre) {
        layer32<NW32, MT, split_nch(KSH), false>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre);
    }
    // K <= 32 * KSN (inputs, actions, atoms), N = hidden width
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void narrow(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre, int lst = -1) {
        layer32<NW32, MT, KSN, true>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre, lst);
    }
};

// KSH == 0: the 16x16x4 path of csrc/mlp16_dev.h (any widths, known at run time; tiles of 8 or 16 rows, MT = 2 / 4).
// The same interface, so the roles below are written once.
//
// KSN selects one of two builds of that path.  0: four column tiles per wave and pass over K, ~170 registers, every
// LDS tile in its own buffer: one workgroup per CU.  1 ("lean"): two column tiles per pass, at most 128 registers,
// LDS tiles that are never live together share a buffer (lean_lds): TWO workgroups per CU, for grids of at least two
// workgroups per CU.  A phase is a chain of layers; the 400 x 400 ones run at the fp32 matrix rate, everything between
// them (row loads, narrow layers, K-slice reductions, the loss rows) leaves the matrix pipe idle, and only a second
// workgroup on the same CU can use it meanwhile.
constexpr bool lean_lds(int ksh, int ksn) { return ksh == 0 && ksn == 1; }
template <int MT, int KSN>
struct Chain32<MT, 0, KSN> {
    static_assert(MT == 2 || MT == 4, "the 16x16x4 path takes tiles of 8 or 16 rows");
    static constexpr int TP = KSN == 1 ? TPW16_LEAN : TPW16;
    Pre16<TP> t0, t1;
    float* red;
    int wave, lane;
    __device__ __forceinline__ Chain32(float* red_) : red(red_) {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_h(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_pass16<NW32, TP>(L, wave, lane, 0, true);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_s(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_pass16<NW32, TP>(L, wave, lane, 0, true);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_n(const LayerW32& L) {
        (SLOT == 0 ? t0 : t1) = fetch_pass16<NW32, TP>(L, wave, lane, 0, true);
    }
    template <int SLOT>
    __device__ __forceinline__ void prefetch_n_if(const LayerW32& L, bool run) {
        (SLOT == 0 ? t0 : t1) = fetch_pass16<NW32, TP>(L, wave, lane, 0, run);
    }
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void hidden(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre, int = -1) {
        layer16<NW32, 4 * MT, TP>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre);
    }
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void split(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre) {
        layer16<NW32, 4 * MT, TP>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre);
    }
    template <int SLOT, typename F, typename P>
    __device__ __forceinline__ void narrow(const float* A, int lda, const LayerW32& L, F&& epi, P&& pre, int = -1) {
        layer16<NW32, 4 * MT, TP>(SLOT == 0 ? t0 : t1, A, lda, L, red, epi, pre);
    }
};

// floats of LDS for the K-slice partial sums of a workgroup
__host__ __device__ constexpr int red_floats32(int mt, bool wide) { return wide ? (NW32 - 1) * 256 : (NW32 - 1) * mt * 256; }

// ---- LDS / stash helpers ------------------------------------------------------------------------
// 4 consecutive rows of one column -> LDS (consecutive lanes = consecutive columns: no bank conflicts)
__device__ __forceinline__ void put_lds4(float* buf, int ld, int row0, int col, const f32x4& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(row0 + r) * ld + col] = v[r];
}
// Activation / gradient stashes for the weight-gradient kernel: element (feature col, batch row b) lives
// where v_mfma_f32_32x32x2_f32 wants it when 16-byte loads feed four MFMAs each, i.e.
// [col / 32][b / 8][lane = 32 * ((b % 8) / 4) + col % 32][b % 4]: one wave-wide load is a contiguous 1 KB
// and the 4 rows a lane holds after an MFMA of the phases are one 16-byte store.  (The k order inside a
// group of 8 rows is permuted the same way in both operands, which a sum over k does not see.)
__device__ __forceinline__ size_t stash32_off(int col, int b, int Kb) {
    return ((size_t)(col >> 5) * (Kb >> 3) + (b >> 3)) * 256 + ((((b >> 2) & 1) << 5) + (col & 31)) * 4 + (b & 3);
}
__device__ __forceinline__ void put_stash4(float* XT, int Kb, int col, int b0, const f32x4& v) {
    *reinterpret_cast<f32x4*>(XT + stash32_off(col, b0, Kb)) = v;
}
// zero columns [c0 & ~3, ld) of every row of an M-row LDS tile (K padding; an epilogue rewrites c0 & ~3 .. c0)
__device__ __forceinline__ void zero_tail(float* buf, int ld, int M, int c0) {
    const int a0 = c0 & ~3, chunks = (ld - a0) >> 2;
    for (int i = threadIdx.x; i < M * chunks; i += blockDim.x) {
        const int r = i / chunks, c = i - r * chunks;
        *reinterpret_cast<f32x4*>(buf + r * ld + a0 + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// `ncols` columns of M batch rows -> LDS tile (remaining columns zero), optionally also into a stash
__device__ __forceinline__ void load_input(const RowSrc& src, int row0, int M, int ncols, float* dst, int ld_dst,
                                           float* stashT, int Kb) {
    // four elements per thread and round, all four loads issued before the first is used: the rows come from
    // memory that nothing has touched since they were gathered, and a tile of 16 rows x 132 columns was five
    // dependent round trips per thread
    const int n = M * ld_dst;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * blockDim.x) {
        float v[4];
        int rr[4], cc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * blockDim.x;
            rr[k] = i / ld_dst;
            cc[k] = i - rr[k] * ld_dst;
            v[k] = (i < n && cc[k] < ncols) ? row_of(src, row0 + rr[k])[cc[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * blockDim.x;
            if (i < n) {
                if (stashT != nullptr && cc[k] < ncols) stashT[stash32_off(cc[k], row0 + rr[k], Kb)] = v[k];
                dst[i] = v[k];
            }
        }
    }
}
// L2 warmer: slice `slice` of `nslices` of a packed weight copy is read and discarded (csrc/fused.hip, warm_layer)
// (`wide`: a packed copy in the order of csrc/mlp16_dev.h, 1 KB per column tile and k-group)
__device__ __forceinline__ void warm_layer(const LayerW32& L, int slice, int nslices, unsigned& acc, bool wide) {
    const int chunks = L.ntiles * L.ks * ((wide ? GROUP16_BYTES : CHUNK_BYTES) / 16);
    const int per = (chunks + nslices - 1) / nslices;
    const int lo = slice * per, hi = min(chunks, lo + per);
    const u32x4* p = reinterpret_cast<const u32x4*>(L.W);
#pragma unroll 4
    for (int i = lo + (int)threadIdx.x; i < hi; i += blockDim.x) {
        const u32x4 v = p[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
}

// ---- arguments (mirrored in ops/hip.py; the layouts of csrc/fused.hip with fp32 pointers) --------
struct CriticAhead32 {
    int mode;        // 0: off; 1: only while the next batch is in the same chunk buffer; 2: always;
                     // 3: prioritised replay, the workgroups of the pass draw their rows of the next batch first
    int role_map;    // how the two roles of a phase launch are dealt to its workgroups (role_of_block)
    float* h1;       // [B][ldh] relu(layer 1) row tiles (LDS images)
    float* h2;       // [B][ldh]
    float* lg;       // [B][nop] logits
};
struct ActorFwd32 {
    LayerW32 af[3];        // online actor, forward
    const float* s;        // [B, ldx_in] state unless the batch is packed
    float* sT;             // stashes for the actor's weight gradients (stash32_off)
    float* h1T;
    float* h2T;
    float* g_xin;          // [B][ldx]  s | pi(s) | 0   (LDS tile images, read back by the actor phase)
    float* g_h1;           // [B][ldh]
    float* g_h2;           // [B][ldh]
};
struct CriticPhaseArgs32 {
    PhaseDims d;
    const float* sa;       // [B, ldx_in]  state | action
    const float* s2;       // [B, ldx_in]  next state
    const float* reward;
    const float* done;
    const float* gamma;
    const float* weights;  // may be null
    PackedIn pk;
    LayerW32 ta[3];        // target actor, forward
    LayerW32 tc[3];        // target critic, forward
    LayerW32 cf[3];        // online critic, forward
    LayerW32 cb3, cb2;     // online critic, backward (transposed copies) of layers 3 and 2
    float* saT;
    float* h1T;
    float* h2T;
    float* dlT;
    float* dh2T;
    float* dh1T;
    float* per_loss;
    float* priority;
    ActorFwd32 fwd;
    CriticAhead32 ahead;
};
struct ActorPhaseArgs32 {
    PhaseDims d;
    const float* g_xin;
    const float* g_h1;
    const float* g_h2;
    LayerW32 cf[3];          // online critic (already updated), forward
    LayerW32 cb3, cb2, cb1;  // critic backward: layers 3, 2 and the action columns of layer 1
    LayerW32 ab3, ab2;       // actor backward: layers 3 and 2
    float* daT;
    float* dh2T;
    float* dh1T;
    float* q_val;
    TreeUpdateArgs tree;     // priority write-back in one extra workgroup (tree.n == 0: none)
    PackedIn pk;
    float* c_saT;            // the critic's stashes, written by the forward pass made ahead
    float* c_h1T;
    float* c_h2T;
    CriticAhead32 ahead;
    const SampleArgs* smp;   // ahead.mode == 3: how the NEXT batch is drawn (prioritised replay)
};

// (never on the 16x16x4 path: there a layer of any kind may be dealt out in K-slices)
__host__ __device__ __forceinline__ bool critic_red_aliased(const PhaseDims& d, bool wide) { return !wide && (NW32 - 1) * 64 <= d.ldh; }
// lean critic phase: the logits and their gradient go where the (s', a') tile was, when they fit there
__host__ __device__ __forceinline__ bool lean_outputs_in_xin2(const PhaseDims& d) { return d.nop + d.ldd <= d.ldx; }

// Which role and row tile a workgroup of a two-role phase launch takes.  role_map 0: the first n_tiles workgroups take
// the long role, the next n_tiles the short one.  role_map 1 (n_tiles a multiple of 256): the roles alternate in runs of
// 256 workgroups.  The dispatcher gives every CU one workgroup before it gives any a second one (measured: roles
// alternating workgroup by workgroup inside an XCD put two long ones on one CU, 274 us instead of 219 for Ant at 4096
// rows), so with two workgroups per CU a run of 256 long ones followed by 256 short ones pairs a long and a short one
// on every CU; with more than 256 row tiles map 0 would start with two long ones everywhere.
struct RoleTile {
    int role;      // 0: long role, 1: short role, 2: beyond both
    int tile;
};
__device__ __forceinline__ RoleTile role_of_block(int blk, int n_tiles, int role_map) {
    if (blk >= 2 * n_tiles) return RoleTile{2, blk - 2 * n_tiles};
    if (role_map == 1) {
        const int g = blk >> 8;
        return RoleTile{g & 1, ((g >> 1) << 8) | (blk & 255)};
    }
    return blk >= n_tiles ? RoleTile{1, blk - n_tiles} : RoleTile{0, blk};
}

// ---- loss rows: one wave per row ------------------------------------------------------------------
struct CriticRowsOut {
    float* dlT;
    float* per_loss;
    float* priority;
};
template <int SL, int M>
__device__ __forceinline__ void critic_rows(const PhaseDims& d, const CriticRowsOut a, int row0, int lane, int wave,
                                            const float* lg_t, const float* lg_o, float* dlg, float* sp, const float* rs) {
    for (int r = wave; r < M; r += NW32) {
        const int b = row0 + r;
        const float rew = rs[r], nd = rs[M + r], g = rs[2 * M + r], scale = rs[3 * M + r];
        if (d.distributional) {
            float lt[SL], lo[SL], m[SL], dl[SL];
            float* scratch = sp + wave * ((d.NO + 63) & ~63);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                lt[s] = j < d.NO ? lg_t[r * d.nop + j] : -INFINITY;
                lo[s] = j < d.NO ? lg_o[r * d.nop + j] : -INFINITY;
            }
            const float loss = d4pg_critic_row<SL, false>(lt, lo, rew, nd, g, d.NO, d.v_min, d.v_max, d.use_ce, scratch,
                                                          scratch, lane, m, dl);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                if (j < d.NO) {
                    const float v = dl[s] * scale;
                    dlg[r * d.ldd + j] = v;
                    a.dlT[stash32_off(j, b, d.Bp)] = v;
                }
            }
            if (lane == 0) {
                a.per_loss[b] = loss * (scale / d.inv_B);      // weighted: the finishing role sums it as is
                a.priority[b] = fabsf(loss) + 1e-4f;
            }
        } else if (lane == 0) {
            float y = rew + nd * g * lg_t[r * d.nop];
            y = fminf(d.max_value, fmaxf(d.min_value, y));
            const float diff = lg_o[r * d.nop] - y;
            const float l = diff * diff;
            const float v = 2.f * diff * scale;
            dlg[r * d.ldd] = v;
            a.dlT[stash32_off(0, b, d.Bp)] = v;
            a.per_loss[b] = l * (scale / d.inv_B);
            a.priority[b] = fabsf(l) + 1e-4f;
        }
    }
}

template <int SL, int M>
__device__ __forceinline__ void actor_rows(const PhaseDims& d, float* q_val, int row0, int lane, int wave, const float* lg_o,
                                           float* dlg) {
    for (int r = wave; r < M; r += NW32) {
        const int b = row0 + r;
        if (d.distributional) {
            float e[SL], dl[SL];
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                e[s] = j < d.NO ? lg_o[r * d.nop + j] : -INFINITY;
            }
            const float qv = d4pg_actor_row<SL, false>(e, d.NO, d.v_min, d.v_max, lane, dl);
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const int j = lane + 64 * s;
                if (j < d.NO) dlg[r * d.ldd + j] = dl[s] * d.inv_B;
            }
            if (lane == 0) q_val[b] = qv;
        } else if (lane == 0) {
            dlg[r * d.ldd] = -d.inv_B;
            q_val[b] = lg_o[r * d.nop];
        }
    }
}

// ---- the online actor's forward pass, in extra workgroups of the critic phase launch --------------
template <int MT, int KSH, int KSN>
__device__ __forceinline__ void actor_forward_role(const CriticPhaseArgs32& a, int row0, char* smem) {
    constexpr int M = 4 * MT;
    const PhaseDims& d = a.d;
    const ActorFwd32& f = a.fwd;
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    float* xin = (float*)carve(M * d.ldx * 4);     // [s | pi(s)]
    float* bufA = (float*)carve(M * d.ldh * 4);    // h1
    float* bufB = (float*)carve(M * d.ldh * 4);    // h2
    float* red = (float*)carve(red_floats32(MT, KSH == 0) * 4);
    Chain32<MT, KSH, KSN> ch(red);
    ch.template prefetch_n<0>(f.af[0]);
    // (the epilogue lambdas capture scalars only: a closure that holds a copy of the kernel arguments
    // sends all of them through scratch memory, which also makes every buffer descriptor look divergent)
    const int ldh = d.ldh, ldx = d.ldx, Bp = d.Bp, S = d.S;
    float* const h1T = f.h1T;
    float* const h2T = f.h2T;
    zero_tail(bufA, ldh, M, d.H);
    zero_tail(bufB, ldh, M, d.H);
    load_input(a.pk.enabled ? packed_src(a.pk, 0) : plain_src(f.s, d.ldx_in, d.B), row0, M, S, xin, ldx, f.sT, Bp);
    __syncthreads();
    ch.template narrow<0>(xin, ldx, f.af[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufA, ldh, r0, col, v);
        put_stash4(h1T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<1>(f.af[1]); });
    __syncthreads();
    copy16(f.g_h1 + (size_t)row0 * ldh, bufA, M * ldh * 4);
    ch.template hidden<1>(bufA, ldh, f.af[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufB, ldh, r0, col, v);
        put_stash4(h2T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_s<0>(f.af[2]); });
    __syncthreads();
    copy16(f.g_h2 + (size_t)row0 * ldh, bufB, M * ldh * 4);
    ch.template split<0>(bufB, ldh, f.af[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fast_tanh(v[r]);
        put_lds4(xin, ldx, r0, S + col, v);
    }, [] {});
    __syncthreads();
    copy16(f.g_xin + (size_t)row0 * d.ldx, xin, M * d.ldx * 4);
}

// Everything after the logits of both critics: loss rows, then dX through critic layers 3 and 2.
template <int MT, int KSH, int KSN>
__device__ __forceinline__ void critic_tail(const CriticPhaseArgs32& a, Chain32<MT, KSH, KSN>& ch, int row0, int lane,
                                            int wave, const float* lg_t, const float* lg_o, float* dlg, float* sp,
                                            const float* rs, float* bufC, float* bufD) {
    constexpr int M = 4 * MT;
    const PhaseDims& d = a.d;
    __syncthreads();
    STAMP(0, 9);
    critic_rows<1, M>(d, CriticRowsOut{a.dlT, a.per_loss, a.priority}, row0, lane, wave, lg_t, lg_o, dlg, sp, rs);
    __syncthreads();
    STAMP(0, 10);
    const int ldh = d.ldh, Bp = d.Bp;
    float* const dh2T = a.dh2T;
    float* const dh1T = a.dh1T;
    ch.template narrow<1>(dlg, d.ldd, a.cb3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufD[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufD, ldh, r0, col, v);       // in place: each element is read and written by one lane
        put_stash4(dh2T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<0>(a.cb2); });
    __syncthreads();
    STAMP(0, 11);
    ch.template hidden<0>(bufD, ldh, a.cb2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufC[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_stash4(dh1T, Bp, col, row0 + r0, v);
    }, [] {});
#ifdef D4PG_STAMPS
    __syncthreads();
#endif
    STAMP(0, 12);
}

// (The lean build asks for four waves per SIMD, i.e. at most 128 registers; 2 is what a 512-thread workgroup implies.)
template <int MT, int KSH, int KSN>
__global__ __launch_bounds__(64 * NW32, lean_lds(KSH, KSN) ? 4 : 2) void critic_phase32_kernel(const CriticPhaseArgs32) {
    const CriticPhaseArgs32& a = kernel_args<CriticPhaseArgs32>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 4 * MT;
    const PhaseDims& d = a.d;
    // Row tile and role of this workgroup.  (Only the lean build deals its roles in more than one way, and the code of
    // the other builds is kept as it was: these kernels are sensitive to what the compiler makes of their first lines.)
    int row0 = blockIdx.x * M;
    bool warmer = row0 >= 2 * d.B, forward = row0 >= d.B;
    unsigned warm_tile = blockIdx.x - 2 * (d.B / M);
    if constexpr (lean_lds(KSH, KSN)) {
        const RoleTile rt = role_of_block(blockIdx.x, d.B / M, a.ahead.role_map);
        warmer = rt.role == 2;
        forward = rt.role == 1;
        warm_tile = rt.tile;
        row0 = rt.tile * M + (forward ? d.B : 0);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (warmer) {              // beyond both halves: L2 warmers
        const int sl = warm_tile >> 3, nsl = (gridDim.x - 2 * (d.B / M) + 7) >> 3;
        unsigned acc = 0;
        warm_layer(a.ta[0], sl, nsl, acc, KSH == 0);
        warm_layer(a.fwd.af[0], sl, nsl, acc, KSH == 0);
        warm_layer(a.ta[1], sl, nsl, acc, KSH == 0);
        warm_layer(a.fwd.af[1], sl, nsl, acc, KSH == 0);
        warm_layer(a.ta[2], sl, nsl, acc, KSH == 0);
        warm_layer(a.fwd.af[2], sl, nsl, acc, KSH == 0);
        warm_layer(a.tc[0], sl, nsl, acc, KSH == 0);
        warm_layer(a.tc[1], sl, nsl, acc, KSH == 0);
        warm_layer(a.tc[2], sl, nsl, acc, KSH == 0);
        if (a.ahead.mode == 0) {      // (else the online critic's forward pass was most likely made ahead)
            warm_layer(a.cf[0], sl, nsl, acc, KSH == 0);
            warm_layer(a.cf[1], sl, nsl, acc, KSH == 0);
            warm_layer(a.cf[2], sl, nsl, acc, KSH == 0);
        }
        warm_layer(a.cb3, sl, nsl, acc, KSH == 0);
        warm_layer(a.cb2, sl, nsl, acc, KSH == 0);
        asm volatile("" ::"v"(acc));
        STAMP_SPAN(0, 25, true);
        return;
    }
    if (forward) {             // the second half of the grid: actor forward (+ the per-update tick)
        STAMP_SPAN(0, 22, false);
        if (a.pk.do_tick && row0 == d.B && threadIdx.x == 0)
            devstate_tick(a.pk.st, a.pk.beta1, a.pk.beta2, a.pk.per_b0, a.pk.per_b1, a.pk.total_steps);
        actor_forward_role<MT, KSH, KSN>(a, row0 - d.B, smem);
        STAMP_SPAN(0, 23, true);
        return;
    }
    STAMP_SPAN(0, 20, false);
    STAMP(0, 31);
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    constexpr bool LEAN = lean_lds(KSH, KSN);
    float* xin = (float*)carve(M * d.ldx * 4);     // [s | a]
    float* xin2 = (float*)carve(M * d.ldx * 4);    // [s' | a']
    float* bufA = (float*)carve(M * d.ldh * 4);
    float* bufB = (float*)carve(M * d.ldh * 4);
    // h1 / h2 of the online critic (masks for the backward; h2 becomes dh2 in place).  Lean: the target networks'
    // tiles are dead once the target logits are out, the online critic's tiles take their place.
    float* bufC = LEAN ? bufA : (float*)carve(M * d.ldh * 4);
    float* bufD = LEAN ? bufB : (float*)carve(M * d.ldh * 4);
    float *dlg, *lg_t, *lg_o;
    if (LEAN && lean_outputs_in_xin2(d)) {         // (s', a') is dead once the target critic's first layer is through
        lg_o = xin2;
        dlg = xin2 + M * d.nop;
        lg_t = (float*)carve(M * d.nop * 4);
    } else {
        dlg = (float*)carve(M * d.ldd * 4);
        lg_t = (float*)carve(M * d.nop * 4);
        lg_o = (float*)carve(M * d.nop * 4);
    }
    // per-wave projection rows of the loss (lean: where the partial sums are; no layer runs beside the loss rows)
    float* sp = LEAN ? nullptr : (float*)carve(NW32 * ((d.NO + 63) & ~63) * 4);
    float* rs = (float*)carve(4 * M * 4);          // reward | 1 - done | gamma | weight / B
    // The K-slice partial sums of the three output layers (a', target logits, logits) go where bufA is:
    // its tile (an input of the hidden layer before) is dead whenever one of them runs.  Only when they
    // fit; a 16-row tile of a 400-wide network needs every byte of the LDS.
    float* red = critic_red_aliased(d, KSH == 0) ? bufA : (float*)carve(red_floats32(MT, KSH == 0) * 4);
    if (LEAN) sp = red;
    Chain32<MT, KSH, KSN> ch(red);
    ch.template prefetch_n<0>(a.ta[0]);
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
    if (!LEAN) zero_block(dlg, M * d.ldd * 4);      // (lean: once its place is free, below)
    zero_tail(bufA, d.ldh, M, d.H);
    zero_tail(bufB, d.ldh, M, d.H);
    // Was the online critic's forward pass on exactly these rows made ahead, after the last change of
    // the critic weights?  (wave-uniform; CriticAhead in csrc/fused.hip)
    bool ahead_ok = false;
    if (a.ahead.mode != 0 && (a.pk.enabled || a.ahead.mode == 3))
        ahead_ok = a.pk.st->pad_[AHEAD_STEP] == a.pk.st->step_done + 1 &&
                   a.pk.st->pad_[AHEAD_ROWS] == (long long)(uintptr_t)src_sa.base;
    STAMP(0, 0);
    if (ahead_ok) {
        if (!LEAN) {      // (lean: their buffers hold the target networks' tiles until the target logits are out)
            copy16(bufC, a.ahead.h1 + (size_t)row0 * d.ldh, M * d.ldh * 4);
            copy16(bufD, a.ahead.h2 + (size_t)row0 * d.ldh, M * d.ldh * 4);
            copy16(lg_o, a.ahead.lg + (size_t)row0 * d.nop, M * d.nop * 4);
        }
    } else {
        if (!LEAN) {
            zero_tail(bufC, d.ldh, M, d.H);
            zero_tail(bufD, d.ldh, M, d.H);
        }
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

    const int ldh = d.ldh, ldx = d.ldx, nop = d.nop, Bp = d.Bp, S = d.S;
    float* const h1T = a.h1T;
    float* const h2T = a.h2T;
    auto relu_to = [&](float* buf) {
        return [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(buf, ldh, r0, col, v);
        };
    };
    // ---- target actor: s' -> a' (written into the action columns of xin2) ----
    ch.template narrow<0>(xin2, d.ldx, a.ta[0], relu_to(bufA), [&] { ch.template prefetch_h<1>(a.ta[1]); }, 13);
    __syncthreads();
    STAMP(0, 2);
    ch.template hidden<1>(bufA, d.ldh, a.ta[1], relu_to(bufB), [&] { ch.template prefetch_s<0>(a.ta[2]); }, 26);
    __syncthreads();
    STAMP(0, 3);
    ch.template split<0>(bufB, d.ldh, a.ta[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fast_tanh(v[r]);
        put_lds4(xin2, ldx, r0, S + col, v);
    }, [&] { ch.template prefetch_n<1>(a.tc[0]); });
    __syncthreads();
    STAMP(0, 4);
    // ---- target critic: (s', a') -> logits_t ----
    ch.template narrow<1>(xin2, d.ldx, a.tc[0], relu_to(bufA), [&] { ch.template prefetch_h<0>(a.tc[1]); });
    __syncthreads();
    STAMP(0, 5);
    ch.template hidden<0>(bufA, d.ldh, a.tc[1], relu_to(bufB), [&] { ch.template prefetch_s<1>(a.tc[2]); });
    __syncthreads();
    STAMP(0, 6);
    // With the online critic's tiles made ahead its three layers are skipped; the first of them is
    // prefetched by the layer before, so skipping is expressed as a prefetch without k-groups to load.
    ch.template split<1>(bufB, d.ldh, a.tc[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lg_t[(r0 + r) * nop + col] = v[r];
    }, [&] { ch.template prefetch_n_if<0>(a.cf[0], !ahead_ok); });
    if (!ahead_ok) {
        // ---- online critic: (s, a) -> logits, activations stashed for wgrad ----
        // (no barrier since the layer before, whose K-slices met through `red`: on the 16x16x4 path the first layer
        // here may put partial sums there as well, and must not while a wave still reads the last ones)
        if constexpr (KSH == 0) __syncthreads();
        ch.template narrow<0>(xin, d.ldx, a.cf[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(bufC, ldh, r0, col, v);
            put_stash4(h1T, Bp, col, row0 + r0, v);
        }, [&] { ch.template prefetch_h<1>(a.cf[1]); });
        __syncthreads();
        ch.template hidden<1>(bufC, d.ldh, a.cf[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(bufD, ldh, r0, col, v);
            put_stash4(h2T, Bp, col, row0 + r0, v);
        }, [&] { ch.template prefetch_s<0>(a.cf[2]); });
        __syncthreads();
        ch.template split<0>(bufD, d.ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lg_o[(r0 + r) * nop + col] = v[r];
        }, [] {});
    } else if (LEAN) {
        __syncthreads();      // the target critic's last layer has read bufB
        copy16x4(bufC, a.ahead.h1 + (size_t)row0 * d.ldh, M * d.ldh * 4);
        copy16x4(bufD, a.ahead.h2 + (size_t)row0 * d.ldh, M * d.ldh * 4);
        copy16x4(lg_o, a.ahead.lg + (size_t)row0 * d.nop, M * d.nop * 4);
    }
    if (LEAN) zero_block(dlg, M * d.ldd * 4);      // (the loss rows start behind a barrier)
    // No prefetched registers are live where the two paths join: the first backward layer is fetched
    // here and lands while the loss rows are computed.
    ch.template prefetch_n<1>(a.cb3);
    critic_tail<MT, KSH, KSN>(a, ch, row0, lane, wave, lg_t, lg_o, dlg, sp, rs, bufC, bufD);
    STAMP_SPAN(0, 21, true);
}

// ---- the online critic's forward pass for the NEXT update, in extra workgroups of the actor phase ----
// Prioritised replay (ahead.mode == 3).  The batch of the next update can only be drawn once this update's
// priorities are in the tree; the tree update rides in the launch BEFORE this one (the critic's weight-gradient
// launch, wgrad32_kernel), so the tree is complete when this launch starts.  The workgroups here draw the rows
// of their own tile (one wave per row, the sampling kernel's code and stream positions, hence the same rows)
// into the learner's input buffers and go on with the forward pass: sampling and pass (~17 us) hide under the
// actor role of the launch instead of following the update's last kernel.
template <int MT, int KSH, int KSN>
__device__ __forceinline__ void critic_ahead_role(const ActorPhaseArgs32& a, int row0, char* smem) {
    constexpr int M = 4 * MT;
    const PhaseDims& d = a.d;
    const bool per = a.ahead.mode == 3;
    if (!per) {
        const long long s_next = a.pk.st->step_done - a.pk.step0 + 1;
        // mode 1: the next chunk buffer may still be receiving its rows (all-gather on another stream)
        if (a.ahead.mode == 1 && s_next % a.pk.K == 0) return;
    }
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    float* xin = (float*)carve(M * d.ldx * 4);
    float* bufC = (float*)carve(M * d.ldh * 4);
    float* bufD = (float*)carve(M * d.ldh * 4);
    float* lg = (float*)carve(M * d.nop * 4);
    float* red = (float*)carve(red_floats32(MT, KSH == 0) * 4);
    Chain32<MT, KSH, KSN> ch(red);
    ch.template prefetch_n<0>(a.cf[0]);
    zero_tail(bufC, d.ldh, M, d.H);
    zero_tail(bufD, d.ldh, M, d.H);
    RowSrc src;
    if (per) {
        const SampleArgs& sa = *a.smp;
        const long long step = a.pk.st->step_cur;      // updates completed once this one is = stream position of the next
        const long long size = sa.st->replay_size;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (size > 0)
            for (int r = wave; r < M; r += NW32) sample_row(sa, row0 + r, step, size, threadIdx.x & 63);
        __syncthreads();                                            // the rows, to the other waves of this workgroup
        src = plain_src(sa.sa, sa.ld_x, d.B);
    } else {
        src = packed_src(a.pk, 0, 1);
    }
    const int ldh = d.ldh, nop = d.nop, Bp = d.Bp;
    float* const h1T = a.c_h1T;
    float* const h2T = a.c_h2T;
    load_input(src, row0, M, d.S + d.A, xin, d.ldx, a.c_saT, Bp);
    __syncthreads();
    ch.template narrow<0>(xin, d.ldx, a.cf[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufC, ldh, r0, col, v);
        put_stash4(h1T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<1>(a.cf[1]); });
    __syncthreads();
    copy16(a.ahead.h1 + (size_t)row0 * ldh, bufC, M * ldh * 4);
    ch.template hidden<1>(bufC, ldh, a.cf[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufD, ldh, r0, col, v);
        put_stash4(h2T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_s<0>(a.cf[2]); });
    __syncthreads();
    copy16(a.ahead.h2 + (size_t)row0 * ldh, bufD, M * ldh * 4);
    ch.template split<0>(bufD, ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[(r0 + r) * nop + col] = v[r];
    }, [] {});
    __syncthreads();
    copy16(a.ahead.lg + (size_t)row0 * d.nop, lg, M * d.nop * 4);
    if (row0 == 0 && threadIdx.x == 0) {     // the tag: which update and which rows these tiles are for
        a.pk.st->pad_[AHEAD_STEP] = a.pk.st->step_done + 2;
        a.pk.st->pad_[AHEAD_ROWS] = (long long)(uintptr_t)src.base;
    }
}

template <int MT, int KSH, int KSN>
__global__ __launch_bounds__(64 * NW32, lean_lds(KSH, KSN) ? 4 : 2) void actor_phase32_kernel(const ActorPhaseArgs32) {
    const ActorPhaseArgs32& a = kernel_args<ActorPhaseArgs32>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 4 * MT;
    const PhaseDims& d = a.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_tree = a.tree.n > 0 ? (a.tree.parts > 1 ? a.tree.parts : 1) : 0;
    const int n_work = d.B / M + n_tree + (a.ahead.mode != 0 ? d.B / M : 0);
    // the actor role and the critic's pass for the next update, dealt alternately (the host allows this without a tree role)
    int row0 = blockIdx.x * M;
    bool dealt = false;
    if constexpr (lean_lds(KSH, KSN)) {      // (as in the critic phase: the other builds keep their code)
        dealt = a.ahead.role_map == 1 && a.ahead.mode != 0 && (int)blockIdx.x < 2 * (d.B / M);
        const RoleTile rt = role_of_block(blockIdx.x, d.B / M, dealt ? 1 : 0);
        if (dealt && rt.role == 1) {
            critic_ahead_role<MT, KSH, KSN>(a, rt.tile * M, smem);
            return;
        }
        if (dealt) row0 = rt.tile * M;
    }
    if ((int)blockIdx.x >= n_work) {                // L2 warmers
        const int sl = (blockIdx.x - n_work) >> 3, nsl = (gridDim.x - n_work + 7) >> 3;
        unsigned acc = 0;
        warm_layer(a.cf[0], sl, nsl, acc, KSH == 0);
        warm_layer(a.cf[1], sl, nsl, acc, KSH == 0);
        warm_layer(a.cf[2], sl, nsl, acc, KSH == 0);
        warm_layer(a.cb3, sl, nsl, acc, KSH == 0);
        warm_layer(a.cb2, sl, nsl, acc, KSH == 0);
        warm_layer(a.cb1, sl, nsl, acc, KSH == 0);
        warm_layer(a.ab3, sl, nsl, acc, KSH == 0);
        warm_layer(a.ab2, sl, nsl, acc, KSH == 0);
        asm volatile("" ::"v"(acc));
        return;
    }
    if (!dealt && (int)blockIdx.x >= d.B / M + n_tree) {      // the online critic's forward pass of the next update
        critic_ahead_role<MT, KSH, KSN>(a, (blockIdx.x - (d.B / M + n_tree)) * M, smem);
        return;
    }
    if (row0 >= d.B) {         // the extra workgroups: priorities back into the tree
        tree_update_body<NW32>(a.tree, (int)blockIdx.x - d.B / M, smem, (lean_lds(KSH, KSN) ? 2 : 4) * M * d.ldh * 4);    // (the hidden tiles alone)
        return;
    }
    STAMP(1, 31);
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    constexpr bool LEAN = lean_lds(KSH, KSN);
    float* xin = (float*)carve(M * d.ldx * 4);     // [s | pi(s)]
    // h1 actor; h2 actor, then dh2 actor in place.  Lean: only needed as masks of the actor's backward layers, when
    // the critic's tiles are dead: fetched into their buffers then.
    float* bufA = LEAN ? nullptr : (float*)carve(M * d.ldh * 4);
    float* bufB = LEAN ? nullptr : (float*)carve(M * d.ldh * 4);
    float* bufC = (float*)carve(M * d.ldh * 4);    // h1 critic, then dh1 critic in place
    float* bufD = (float*)carve(M * d.ldh * 4);    // h2 critic, then dh2 critic in place
    if (LEAN) {
        bufA = bufC;
        bufB = bufD;
    }
    float* dlg = (float*)carve(M * d.ldd * 4);
    float* dab = (float*)carve(M * d.lda_ * 4);
    float* lg_o = (float*)carve(M * d.nop * 4);
    float* red = (float*)carve(red_floats32(MT, KSH == 0) * 4);
    Chain32<MT, KSH, KSN> ch(red);
    ch.template prefetch_n<0>(a.cf[0]);
    // ---- the actor forward pass of this tile was made alongside the critic phase ----
    copy16(xin, a.g_xin + (size_t)row0 * d.ldx, M * d.ldx * 4);
    if (!LEAN) {
        copy16(bufA, a.g_h1 + (size_t)row0 * d.ldh, M * d.ldh * 4);
        copy16(bufB, a.g_h2 + (size_t)row0 * d.ldh, M * d.ldh * 4);
    }
    zero_block(dlg, M * (d.ldd + d.lda_) * 4);      // dlg and dab are adjacent
    zero_tail(bufC, d.ldh, M, d.H);
    zero_tail(bufD, d.ldh, M, d.H);
    __syncthreads();
    STAMP(1, 0);
    const int ldh = d.ldh, ldx = d.ldx, nop = d.nop, Bp = d.Bp, S = d.S, lda_ = d.lda_;
    float* const daT = a.daT;
    float* const dh2T = a.dh2T;
    float* const dh1T = a.dh1T;
    auto relu_to = [&](float* buf) {
        return [=](int r0, int col, f32x4 v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            put_lds4(buf, ldh, r0, col, v);
        };
    };
    // ---- critic forward at (s, pi(s)) ----
    ch.template narrow<0>(xin, d.ldx, a.cf[0], relu_to(bufC), [&] { ch.template prefetch_h<1>(a.cf[1]); });
    __syncthreads();
    STAMP(1, 1);
    ch.template hidden<1>(bufC, ldh, a.cf[1], relu_to(bufD), [&] { ch.template prefetch_s<0>(a.cf[2]); });
    __syncthreads();
    STAMP(1, 2);
    ch.template split<0>(bufD, ldh, a.cf[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lg_o[(r0 + r) * nop + col] = v[r];
    }, [&] { ch.template prefetch_n<1>(a.cb3); });
    __syncthreads();
    STAMP(1, 3);
    // ---- actor loss: -mean Q ----
    actor_rows<1, M>(d, a.q_val, row0, lane, wave, lg_o, dlg);
    __syncthreads();
    STAMP(1, 4);
    // ---- critic backward (dX only) down to the action ----
    ch.template narrow<1>(dlg, d.ldd, a.cb3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufD[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufD, ldh, r0, col, v);
    }, [&] { ch.template prefetch_h<0>(a.cb2); });
    __syncthreads();
    STAMP(1, 5);
    ch.template hidden<0>(bufD, ldh, a.cb2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufC[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufC, ldh, r0, col, v);
    }, [&] { ch.template prefetch_s<1>(a.cb1); });
    __syncthreads();
    STAMP(1, 6);
    if (LEAN) copy16x4(bufB, a.g_h2 + (size_t)row0 * d.ldh, M * d.ldh * 4);      // dh2 critic is dead
    ch.template split<1>(bufC, ldh, a.cb1, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y = xin[(r0 + r) * ldx + S + col];      // pi(s): d tanh = 1 - y^2
            v[r] *= (1.f - y * y);
        }
        put_lds4(dab, lda_, r0, col, v);
        put_stash4(daT, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_n<0>(a.ab3); });
    __syncthreads();
    STAMP(1, 7);
    if (LEAN) copy16x4(bufA, a.g_h1 + (size_t)row0 * d.ldh, M * d.ldh * 4);      // dh1 critic is dead
    // ---- actor backward ----
    ch.template narrow<0>(dab, lda_, a.ab3, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufB[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_lds4(bufB, ldh, r0, col, v);
        put_stash4(dh2T, Bp, col, row0 + r0, v);
    }, [&] { ch.template prefetch_h<1>(a.ab2); });
    __syncthreads();
    STAMP(1, 8);
    ch.template hidden<1>(bufB, ldh, a.ab2, [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bufA[(r0 + r) * ldh + col] > 0.f ? v[r] : 0.f;
        put_stash4(dh1T, Bp, col, row0 + r0, v);
    }, [] {});
#ifdef D4PG_STAMPS
    __syncthreads();
#endif
    STAMP(1, 9);
}

// ---- batched policy inference: a = tanh(actor(obs)) for E rows, ONE launch ------------------------------
// What an agent needs per environment step (models/d4pg/networks.py:78-81 is six kernels, an H2D and a D2H
// copy per step and environment): the three actor layers on a tile of 4 * MT environments per workgroup,
// weights from packed fp32 copies (csrc/mlp32_dev.h), activations in LDS.  Used by GPU exploiters /
// explorers (actors/gpu_actor.py) and by the device-resident rollout (actors/gpu_rollout.py).
// Exploration noise in the policy's last epilogue (utils/utils.py:9-34 of the reference: Ornstein-Uhlenbeck state per
// action dimension, x <- x + theta (mu - x) + sigma N(0, 1), action = clip(policy + x)).  The draw is the one of
// ou_noise_kernel (csrc/nstep.hip): Philox-4x32 at (element index, counter), Box-Muller; so a rollout step is the
// policy launch and the environment launch, nothing in between.
struct OuFuse32 {
    float* x;                             // [E][ld_act] noise state, updated in place; null: no noise, `act` gets the policy's output
    const unsigned long long* counter;    // Philox stream position of this step (the environment's step counter)
    unsigned long long seed;
    float theta, mu, sigma, low, high;
    int pad_;
};
struct ActorInferArgs32 {
    LayerW32 l[3];
    const float* obs;     // [E][ld_obs], first S columns are the observation
    float* act;           // [E][ld_act], first A columns are written
    int E, S, A, H;
    int ld_obs, ld_act, ldx, ldh;
    float action_scale;
    int pad_;
    OuFuse32 ou;
};

template <int MT, int KSH, int KSN>
__global__ __launch_bounds__(64 * NW32) void actor_infer32_kernel(const ActorInferArgs32) {
    const ActorInferArgs32& a = kernel_args<ActorInferArgs32>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 4 * MT;
    const int row0 = blockIdx.x * M;
    int off = 0;
    auto carve = [&](int bytes) { char* p = smem + off; off += (bytes + 15) & ~15; return p; };
    float* xin = (float*)carve(M * a.ldx * 4);
    float* bufA = (float*)carve(M * a.ldh * 4);
    float* bufB = (float*)carve(M * a.ldh * 4);
    float* red = (float*)carve(red_floats32(MT, KSH == 0) * 4);
    Chain32<MT, KSH, KSN> ch(red);
    ch.template prefetch_n<0>(a.l[0]);
    const int ldh = a.ldh, ldx = a.ldx, E = a.E, ld_act = a.ld_act;
    float* const act = a.act;
    const float scale = a.action_scale;
    zero_tail(bufA, ldh, M, a.H);
    zero_tail(bufB, ldh, M, a.H);
    for (int i = threadIdx.x; i < M * ldx; i += blockDim.x) {
        const int r = i / ldx, c = i - r * ldx;
        xin[i] = (c < a.S && row0 + r < E) ? a.obs[(size_t)(row0 + r) * a.ld_obs + c] : 0.f;
    }
    __syncthreads();
    ch.template narrow<0>(xin, ldx, a.l[0], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufA, ldh, r0, col, v);
    }, [&] { ch.template prefetch_h<1>(a.l[1]); });
    __syncthreads();
    ch.template hidden<1>(bufA, ldh, a.l[1], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        put_lds4(bufB, ldh, r0, col, v);
    }, [&] { ch.template prefetch_s<0>(a.l[2]); });
    __syncthreads();
    const OuFuse32 ou = a.ou;
    const int A = a.A;
    const unsigned long long ctr = ou.x != nullptr ? *ou.counter : 0ull;
    ch.template split<0>(bufB, ldh, a.l[2], [=](int r0, int col, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + r0 + r;
            if (row >= E) continue;
            float y = fast_tanh(v[r]) * scale;
            if (ou.x != nullptr && col < A) {
                const size_t i = (size_t)row * ld_act + col;
                const float x = ou_advance(ou.x[i], (uint32_t)i, ctr, ou.seed, ou.theta, ou.mu, ou.sigma);
                ou.x[i] = x;
                y = ou_apply(y, x, ou.low, ou.high);
            }
            act[(size_t)row * ld_act + col] = y;
        }
    }, [] {});
}

// ---- host launchers -----------------------------------------------------------------------------
static int lds_bytes_critic32(const PhaseDims& d, int M, int mt, bool wide = false, bool lean = false) {
    auto al = [](int b) { return (b + 15) & ~15; };
    if (lean)      // (the carve-up of critic_phase32_kernel with lean_lds)
        return 2 * al(M * d.ldx * 4) + 2 * al(M * d.ldh * 4) + (lean_outputs_in_xin2(d) ? 0 : al(M * d.ldd * 4) + al(M * d.nop * 4)) +
               al(M * d.nop * 4) + al(4 * M * 4) + al(red_floats32(mt, true) * 4);
    return 2 * al(M * d.ldx * 4) + 4 * al(M * d.ldh * 4) + al(M * d.ldd * 4) + 2 * al(M * d.nop * 4) +
           al(NW32 * ((d.NO + 63) & ~63) * 4) + al(4 * M * 4) + (critic_red_aliased(d, wide) ? 0 : al(red_floats32(mt, wide) * 4));
}
static int lds_bytes_actor32(const PhaseDims& d, int M, int mt, bool wide = false, bool lean = false) {
    auto al = [](int b) { return (b + 15) & ~15; };
    return al(M * d.ldx * 4) + (lean ? 2 : 4) * al(M * d.ldh * 4) + al(M * d.ldd * 4) + al(M * d.lda_ * 4) + al(M * d.nop * 4) +
           al(red_floats32(mt, wide) * 4);
}

#define D4PG_PHASE32_LAUNCHER(NAME, KERNEL, ARGS)                                                         \
    template <int KSH>                                                                                     \
    int NAME(const ARGS* a, int mt, int ksn, int grid, int lds, hipStream_t stream) {                      \
        if (mt == 1) {                                                                                     \
            if (ksn <= 2) return launch_phase(KERNEL<1, KSH, 2>, *a, grid, NW32, lds, stream);             \
            return launch_phase(KERNEL<1, KSH, 4>, *a, grid, NW32, lds, stream);                           \
        }                                                                                                  \
        if (mt == 2) {                                                                                     \
            if (ksn <= 2) return launch_phase(KERNEL<2, KSH, 2>, *a, grid, NW32, lds, stream);             \
            return launch_phase(KERNEL<2, KSH, 4>, *a, grid, NW32, lds, stream);                           \
        }                                                                                                  \
        if (ksn <= 2) return launch_phase(KERNEL<4, KSH, 2>, *a, grid, NW32, lds, stream);                 \
        return launch_phase(KERNEL<4, KSH, 4>, *a, grid, NW32, lds, stream);                               \
    }
D4PG_PHASE32_LAUNCHER(launch_critic32_ksh, critic_phase32_kernel, CriticPhaseArgs32)
D4PG_PHASE32_LAUNCHER(launch_infer32_ksh, actor_infer32_kernel, ActorInferArgs32)
D4PG_PHASE32_LAUNCHER(launch_actor32_ksh, actor_phase32_kernel, ActorPhaseArgs32)

// the 16x16x4 path (KSH = 0): tiles of 8 or 16 rows, every width at run time
#define D4PG_PHASE32W_LAUNCHER(NAME, KERNEL, ARGS)                                                        \
    int NAME(const ARGS* a, int mt, int grid, int lds, hipStream_t stream) {                               \
        if (mt == 2) return launch_phase(KERNEL<2, 0, 0>, *a, grid, NW32, lds, stream);                    \
        return launch_phase(KERNEL<4, 0, 0>, *a, grid, NW32, lds, stream);                                 \
    }
// ... its lean build (two workgroups per CU; tiles of 16 rows)
#define D4PG_PHASE32WL_LAUNCHER(NAME, KERNEL, ARGS)                                                       \
    int NAME(const ARGS* a, int grid, int lds, hipStream_t stream) {                                       \
        return launch_phase(KERNEL<4, 0, 1>, *a, grid, NW32, lds, stream);                                 \
    }
#if D4PG_KSH_PART == 0 || D4PG_KSH_PART == 99
D4PG_PHASE32W_LAUNCHER(launch_critic32_wide, critic_phase32_kernel, CriticPhaseArgs32)
D4PG_PHASE32W_LAUNCHER(launch_actor32_wide, actor_phase32_kernel, ActorPhaseArgs32)
D4PG_PHASE32W_LAUNCHER(launch_infer32_wide, actor_infer32_kernel, ActorInferArgs32)
D4PG_PHASE32WL_LAUNCHER(launch_critic32_lean, critic_phase32_kernel, CriticPhaseArgs32)
D4PG_PHASE32WL_LAUNCHER(launch_actor32_lean, actor_phase32_kernel, ActorPhaseArgs32)
#else
int launch_critic32_wide(const CriticPhaseArgs32*, int, int, int, hipStream_t);
int launch_actor32_wide(const ActorPhaseArgs32*, int, int, int, hipStream_t);
int launch_infer32_wide(const ActorInferArgs32*, int, int, int, hipStream_t);
int launch_critic32_lean(const CriticPhaseArgs32*, int, int, hipStream_t);
int launch_actor32_lean(const ActorPhaseArgs32*, int, int, hipStream_t);
#endif

#define D4PG_PHASE32_INSTANCES(PREFIX, KSHV)                                                                   \
    PREFIX template int launch_critic32_ksh<KSHV>(const CriticPhaseArgs32*, int, int, int, int, hipStream_t);  \
    PREFIX template int launch_actor32_ksh<KSHV>(const ActorPhaseArgs32*, int, int, int, int, hipStream_t);   \
    PREFIX template int launch_infer32_ksh<KSHV>(const ActorInferArgs32*, int, int, int, int, hipStream_t);
// Hidden widths with compile-time 4x4x1 kernels, as chunks of 32 (KSH = ceil(width / 32)): 400, 128 / 100, 256, and the
// other sizes people use (64, 200, 300, 512); one object each (build.py::FUSED_PARTS).  Any other width runs the
// run-time-width 16x16x4 kernels.
#define D4PG_KSH_LIST(X) X(13) X(4) X(8) X(2) X(7) X(10) X(16)
#if D4PG_KSH_PART == -1          // main unit: the widths live in their own objects
#define D4PG_EXTERN_INSTANCE(K) D4PG_PHASE32_INSTANCES(extern, K)
D4PG_KSH_LIST(D4PG_EXTERN_INSTANCE)
#elif D4PG_KSH_PART != 99 && D4PG_KSH_PART != 0        // the unit of one width
D4PG_PHASE32_INSTANCES(, D4PG_KSH_PART)
#endif

#if D4PG_MAIN_UNIT
static int check_dims32(const PhaseDims& d, int M) {
    if (d.B < M || d.B % M) return -1;
    if (d.Bp < d.B || (d.Bp & 31)) return -2;
    if ((d.ldx & 3) || (d.ldh & 3) || (d.ldd & 3) || (d.lda_ & 3)) return -3;
    if (d.NO < 1 || d.NO > 64 || d.A > 64 || d.nop < d.NO || (d.nop & 3)) return -4;
    if (d.ldx < d.S + d.A || d.ldx_in < d.S + d.A) return -5;
    return 0;
}
// kind 0: N = hidden width, K = hidden width; 1: single column tile, K = hidden width in K-slices;
// 2: K <= 32 * ksn, N = hidden width.  `ld` = floats readable per row of the LDS tile it multiplies.
static int check_layer32(const LayerW32& L, int kind, int ksh, int ksn, int ld) {
    if (L.W == nullptr || L.bias == nullptr || (((uintptr_t)L.W) & 15)) return -6;
    if (L.ntiles < 1 || L.ntiles > NW32 || L.ksplit < 1 || L.ksplit > NW32 || L.nch < 1) return -6;
    if (L.ks != L.nch * L.ksplit || L.N < 1 || L.N > 64 * L.ntiles || L.ks * 32 + LDS_PAD32 > ld) return -7;
    if ((long long)L.ntiles * L.ks * CHUNK_BYTES >= (1ll << 31)) return -7;
    if (kind == 0) return (L.ksplit == 1 && L.nch == ksh) ? 0 : -10;
    if (kind == 1) return (L.ntiles == 1 && L.nch == split_nch(ksh) && L.ks >= ksh) ? 0 : -10;
    return (L.ksplit == 1 && L.nch <= ksn && L.kg >= 1 && L.kg <= KCH * L.nch) ? 0 : -10;
}
static int hidden_chunks(const PhaseDims& d) { return (d.H + 31) / 32; }
static int max2i(int a, int b) { return a > b ? a : b; }

// L2 warmer workgroups per phase launch, only while the grid leaves CUs idle (csrc/fused.hip, n_warmers)
static int n_warmers32(const PhaseDims& d, int M) {
    if (d.B / M > 64) return 0;
    static const int n = [] {
        const char* e = getenv("D4PG_L2_WARMERS");
        // measured per update at batch 256: 0: 84.8 us, 32: 89.5 (the warmers themselves, which miss in L2 by
        // design, then end the launch 6 us after the last row tile), 64: 80.9, 128: 81.3
        const int v = e ? atoi(e) : 64;
        return v < 0 ? 0 : (v > 256 ? 256 : (v + 7) & ~7);
    }();
    return n;
}

#ifdef D4PG_STAMPS       // one translation unit: only the 400-wide kernels are instantiated
#define DISPATCH_PHASE32(LAUNCHER, LDSFN)                                                   \
    const int lds = LDSFN(d, M, mt);                                                        \
    const int grid = d.B / M + extra_blocks;                                                \
    if (ksh == 13) return LAUNCHER<13>(a, mt, ksn, grid, lds, stream);                      \
    return -10;
#else
#define DISPATCH_PHASE32(LAUNCHER, LDSFN)                                                   \
    const int lds = LDSFN(d, M, mt);                                                        \
    const int grid = d.B / M + extra_blocks;                                                \
    D4PG_KSH_LIST(D4PG_DISPATCH_ONE_##LAUNCHER)                                             \
    return -10;
#define D4PG_DISPATCH_ONE_launch_critic32_ksh(K) if (ksh == K) return launch_critic32_ksh<K>(a, mt, ksn, grid, lds, stream);
#define D4PG_DISPATCH_ONE_launch_actor32_ksh(K) if (ksh == K) return launch_actor32_ksh<K>(a, mt, ksn, grid, lds, stream);
#define D4PG_DISPATCH_ONE_launch_infer32_ksh(K) if (ksh == K) return launch_infer32_ksh<K>(a, mt, ksn, grid, lds, stream);
#endif
D4PG_API int d4pg_phase32_has_width(int ksh) {
#ifndef D4PG_STAMPS
#define D4PG_HAS_ONE(K) if (ksh == K) return 1;
    D4PG_KSH_LIST(D4PG_HAS_ONE)
    return 0;
#else
    return ksh == 13;
#endif
}

D4PG_API int d4pg_critic_phase32(const CriticPhaseArgs32* a, int mt, hipStream_t stream) {
    const PhaseDims& d = a->d;
    if (mt != 1 && mt != 2 && mt != 4) return -9;
    const int M = 4 * mt;
    if (int rc = check_dims32(d, M)) return rc;
    const int ksh = hidden_chunks(d);
    const int ksn = max2i(max2i(a->ta[0].nch, a->tc[0].nch), max2i(max2i(a->cf[0].nch, a->cb3.nch), a->fwd.af[0].nch)) <= 2 ? 2 : 4;
    const LayerW32* nets[4] = {a->ta, a->tc, a->cf, a->fwd.af};
    for (int i = 0; i < 4; ++i) {
        if (int rc = check_layer32(nets[i][0], 2, ksh, ksn, d.ldx)) return rc;
        if (int rc = check_layer32(nets[i][1], 0, ksh, ksn, d.ldh)) return rc;
        if (int rc = check_layer32(nets[i][2], 1, ksh, ksn, d.ldh)) return rc;
    }
    if (int rc = check_layer32(a->cb3, 2, ksh, ksn, d.ldd)) return rc;
    if (int rc = check_layer32(a->cb2, 0, ksh, ksn, d.ldh)) return rc;
    if (a->fwd.g_xin == nullptr || a->fwd.g_h1 == nullptr || a->fwd.g_h2 == nullptr || a->fwd.sT == nullptr ||
        a->fwd.h1T == nullptr || a->fwd.h2T == nullptr)
        return -14;
    if (int prc = check_packed(a->pk, d)) return prc;
    if (a->ahead.mode != 0 && (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr)) return -15;
    if (a->ahead.role_map != 0) return -18;      // (the lean build only)
    const int extra_blocks = d.B / M + n_warmers32(d, M);      // the actor forward pass rides along, then the L2 warmers
    DISPATCH_PHASE32(launch_critic32_ksh, lds_bytes_critic32)
}

D4PG_API int d4pg_actor_phase32(const ActorPhaseArgs32* a, int mt, hipStream_t stream) {
    const PhaseDims& d = a->d;
    if (mt != 1 && mt != 2 && mt != 4) return -9;
    const int M = 4 * mt;
    if (int rc = check_dims32(d, M)) return rc;
    const int ksh = hidden_chunks(d);
    const int ksn = max2i(a->cf[0].nch, max2i(a->cb3.nch, a->ab3.nch)) <= 2 ? 2 : 4;
    if (int rc = check_layer32(a->cf[0], 2, ksh, ksn, d.ldx)) return rc;
    if (int rc = check_layer32(a->cf[1], 0, ksh, ksn, d.ldh)) return rc;
    if (int rc = check_layer32(a->cf[2], 1, ksh, ksn, d.ldh)) return rc;
    if (int rc = check_layer32(a->cb3, 2, ksh, ksn, d.ldd)) return rc;
    if (int rc = check_layer32(a->cb2, 0, ksh, ksn, d.ldh)) return rc;
    if (int rc = check_layer32(a->cb1, 1, ksh, ksn, d.ldh)) return rc;
    if (int rc = check_layer32(a->ab3, 2, ksh, ksn, d.lda_)) return rc;
    if (int rc = check_layer32(a->ab2, 0, ksh, ksn, d.ldh)) return rc;
    if (a->g_xin == nullptr || a->g_h1 == nullptr || a->g_h2 == nullptr) return -14;
    if (a->tree.n < 0 || (a->tree.n > 0 && (a->tree.idx == nullptr || a->tree.prio == nullptr || a->tree.t.L < 0 ||
                                            a->tree.t.L > MAXL || a->tree.parts > 64 ||
                                            (a->tree.parts > 1 && (a->tree.t.L < 2 || a->tree.ticket == nullptr)))))
        return -11;
    if (a->ahead.mode != 0) {
        if (a->ahead.mode < 0 || a->ahead.mode > 3 || (a->ahead.mode != 3 && !a->pk.enabled)) return -15;
        if (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr || a->c_saT == nullptr ||
            a->c_h1T == nullptr || a->c_h2T == nullptr)
            return -15;
        if (a->ahead.mode == 3) {      // the batch of the next update is drawn inside the launch: the tree must be
                                       // complete when it starts (its update rides in the launch before, d4pg_wgrad32)
            if (a->smp == nullptr || a->pk.st == nullptr || a->pk.enabled || a->tree.n > 0) return -16;
        } else if (int prc = check_packed(a->pk, d)) {
            return prc;
        }
    }
    if (a->ahead.role_map != 0) return -18;      // (the lean build only)
    const int n_ahead = a->ahead.mode != 0 ? d.B / M : 0;
    const int extra_blocks = (a->tree.n > 0 ? (a->tree.parts > 1 ? a->tree.parts : 1) : 0) + n_ahead + n_warmers32(d, M);
    DISPATCH_PHASE32(launch_actor32_ksh, lds_bytes_actor32)
}


// Batched actor inference (one launch).  mt: rows per workgroup / 4 (1, 2 or 4).
D4PG_API int d4pg_actor_infer32(const ActorInferArgs32* a, int mt, hipStream_t stream) {
    if (mt != 1 && mt != 2 && mt != 4) return -9;
    if (a->E < 1 || a->S < 1 || a->A < 1 || a->A > 64 || a->obs == nullptr || a->act == nullptr) return -1;
    if ((a->ldx & 3) || (a->ldh & 3) || a->ldx < a->S || a->ld_obs < a->S || a->ld_act < a->A) return -3;
    if (a->ou.x != nullptr && a->ou.counter == nullptr) return -4;
    const int M = 4 * mt;
    const int ksh = (a->H + 31) / 32;
    const int ksn = a->l[0].nch <= 2 ? 2 : 4;
    if (int rc = check_layer32(a->l[0], 2, ksh, ksn, a->ldx)) return rc;
    if (int rc = check_layer32(a->l[1], 0, ksh, ksn, a->ldh)) return rc;
    if (int rc = check_layer32(a->l[2], 1, ksh, ksn, a->ldh)) return rc;
    auto al = [](int b) { return (b + 15) & ~15; };
    const int lds = al(M * a->ldx * 4) + 2 * al(M * a->ldh * 4) + al((NW32 - 1) * mt * 256 * 4);
    const int grid = (a->E + M - 1) / M;
#ifdef D4PG_STAMPS
    if (ksh == 13) return launch_infer32_ksh<13>(a, mt, ksn, grid, lds, stream);
    return -10;
#else
    D4PG_KSH_LIST(D4PG_DISPATCH_ONE_launch_infer32_ksh)
    return -10;
#endif
}

// ---- the 16x16x4 path: any widths at run time (csrc/mlp16_dev.h) ---------------------------------
// `ld` = floats readable per row of the LDS tile the layer multiplies
static int check_layer16(const LayerW32& L, int ld) {
    if (L.W == nullptr || L.bias == nullptr || (((uintptr_t)L.W) & 15)) return -6;
    if (L.ntiles < 1 || L.ks < 1 || L.ksplit < 1 || L.nch < 1 || L.N < 1 || L.N > 16 * L.ntiles) return -6;
    if (L.ksplit > 1 && L.ntiles * L.ksplit > NW32) return -7;                       // one K-slice of one tile per wave
    if (L.nch * L.ksplit < L.ks || (L.ksplit - 1) * L.nch >= L.ks) return -7;        // the slices cover K, none is empty
    if (L.ks * 16 > ld) return -7;
    if ((long long)L.ntiles * L.ks * GROUP16_BYTES >= (1ll << 31)) return -7;
    return 0;
}
static int check_dims16(const PhaseDims& d, int M) {
    if (int rc = check_dims32(d, M)) return rc;
    if (d.ldh < ((d.H + 15) & ~15) || d.ldd < ((d.NO + 15) & ~15) || d.lda_ < ((d.A + 15) & ~15) ||
        d.ldx < ((d.S + d.A + 15) & ~15))
        return -3;
    return 0;
}

// LDS a workgroup may ask for while two of them still fit a CU (160 KB, allocated in units of 1280 bytes)
constexpr int LEAN_LDS_LIMIT = 81920;

static int critic_phase32w(const CriticPhaseArgs32* a, int mt, bool lean, hipStream_t stream) {
    const PhaseDims& d = a->d;
    if (mt != 2 && mt != 4) return -9;
    if (lean && mt != 4) return -17;      // (beyond LEAN_LDS_LIMIT it runs as one workgroup per CU; launch_phase bounds the size)
    const int M = 4 * mt;
    if (int rc = check_dims16(d, M)) return rc;
    const LayerW32* nets[4] = {a->ta, a->tc, a->cf, a->fwd.af};
    for (int i = 0; i < 4; ++i) {
        if (int rc = check_layer16(nets[i][0], d.ldx)) return rc;
        if (int rc = check_layer16(nets[i][1], d.ldh)) return rc;
        if (int rc = check_layer16(nets[i][2], d.ldh)) return rc;
    }
    if (int rc = check_layer16(a->cb3, d.ldd)) return rc;
    if (int rc = check_layer16(a->cb2, d.ldh)) return rc;
    if (a->fwd.g_xin == nullptr || a->fwd.g_h1 == nullptr || a->fwd.g_h2 == nullptr || a->fwd.sT == nullptr ||
        a->fwd.h1T == nullptr || a->fwd.h2T == nullptr)
        return -14;
    if (int prc = check_packed(a->pk, d)) return prc;
    if (a->ahead.mode != 0 && (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr)) return -15;
    if (a->ahead.role_map != 0 && (!lean || a->ahead.role_map != 1 || (d.B / M) % 256)) return -18;
    const int grid = 2 * (d.B / M) + n_warmers32(d, M);
    if (lean) return launch_critic32_lean(a, grid, lds_bytes_critic32(d, M, mt, true, true), stream);
    return launch_critic32_wide(a, mt, grid, lds_bytes_critic32(d, M, mt, true), stream);
}
D4PG_API int d4pg_critic_phase32w(const CriticPhaseArgs32* a, int mt, hipStream_t stream) { return critic_phase32w(a, mt, false, stream); }
D4PG_API int d4pg_critic_phase32wl(const CriticPhaseArgs32* a, int mt, hipStream_t stream) { return critic_phase32w(a, mt, true, stream); }

static int actor_phase32w(const ActorPhaseArgs32* a, int mt, bool lean, hipStream_t stream) {
    const PhaseDims& d = a->d;
    if (mt != 2 && mt != 4) return -9;
    if (lean && mt != 4) return -17;
    const int M = 4 * mt;
    if (int rc = check_dims16(d, M)) return rc;
    if (int rc = check_layer16(a->cf[0], d.ldx)) return rc;
    if (int rc = check_layer16(a->cf[1], d.ldh)) return rc;
    if (int rc = check_layer16(a->cf[2], d.ldh)) return rc;
    if (int rc = check_layer16(a->cb3, d.ldd)) return rc;
    if (int rc = check_layer16(a->cb2, d.ldh)) return rc;
    if (int rc = check_layer16(a->cb1, d.ldh)) return rc;
    if (int rc = check_layer16(a->ab3, d.lda_)) return rc;
    if (int rc = check_layer16(a->ab2, d.ldh)) return rc;
    if (a->g_xin == nullptr || a->g_h1 == nullptr || a->g_h2 == nullptr) return -14;
    if (a->tree.n < 0 || (a->tree.n > 0 && (a->tree.idx == nullptr || a->tree.prio == nullptr || a->tree.t.L < 0 ||
                                            a->tree.t.L > MAXL || a->tree.parts > 64 ||
                                            (a->tree.parts > 1 && (a->tree.t.L < 2 || a->tree.ticket == nullptr)))))
        return -11;
    if (a->ahead.mode != 0) {
        if (a->ahead.mode < 0 || a->ahead.mode > 3 || (a->ahead.mode != 3 && !a->pk.enabled)) return -15;
        if (a->ahead.h1 == nullptr || a->ahead.h2 == nullptr || a->ahead.lg == nullptr || a->c_saT == nullptr ||
            a->c_h1T == nullptr || a->c_h2T == nullptr)
            return -15;
        if (a->ahead.mode == 3) {
            if (a->smp == nullptr || a->pk.st == nullptr || a->pk.enabled || a->tree.n > 0) return -16;
        } else if (int prc = check_packed(a->pk, d)) {
            return prc;
        }
    }
    if (a->ahead.role_map != 0 && (!lean || a->ahead.role_map != 1 || (d.B / M) % 256 || a->tree.n > 0)) return -18;
    const int n_ahead = a->ahead.mode != 0 ? d.B / M : 0;
    const int grid = d.B / M + (a->tree.n > 0 ? (a->tree.parts > 1 ? a->tree.parts : 1) : 0) + n_ahead + n_warmers32(d, M);
    if (lean) return launch_actor32_lean(a, grid, lds_bytes_actor32(d, M, mt, true, true), stream);
    return launch_actor32_wide(a, mt, grid, lds_bytes_actor32(d, M, mt, true), stream);
}
D4PG_API int d4pg_actor_phase32w(const ActorPhaseArgs32* a, int mt, hipStream_t stream) { return actor_phase32w(a, mt, false, stream); }
D4PG_API int d4pg_actor_phase32wl(const ActorPhaseArgs32* a, int mt, hipStream_t stream) { return actor_phase32w(a, mt, true, stream); }

D4PG_API int d4pg_actor_infer32w(const ActorInferArgs32* a, int mt, hipStream_t stream) {
    if (mt != 2 && mt != 4) return -9;
    if (a->E < 1 || a->S < 1 || a->A < 1 || a->A > 64 || a->obs == nullptr || a->act == nullptr) return -1;
    if ((a->ldx & 3) || (a->ldh & 3) || a->ldx < a->S || a->ld_obs < a->S || a->ld_act < a->A) return -3;
    if (a->ou.x != nullptr && a->ou.counter == nullptr) return -4;
    const int M = 4 * mt;
    if (int rc = check_layer16(a->l[0], a->ldx)) return rc;
    if (int rc = check_layer16(a->l[1], a->ldh)) return rc;
    if (int rc = check_layer16(a->l[2], a->ldh)) return rc;
    auto al = [](int b) { return (b + 15) & ~15; };
    const int lds = al(M * a->ldx * 4) + 2 * al(M * a->ldh * 4) + al(red_floats32(mt, true) * 4);
    return launch_infer32_wide(a, mt, (a->E + M - 1) / M, lds, stream);
}

D4PG_API int d4pg_phase32w_lds_bytes(const PhaseDims* d, int mt, int actor) {
    return actor ? lds_bytes_actor32(*d, 4 * mt, mt, true) : lds_bytes_critic32(*d, 4 * mt, mt, true);
}
// the lean build (16-row tiles); d4pg_phase32wl_lds_limit: what two workgroups per CU allow
D4PG_API int d4pg_phase32wl_lds_bytes(const PhaseDims* d, int actor) {
    return actor ? lds_bytes_actor32(*d, 16, 4, true, true) : lds_bytes_critic32(*d, 16, 4, true, true);
}
D4PG_API int d4pg_phase32wl_lds_limit() { return LEAN_LDS_LIMIT; }

D4PG_API int d4pg_phase32_lds_bytes(const PhaseDims* d, int mt, int actor) {
    return actor ? lds_bytes_actor32(*d, 4 * mt, mt) : lds_bytes_critic32(*d, 4 * mt, mt);
}
D4PG_API int d4pg_phase32_waves() { return NW32; }

// ---------------------------------------------------------------------------------------
// Weight gradients:  dW[o][i] = sum_b dY[b][o] * X[b][i],  db[o] = sum_b dY[b][o]
// All layers of a network in one launch, 32x32 tiles on v_mfma_f32_32x32x2_f32; NWV waves split the
// batch and combine through LDS, then ALL of them finish the tile: Adam, Polyak and the refresh of
// the packed fp32 copies (forward: packed32_off(o, i); transposed: packed32_off(i - c0, o)) when the
// optimiser is fused, or the plain gradient otherwise.
// ---------------------------------------------------------------------------------------
struct WgradDesc32 {
    const float* AT;    // dY stash (stash32_off), rowsA features x Bp batch rows
    const float* BT;    // X stash, rowsB features x Bp batch rows
    float* C;           // dW fp32 [M][ldc]
    float* rowsum;      // db fp32 [M] or null
    int M, N;
    int rowsA, rowsB;
    int ldc;
    int tiles_m, tiles_n, tile_base;
    long long w_off, b_off;        // offsets (floats) of W and bias in the flat parameter buffer
    long long sf_off, sb_off;      // forward / transposed packed-copy offsets (floats), -1: none
    int sf_ks, sb_ks, sb_c0, sb_c1;
};
struct WgradGroup32 {
    WgradDesc32 p[3];
    int count, total_tiles;
    int Bp;
    int wide;                  // packed copies in the order of csrc/mlp16_dev.h (packed16_off, sf_ks / sb_ks = 16-wide k-groups)
    WgradAdam opt;
    const SampleArgs* smp;     // prioritised replay: the next batch is drawn by extra workgroups (csrc/fused.hip)
    int smp_rows;
    int tree_parts;            // > 0: that many more workgroups write the update's priorities into the tree (*tree)
    const TreeUpdateArgs* tree;
    // Gradient-exchange mode: 1 = the gradients are already in C / rowsum (written by a launch without optimiser, summed
    // over the ranks by the all-reduce between the two launches); every tile only APPLIES them: Adam, Polyak, packed
    // copies, with the tile-shaped stores of the fused optimiser (a flat element-wise kernel scatters 4-byte stores over
    // the packed copies and takes 10 us for a 190 k-parameter network; this launch takes what the epilogue takes).
    int apply_only;
    int pad_;
};

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void wgrad32_kernel(const WgradGroup32 g) {
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
        if (b < sa.B && size > 0) sample_row(sa, b, sa.st->step_cur, size, threadIdx.x & 63);
        return;
    }
    const int xcd = id & 7, local = id >> 3;
    const int q = total >> 3, rem = total & 7;
    const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    int p = 0;
#pragma unroll
    for (int i = 1; i < 3; ++i)
        if (i < g.count && tile >= g.p[i].tile_base) p = i;
    const WgradDesc32& d = g.p[p];
    const int t = tile - d.tile_base;
    const int tm = t / d.tiles_n, tn = t - tm * d.tiles_n;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool m_ok = tm * 32 + r < d.M, n_ok = tn * 32 + r < d.N;
    const int groups = g.Bp >> 3;              // groups of 8 batch rows: one 16-byte load per operand = 4 MFMAs
    const float* arow = d.AT + ((size_t)tm * groups * 64 + lane) * 4;
    const float* brow = d.BT + ((size_t)tn * groups * 64 + lane) * 4;
    constexpr int EPT = 16 / NWV;              // accumulator registers (rows of column pn) finished per wave
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
    const int per = (groups + NWV - 1) / NWV;
    const int g0 = wave * per, g1 = min(groups, g0 + per);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float asum = 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool want_rowsum = d.rowsum != nullptr && tn == 0;
    // Software pipeline over the wave's groups of 8 batch rows: WG_DEPTH operand pairs in flight.  (Left to the
    // compiler the loop is not unrolled and every iteration waits for its own two loads: with the stashes of a
    // 4096-row batch coming from beyond L2 the matrix pipe was 43 % busy.)  Loads past the wave's range re-read its
    // last group and are not multiplied.  (Deeper does not help: 8 waves with 8, 12 or 16 pairs in flight each took
    // 220 - 222 us per Ant update of 4096 rows where 16 waves with 4 take 216.6, bench/exp/r3_wgrad_depth.sh.)
    constexpr int WG_DEPTH = 4;
    f32x4 avq[WG_DEPTH], bvq[WG_DEPTH];
    const bool apply_only = g.apply_only != 0;      // (uniform over the launch)
    if (apply_only) {
    } else if (g1 - g0 <= 2) {      // (workgroup-uniform) small batches: one or two groups per wave, nothing to pipeline
        for (int gi = g0; gi < g1; ++gi) {
            f32x4 av = *reinterpret_cast<const f32x4*>(arow + gi * 256);
            f32x4 bv = *reinterpret_cast<const f32x4*>(brow + gi * 256);
            av = m_ok ? av : zero;
            bv = n_ok ? bv : zero;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
            if (want_rowsum) asum += (av[0] + av[1]) + (av[2] + av[3]);
        }
    } else {
#pragma unroll
        for (int s = 0; s < WG_DEPTH; ++s) {
            const int gl = min(g0 + s, g1 - 1);
            avq[s] = *reinterpret_cast<const f32x4*>(arow + gl * 256);
            bvq[s] = *reinterpret_cast<const f32x4*>(brow + gl * 256);
        }
        for (int gi = g0; gi < g1; gi += WG_DEPTH) {
            const bool more = gi + WG_DEPTH < g1;      // wave-uniform: a next block of WG_DEPTH groups exists
#pragma unroll
            for (int s = 0; s < WG_DEPTH; ++s) {
                f32x4 av = avq[s], bv = bvq[s];
                if (more) {
                    const int gl = min(gi + s + WG_DEPTH, g1 - 1);
                    avq[s] = *reinterpret_cast<const f32x4*>(arow + gl * 256);
                    bvq[s] = *reinterpret_cast<const f32x4*>(brow + gl * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (gi + s < g1) {      // wave-uniform
                    av = m_ok ? av : zero;
                    bv = n_ok ? bv : zero;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
                    if (want_rowsum) asum += (av[0] + av[1]) + (av[2] + av[3]);
                }
            }
        }
    }
    if (!apply_only) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
        if (want_rowsum) {
            asum += __shfl_xor(asum, 32, 64);
            if (lane < 32) cs[wave * 32 + lane] = asum;
        }
        __syncthreads();
    }
    if (fuse) {
        step_size = o.lr / step_size;
        inv_sqrt_bc2 = 1.f / sqrtf(inv_sqrt_bc2);
    }
    const int cn = pn;
    float* ps = (float*)o.p_shadow;
    float* ts = (float*)o.t_shadow;
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
        if (cm < d.M && cn < d.N) {
            if (!fuse) {
                d.C[(size_t)cm * d.ldc + cn] = gv;       // read by the gradient all-reduce
            } else {
                float tv;
                const float pv = adam_apply(o, d.w_off + (long long)cm * d.ldc + cn, ast[i], gv, step_size, inv_sqrt_bc2, tv);
                if (d.sf_off >= 0) {
                    const long long so = d.sf_off + (g.wide ? packed16_off(cm, cn, d.sf_ks) : packed32_off(cm, cn, d.sf_ks));
                    ps[so] = pv;
                    ts[so] = tv;
                }
                if (d.sb_off >= 0 && cn >= d.sb_c0 && cn < d.sb_c1)
                    ps[d.sb_off + (g.wide ? packed16_off(cn - d.sb_c0, cm, d.sb_ks) : packed32_off(cn - d.sb_c0, cm, d.sb_ks))] = pv;
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
}

template <typename K>
static int launch_wgrad32(K kernel, int nwv, int grid, const WgradGroup32& g, hipStream_t stream) {
    const int lds = nwv * (16 * 64 + 32) * 4;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
    }
    int extra = (g.opt.p != nullptr && g.opt.finish_step) ? 1 : 0;       // wgrad_finish_role
    if (g.smp != nullptr) {
        if (!extra || g.smp_rows < 1) return -5;
        extra += (g.smp_rows + nwv - 1) / nwv;
    }
    if (g.tree_parts < 0 || g.tree_parts > 64 || (g.tree_parts > 0 && g.tree == nullptr)) return -6;
    extra += g.tree_parts;
    hipLaunchKernelGGL(kernel, dim3(grid + extra), dim3(64 * nwv), lds, stream, g);
    return (int)hipGetLastError();
}

D4PG_API int d4pg_wgrad32(WgradGroup32* g, hipStream_t stream) {
    if (g->count < 1 || g->count > 3 || g->Bp < 32 || (g->Bp & 31)) return -1;
    int base = 0;
    for (int i = 0; i < g->count; ++i) {
        WgradDesc32& d = g->p[i];
        if (d.M < 1 || d.N < 1 || d.ldc < d.N) return -2;
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
    int nwv = g->Bp >= 128 ? 16 : (g->Bp >= 64 ? 8 : 4);     // one group of 8 batch rows per wave up to 16 waves
    static const int forced_waves = [] { const char* e = getenv("D4PG_WGRAD_WAVES"); return e ? atoi(e) : 0; }();      // (A/B runs)
    if (forced_waves) nwv = forced_waves;
    if (nwv == 16) return launch_wgrad32(wgrad32_kernel<16>, 16, base, *g, stream);
    if (nwv == 8) return launch_wgrad32(wgrad32_kernel<8>, 8, base, *g, stream);
    return launch_wgrad32(wgrad32_kernel<4>, 4, base, *g, stream);
}
D4PG_SIZEOF(LayerW32)
D4PG_SIZEOF(ActorFwd32)
D4PG_SIZEOF(CriticPhaseArgs32)
D4PG_SIZEOF(ActorPhaseArgs32)
D4PG_SIZEOF(WgradDesc32)
D4PG_SIZEOF(WgradGroup32)
D4PG_SIZEOF(ActorInferArgs32)
D4PG_SIZEOF(OuFuse32)
#endif   // D4PG_MAIN_UNIT

// One dense layer of the row-local fp32 pipeline on a tile of 4 * MT batch rows held in LDS (gfx950).
//
// Arithmetic: v_mfma_f32_4x4x1_16b_f32, the 16-block form of the f32-input MFMA.  One instruction
// multiplies sixteen independent 4x1 by 1x4 blocks; here every block gets the SAME four batch rows
// as its A operand (lane l holds x[l & 3][k]) and its own four output columns as B (lane l holds
// W[col = 64 * tile + l][k]), so one instruction is a rank-1 update of a 4-row x 64-column tile
// and its result D[row r][col l] sits in register r of lane l.  The product of every pair is
// rounded once and accumulated in f32, exactly like an fmaf chain: this is reference (fp32)
// precision, at the f32 vector rate (64 FLOP / clk / SIMD).
//
// Why 4-row tiles: a layer is bound either by the MFMA pipe of the one CU that owns the row tile
// (proportional to the rows) or by streaming the layer's weights through that CU (independent of
// the rows).  At 16 rows (v_mfma_f32_16x16x4_f32) a 400 x 400 layer is 9 us of MFMA issue; at 4
// rows it is 2.3 us and the layer runs at the weight-streaming rate of a CU instead, while four
// times as many CUs are busy.  MT = 2 / 4 gives 8 / 16-row tiles for batches that fill the chip.
//
// Weights are read from a packed copy in "fragment order": element (n, k) of a B operand
// (n = output column, k = reduction index) lives at packed32_off(n, k, ks), i.e.
// [n / 64][k / 4][n % 64][k % 4].  The 64 lanes x 16 bytes of one wave-wide load are one contiguous
// 1 KB block and give every lane four consecutive k of its column: four MFMAs per load.
// K is padded to a multiple of 32 (a "chunk" of 8 such loads = 8 KB per wave, ks chunks per
// column tile); pad entries are zero and the LDS tiles carry zero pad columns.
//
// The loads are buffer loads (a buffer descriptor with the size of the packed copy): a lane
// whose column does not exist (the last tile of a 400-wide layer has 16 of 64 columns) gets an
// out-of-range offset, for which the hardware returns zeros without touching memory, so the
// streaming code has no branches and the compiler can count the loads in flight exactly (any
// load behind a run-time condition makes it wait for everything, and a layer is bound by how many
// weight loads it keeps in flight).
#pragma once
#include "common.h"

// Diagnostic build (-DD4PG_STAMPS): four clock stamps inside one layer of workgroup 0 (slots lst .. lst + 3 of the
// critic phase's row: entry, own MFMAs before `pre`, `pre` issued, tail + epilogue), for bench/stamps.py.
#ifdef D4PG_STAMPS
extern __device__ unsigned long long g_stamps32[2][32];
#define LSTAMP(i)                                                                                   \
    do {                                                                                            \
        if (lst >= 0 && blockIdx.x == 0 && threadIdx.x == 0) g_stamps32[0][lst + (i)] = wall_clock64(); \
    } while (0)
#else
#define LSTAMP(i)
#endif

constexpr int KCH = 8;                          // 16-byte k-groups per chunk (32 k)
constexpr int CHUNK_FLOATS = KCH * 64 * 4;      // one chunk of one column tile: 8 KB
constexpr int CHUNK_BYTES = CHUNK_FLOATS * 4;

__host__ __device__ __forceinline__ long long packed32_off(int n, int k, int ks) {
    return (((long long)(n >> 6) * (ks * KCH) + (k >> 2)) << 8) + ((n & 63) << 2) + (k & 3);
}

struct LayerW32 {
    const float* W;      // packed copy (see above), ntiles * ks * CHUNK_FLOATS floats
    const float* bias;   // fp32 [N]; a zero vector for bias-free (backward) layers, never null
    int ks;              // chunks (32-wide K-steps) per column tile; a multiple of ksplit
    int ntiles;          // column tiles of 64
    int N;               // true number of outputs
    int ksplit;          // K-slices of the column tile; > 1 only for single-tile layers (N <= 64)
    int nch;             // chunks per unit = ks / ksplit
    int kg;              // 16-byte k-groups of a unit that hold weights (= ceil(K / 4) for layers whose K is
                         // not the hidden width: the rest of their chunks is padding and is not loaded)
};
// Work is dealt to the waves in units, one per wave: wave w of a layer with ksplit == 1 owns column
// tile w; wave w of a single-tile layer owns K-slice w = chunks [w * n, (w + 1) * n), n = ks / ksplit.

struct Chunk32 {
    f32x4 b[KCH];
};
// The first PF chunks of a wave's unit of some layer, fetched while the layer before it still runs: the
// stream of weight loads of a workgroup must not stop at a layer boundary (a layer is as fast as the
// loads it keeps in flight; with one chunk fetched ahead a boundary cost ~0.6 us of idle load pipe).
#ifndef D4PG_PF32
#define D4PG_PF32 2
#endif
constexpr int PF32 = D4PG_PF32;
struct Pre32 {
    Chunk32 c[PF32];
    float bias;
};

// Per-lane view of the wave's unit: descriptor + this lane's byte offset inside a chunk (out of range
// for lanes without a column and for waves without a unit) + the scalar byte offset of the unit.
struct WSrc32 {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
    int soff;
};
__device__ __forceinline__ WSrc32 wsrc_of(const LayerW32& L, int wave, int lane) {
    const int bytes = L.ntiles * L.ks * CHUNK_BYTES;
    const bool split = L.ksplit > 1;
    const bool has = wave < (split ? L.ksplit : L.ntiles);
    const int tile = split ? 0 : wave;
    WSrc32 s;
    s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)L.W, 0, bytes, 0x00020000);
    s.voff = (has && tile * 64 + lane < L.N) ? lane * 16 : bytes;
    // unit of wave w: chunks [w * nch, (w + 1) * nch) of the packed copy (nch = ks when the tiles are dealt out)
    s.soff = has ? wave * L.nch * CHUNK_BYTES : 0;
    return s;
}
__device__ __forceinline__ Chunk32 load_chunk(const WSrc32& s, int chunk) {
    Chunk32 c;
#pragma unroll
    for (int g = 0; g < KCH; ++g)
        c.b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff,
                                                                                s.soff + chunk * CHUNK_BYTES + g * 1024, 0));
    // The loads stay HERE in the instruction stream: left alone, the scheduler sinks every load to
    // just ahead of its first use (fewer live registers), which leaves one load in flight per wave.
    __builtin_amdgcn_sched_barrier(0);
    return c;
}

// What a layer costs is the bytes its CU pulls through the vector cache (64 B / clk: 640 KB of a 400 x 400
// layer = 4.2 us, 5.3 us measured), so k-padding (real zeros in memory) is not loaded: see the run-time
// variants below.  Lanes that are out of range return zeros without a fetch and cost nothing measurable:
// storing the 16-column last tile of a 400-wide layer so that its wave issues a quarter of the loads
// (bench/exp/notes/rem16_tile_experiment.patch) left the update time unchanged (77.4 vs 77.1 us).

// The first min(PF32, NCH) chunks of the wave's unit, unconditionally (a wave without a unit reads zeros
// from out-of-range offsets): the consumer's load counting stays exact.
template <int NCH>
__device__ __forceinline__ Pre32 fetch_first(const LayerW32& L, int wave, int lane) {
    Pre32 t;
    const int col = (L.ksplit > 1 ? 0 : wave * 64) + lane;
    t.bias = L.bias[col < L.N ? col : 0];
    const WSrc32 src = wsrc_of(L, wave, lane);
#pragma unroll
    for (int j = 0; j < PF32; ++j)
        if (j < NCH) t.c[j] = load_chunk(src, j);
    return t;
}
// Run-time variant for layers whose K is an input / action / atom count: only the k-groups that hold
// weights are loaded (K = 8 is 2 of the 8 groups of a chunk).  Groups that are not loaded stay undefined
// and are not multiplied.
__device__ __forceinline__ void load_groups(Chunk32& c, const WSrc32& s, int chunk, int kg) {
#pragma unroll
    for (int g = 0; g < KCH; ++g)
        if (chunk * KCH + g < kg)      // wave-uniform
            c.b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.voff,
                                                                                    s.soff + chunk * CHUNK_BYTES + g * 1024, 0));
    __builtin_amdgcn_sched_barrier(0);
}
// (kg_of_layer: L.kg, or 0 for a layer that is not going to run: nothing is loaded then)
__device__ __forceinline__ Pre32 fetch_first_rt(const LayerW32& L, int kg_of_layer, int wave, int lane) {
    Pre32 t;
    const int col = wave * 64 + lane;
    t.bias = L.bias[col < L.N ? col : 0];
    const WSrc32 src = wsrc_of(L, wave, lane);
    const int kg = wave < L.ntiles ? kg_of_layer : 0;
#pragma unroll
    for (int j = 0; j < PF32; ++j) load_groups(t.c[j], src, j, kg);
    return t;
}

// acc[mt][j & 1] += x[4 mt + (lane & 3)][k] * W[col][k] over the 32 k of one chunk.  Two
// accumulators per row group: consecutive MFMAs never wait for each other's result.
template <int MT>
__device__ __forceinline__ void mma_chunk(const Chunk32& c, const float* Arow, int lda, f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int g = 0; g < KCH; ++g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Arow + mt * 4 * lda + 4 * g);
            acc[mt][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], c.b[g][0], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], c.b[g][1], acc[mt][1], 0, 0, 0);
            acc[mt][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], c.b[g][2], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], c.b[g][3], acc[mt][1], 0, 0, 0);
        }
    }
}

// the same for the first `ng` k-groups of a chunk only (run-time K)
template <int MT>
__device__ __forceinline__ void mma_groups(const Chunk32& c, int ng, const float* Arow, int lda, f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int g = 0; g < KCH; ++g) {
        if (g < ng) {       // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Arow + mt * 4 * lda + 4 * g);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], c.b[g][0], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], c.b[g][1], acc[mt][1], 0, 0, 0);
                acc[mt][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], c.b[g][2], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], c.b[g][3], acc[mt][1], 0, 0, 0);
            }
        }
    }
}

// The wave's unit has exactly NCH chunks (= L.nch, checked by the host), known at compile time:
// the streaming loop is straight-line code with three chunk buffers in rotation (the chunk being
// multiplied + two in flight; measured on one CU, a 400 x 400 layer on a 4-row tile: 6.1 us with
// three, 6.5 with four, 6.9 with five buffers, 5.3 us for the loads alone, which is the rate at which
// one CU is served by its L2).
// RT = true: the unit has L.kg <= 8 * NCH k-groups, known at run time (layers whose K is an input,
// action or atom count).  Everything is loaded up front, group by group, and only what was loaded is
// multiplied; `pre` follows the multiplications (the waits of this variant are for all loads in flight).
//
// epi(row0, col, v): v = rows row0 .. row0 + 3 of output column col (bias added), col < L.N.
// A   : LDS fp32 [4 * MT][lda], columns up to 32 * L.ks readable (zero beyond the true K)
// t   : the first PF32 chunks of this wave's unit (fetch_first*, issued by the layer before)
// red : LDS scratch for the K-slice partial sums, (NW - 1) * MT * 256 floats
// One unit per wave, so max(ntiles, ksplit) <= NW.  With ksplit > 1 the slices of the tile meet
// through `red` behind one workgroup barrier, and wave 0 finishes the tile (fixed summation order).
// `pre` issues the loads of the NEXT layer's first chunks, right behind this layer's own last load, so
// that the two streams join without a gap.  The streaming code of a wave with a unit is straight-line
// (every load unconditional, so the compiler's count of the loads in flight is exact); a wave without a
// unit only runs `pre`.
template <int NW, int MT, int NCH, bool RT, typename F, typename P>
__device__ __forceinline__ void layer32(Pre32& t, const float* A, int lda, const LayerW32& L, float* red, F&& epi,
                                        P&& pre, int lst = -1) {
    LSTAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool split = L.ksplit > 1;
    const bool has = wave < (split ? L.ksplit : L.ntiles);
    const int col = (split ? 0 : wave * 64) + lane;
    const int slice = split ? wave : 0;
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // `pre` has ONE call site, between the two halves of the streaming code (a second call site makes the
    // compiler route every kernel argument through scratch memory).
    const WSrc32 src = wsrc_of(L, wave, lane);
    const float* Al = A + (lane & 3) * lda + slice * L.nch * 32;
    constexpr int RING = RT ? NCH : (NCH < 3 ? NCH : 3);
    constexpr int HAVE = PF32 < RING ? PF32 : RING;                       // chunks that arrive in `t`
    constexpr int PRE_AT = (RT || NCH <= RING) ? (RT ? NCH : 0) : NCH - RING;   // chunks multiplied ahead of `pre`
    Chunk32 ring[RING];
    if (has) {
#pragma unroll
        for (int c = 0; c < HAVE; ++c) ring[c] = t.c[c];
        if constexpr (RT) {
#pragma unroll
            for (int c = HAVE; c < RING; ++c) load_groups(ring[c], src, c, L.kg);
        } else {
#pragma unroll
            for (int c = HAVE; c < RING; ++c) ring[c] = load_chunk(src, c);
        }
#pragma unroll
        for (int c = 0; c < PRE_AT; ++c) {
            if constexpr (RT) {
                mma_groups<MT>(ring[c], L.kg - c * KCH, Al + c * 32, lda, acc);
            } else {
                mma_chunk<MT>(ring[c % RING], Al + c * 32, lda, acc);
                if (c + RING < NCH) ring[c % RING] = load_chunk(src, c + RING);      // the last of them is this layer's last load
            }
        }
    }
    LSTAMP(1);
    pre();
    LSTAMP(2);
    if (has) {
        if constexpr (!RT) {
#pragma unroll
            for (int c = PRE_AT; c < NCH; ++c) mma_chunk<MT>(ring[c % RING], Al + c * 32, lda, acc);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] += acc[mt][1];
    }
    if (split) {       // workgroup-uniform
        if (has && slice != 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                *reinterpret_cast<f32x4*>(red + ((size_t)((slice - 1) * MT + mt) * 64 + lane) * 4) = acc[mt][0];
        }
        __syncthreads();
        if (wave != 0) return;
        for (int s = 1; s < L.ksplit; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt][0] += *reinterpret_cast<const f32x4*>(red + ((size_t)((s - 1) * MT + mt) * 64 + lane) * 4);
        }
    }
    if (!has || col >= L.N) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 v = acc[mt][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t.bias;
        epi(4 * mt, col, v);
    }
    LSTAMP(3);
}

// One dense layer of the row-local fp32 pipeline on a tile of 16 batch rows held in LDS, for batches
// that fill the chip and for hidden widths the 4-row kernels (csrc/mlp32_dev.h) are not compiled for.
//
// Arithmetic: v_mfma_f32_16x16x4_f32.  One instruction multiplies the 16 batch rows of the tile by a
// 16-column tile of the layer over 4 k: lane l = 16 q + c holds x[row c][k = q] as A and W[col c][k = q]
// as B, and its four results are rows 4 q .. 4 q + 3 of column c.  Exact f32 products and sums like the
// 4x4x1 form of csrc/mlp32_dev.h, at the same 64 FLOP / clk / SIMD, but 1024 multiply-adds per
// instruction instead of 256: the 4-row kernels spend as many issue cycles on the instructions BETWEEN
// their 8-cycle MFMAs (operand reads from LDS, waits) as on the MFMAs themselves once a tile has 8 or 16
// rows (rocprofv3 at batch 2048 / 4096: 1.4 - 1.9 vector instructions per MFMA, matrix pipe 44 % busy),
// while one 16-byte LDS read here feeds 4 x T MFMAs of 32 cycles each (T = column tiles of the wave).
//
// Weights are read from a packed copy in the fragment order of THIS instruction: element (n, k) lives at
// packed16_off(n, k, kgs) = [n / 16][k / 16][q = (k % 16) / 4][c = n % 16][k % 4], so the 64 lanes x 16
// bytes of one wave-wide load are one contiguous 1 KB block (one column tile x 16 k) and give every lane
// the operands of four MFMAs (the k order inside a group of 16 is permuted the same way in A and B,
// which a sum over k does not see).  K and N are padded to multiples of 16 with zeros; everything else
// (widths, tile counts, K-slices) is a run-time quantity, so one instantiation serves every shape.
#pragma once
#include "common.h"
#include "mlp32_dev.h"

constexpr int GROUP16_FLOATS = 256;        // one (16-column tile, 16-k group) block
constexpr int GROUP16_BYTES = 1024;
constexpr int TPW16 = 4;                   // column tiles a wave multiplies in one pass over K
constexpr int TPW16_LEAN = 2;              // ... in the kernels built for two workgroups per CU (half the registers per wave:
                                           // accumulators, weight ring and prefetch slots all scale with this number)

__host__ __device__ __forceinline__ long long packed16_off(int n, int k, int kgs) {
    return (((long long)(n >> 4) * kgs + (k >> 4)) << 8) + ((((k >> 2) & 3) << 4) + (n & 15)) * 4 + (k & 3);
}

// A layer of this path is described by a LayerW32 read as:  ks = kgs (16-wide k-groups), ntiles = column
// tiles of 16, N = outputs, ksplit = K-slices (> 1 only when ntiles * ksplit <= waves), nch = k-groups
// per slice; kg is not used.
//
// Work of wave w of NW:  ksplit == 1: column tiles w, w + NW, w + 2 NW, ... (TP of them per pass over K);
// ksplit > 1: tile w % ntiles, K-slice w / ntiles, the slices of a tile meeting through `red`.
template <int TP>
struct Grp16 {
    f32x4 b[TP];         // B fragments of one k-group for up to TP column tiles
};
template <int TP>
struct Pre16 {
    Grp16<TP> g[2];      // the first two k-groups of the wave's first pass
};
template <int TP>
struct Unit16 {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[TP];        // this lane's byte offset inside a group, out of range for tiles the wave does not own
    int soff[TP];        // scalar byte offset of the tile's first k-group
    int g0, g1;          // k-groups [g0, g1) of the wave's slice
    int tiles;           // column tiles of this pass (0: the wave idles)
    int tile0;           // the first of them; the others follow at a stride of NW (ksplit == 1)
    int oob;
};

template <int NW, int TP>
__device__ __forceinline__ Unit16<TP> unit16_of(const LayerW32& L, int wave, int lane, int pass) {
    Unit16<TP> u;
    const int bytes = L.ntiles * L.ks * GROUP16_BYTES;
    u.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)L.W, 0, bytes, 0x00020000);
    u.oob = bytes;
    const bool split = L.ksplit > 1;
    int slice = 0;
    if (split) {
        const bool has = pass == 0 && wave < L.ntiles * L.ksplit;
        slice = wave / L.ntiles;
        u.tile0 = wave - slice * L.ntiles;
        u.tiles = has ? 1 : 0;
    } else {
        u.tile0 = wave + pass * TP * NW;
        const int left = L.ntiles - u.tile0;
        u.tiles = left <= 0 ? 0 : min(TP, (left + NW - 1) / NW);
    }
    u.g0 = slice * L.nch;
    u.g1 = min(L.ks, u.g0 + L.nch);      // (never empty: the host checks (ksplit - 1) * nch < ks)
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int tile = u.tile0 + j * NW;
        u.voff[j] = j < u.tiles ? lane * 16 : bytes;
        u.soff[j] = j < u.tiles ? tile * L.ks * GROUP16_BYTES : 0;
    }
    return u;
}

// k-group g of the wave's tiles; a group beyond the slice reads nothing (offsets out of range return zeros
// without a fetch), so every load is unconditional and the compiler's count of loads in flight stays exact
template <int T, int TP>
__device__ __forceinline__ void load_grp16(Grp16<TP>& r, const Unit16<TP>& u, int g) {
    const bool in = g < u.g1;
#pragma unroll
    for (int j = 0; j < T; ++j)
        r.b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u.rsrc, in ? u.voff[j] : u.oob,
                                                                                u.soff[j] + g * GROUP16_BYTES, 0));
    __builtin_amdgcn_sched_barrier(0);      // the loads stay here in the instruction stream (csrc/mlp32_dev.h, load_chunk)
}

// acc[t][.] += x[16 rows][16 k of the group] * W[tile t][the same k]; the two accumulators of a tile alternate,
// so consecutive MFMAs of a wave with one tile do not wait for each other (40-cycle result latency, 32-cycle issue)
template <int T, int TP>
__device__ __forceinline__ void mma_grp16(const f32x4& a, const Grp16<TP>& r, f32x4 (&acc)[TP][2]) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int t = 0; t < T; ++t)
            acc[t][T == 1 ? (j4 & 1) : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j4], r.b[t][j4], acc[t][T == 1 ? (j4 & 1) : 0], 0, 0, 0);
}

// One pass of a wave over its k-groups with T column tiles: RING16 group buffers in rotation (the one being
// multiplied + RING16 - 1 in flight), the A fragment of the next group read from LDS while this one is multiplied.
// `t`: the first two groups of the pass (fetch_pass16).
#ifndef D4PG_RING16
#define D4PG_RING16 3
#endif
constexpr int RING16 = D4PG_RING16;
static_assert(RING16 >= 3 && RING16 <= 6, "two groups arrive prefetched; the ring holds them and at least one more");
#ifndef D4PG_RING16_LEAN
#define D4PG_RING16_LEAN 3
#endif
// ring depth by column tiles per pass (the lean build's groups are half as long in time)
constexpr int ring16_of(int tp) { return tp == TPW16_LEAN ? D4PG_RING16_LEAN : RING16; }

#ifdef D4PG_LDS_ASM      // A fragments by an explicit ds_read_b128 one group ahead of their use (the compiler sinks a plain
                         // LDS load to just in front of the first MFMA that needs it)
__device__ __forceinline__ f32x4 lds_frag16(const float* p) {
    f32x4 v;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_frag16_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#else
__device__ __forceinline__ f32x4 lds_frag16(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void lds_frag16_wait() {}
#endif

template <int T, int TP>
__device__ __forceinline__ void pass16(const Unit16<TP>& u, const Pre16<TP>& t, const float* Al, f32x4 (&acc)[TP][2]) {
    constexpr int RING = ring16_of(TP);
    Grp16<TP> r[RING];
    r[0] = t.g[0];
    r[1] = t.g[1];
#pragma unroll
    for (int j = 2; j < RING - 1; ++j) load_grp16<T>(r[j], u, u.g0 + j);
    const int last = u.g1 - 1;
    int g = u.g0;
    f32x4 a = lds_frag16(Al + 16 * g), an;
    lds_frag16_wait();
    for (; g + RING <= u.g1; g += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            load_grp16<T>(r[(s + RING - 1) % RING], u, g + s + RING - 1);
            an = lds_frag16(Al + 16 * min(g + s + 1, last));
            mma_grp16<T>(a, r[s], acc);
            lds_frag16_wait();
            a = an;
        }
    }
    const int rem = u.g1 - g;      // 0 .. RING - 1 groups left, all loaded
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) {
        if (s < rem) {
            an = lds_frag16(Al + 16 * min(g + s + 1, last));
            mma_grp16<T>(a, r[s], acc);
            lds_frag16_wait();
            a = an;
        }
    }
}

// The first two k-groups of pass `pass` of the wave's work in layer L, unconditionally (a wave without work
// reads zeros from out-of-range offsets).  Pass 0 is issued by the layer BEFORE, so the stream of weight
// loads of a workgroup does not stop at a layer boundary.
// (`run` false, wave-uniform: a layer that is going to be skipped; nothing is fetched)
template <int NW, int TP>
__device__ __forceinline__ Pre16<TP> fetch_pass16(const LayerW32& L, int wave, int lane, int pass, bool run) {
    Unit16<TP> u = unit16_of<NW, TP>(L, wave, lane, pass);
    if (!run) {
#pragma unroll
        for (int j = 0; j < TP; ++j) u.voff[j] = u.oob;
    }
    Pre16<TP> t;
    load_grp16<TP>(t.g[0], u, u.g0);
    load_grp16<TP>(t.g[1], u, u.g0 + 1);
    return t;
}

// epi(row0, col, v): v = rows row0 .. row0 + 3 of output column col (bias added), col < L.N.
// A   : LDS fp32 [M][lda] (M = 8 or 16 rows: with 8, the upper half of every MFMA is computed on copies of the
//       rows and dropped), columns up to 16 * L.ks readable (zero beyond the true K)
// t   : the first two k-groups of this wave's first pass (fetch_pass16, issued by the layer before)
// red : LDS scratch for K-slice partial sums, (NW - 1) * 256 floats
// pre : issues the loads of the NEXT layer's first groups; ONE call site, behind this layer's last own load
template <int NW, int M, int TP, typename F, typename P>
__device__ __forceinline__ void layer16(Pre16<TP>& t, const float* A, int lda, const LayerW32& L, float* red, F&& epi, P&& pre) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, q = lane >> 4;
    const float* Al = A + (c & (M - 1)) * lda + 4 * q;
    const bool split = L.ksplit > 1;
    const int passes = split ? 1 : (L.ntiles + TP * NW - 1) / (TP * NW);
    for (int pass = 0; pass < passes; ++pass) {      // (more than one: layers wider than 16 * TP * NW = 512 / 256)
        const Unit16<TP> u = unit16_of<NW, TP>(L, wave, lane, pass);
        f32x4 acc[TP][2];
        float bias[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int col = (u.tile0 + j * NW) * 16 + c;
            bias[j] = L.bias[(j < u.tiles && col < L.N) ? col : 0];
        }
        switch (u.tiles) {      // wave-uniform
            case 1: pass16<1>(u, t, Al, acc); break;
            case 2: pass16<2>(u, t, Al, acc); break;
            case 3: if constexpr (TP >= 3) pass16<3>(u, t, Al, acc); break;
            case 4: if constexpr (TP >= 4) pass16<4>(u, t, Al, acc); break;
            default: break;
        }
        if (pass + 1 < passes) t = fetch_pass16<NW, TP>(L, wave, lane, pass + 1, true);
        else pre();
        bool write = u.tiles > 0 && 4 * q < M;
        if (split) {      // workgroup-uniform; the slices of a tile meet through `red`, slice 0 finishes the tile
            const int slice = wave / L.ntiles;
            if (u.tiles > 0 && slice != 0)
                *reinterpret_cast<f32x4*>(red + ((size_t)((slice - 1) * L.ntiles + u.tile0) * 64 + lane) * 4) = acc[0][0] + acc[0][1];
            __syncthreads();
            write = write && slice == 0;
            if (write)
                for (int s = 1; s < L.ksplit; ++s)
                    acc[0][0] += *reinterpret_cast<const f32x4*>(red + ((size_t)((s - 1) * L.ntiles + u.tile0) * 64 + lane) * 4);
        }
        if (write) {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int col = (u.tile0 + j * NW) * 16 + c;
                if (j < u.tiles && col < L.N) {
                    f32x4 v = acc[j][0] + acc[j][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bias[j];
                    epi(4 * q, col, v);
                }
            }
        }
    }
}

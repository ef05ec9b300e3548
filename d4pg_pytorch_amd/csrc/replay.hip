// HBM-resident replay buffer and prioritised-sampling tree (gfx950).
//
// Storage: array of rows, row = [state S | action A | reward | done | gamma^k | next_state S]
// padded to a multiple of 4 floats, so a sampled transition is ONE contiguous
// row read and a batch gather is a row-gather (no host, no pickling; the
// reference builds each batch with a Python loop over 256 tuples,
// models/d4pg/replay_buffer.py:41-53).
//
// Prioritised sampling (models/d4pg/replay_buffer.py:89-209,
// models/d4pg/segment_tree.py) keeps the same maths — proportional, stratified,
// p_i = priority^alpha, w_i = (P(i) n)^-beta / max_w — but the binary
// sum-tree (21 dependent levels for 1M items) is replaced by a fan-out-64 tree:
// one 64-lane wave scans a whole node per step, so a descent or an update is
// 3-4 dependent memory round trips.  Internal sums are fp64 and every parent is
// RECOMPUTED from its 64 children on update (never incrementally adjusted), so
// there is no drift and duplicate indices in one batch are idempotent.
// A parallel min-tree yields p_min for max_w.  Since max_w = (p_min/total*n)^-beta
// the weight reduces to w_i = (p_min / p_i)^beta.
#include "common.h"

#include "replay_dev.h"

// ---------------------------------------------------------------------------
// Insert a staged chunk of rows at the ring cursor.
struct InsertArgs {
    ReplayDesc rp;
    const float* staged;     // [n, row_ld]
    long long head;          // ring cursor before this chunk (host-tracked)
    long long new_size;      // valid rows after this chunk
    long long new_head;
    int n;
    int write_tree_inputs;   // 1: emit slot indices + initial priorities for the tree update
    long long* slot_out;     // [n]
    float* prio_out;         // [n]  (= running max priority, models/d4pg/replay_buffer.py:121-122)
    DevState* st;
};

__global__ __launch_bounds__(256) void replay_insert_kernel(const InsertArgs a) {
    const int ld4 = a.rp.row_ld >> 2;
    const long long total = (long long)a.n * ld4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long row = i / ld4;
        const int c = (int)(i - row * ld4);
        long long slot = a.head + row;
        if (slot >= a.rp.capacity) slot -= a.rp.capacity;
        reinterpret_cast<f32x4*>(a.rp.rows + slot * a.rp.row_ld)[c] =
            reinterpret_cast<const f32x4*>(a.staged + row * a.rp.row_ld)[c];
        if (c == 0 && a.write_tree_inputs) {
            a.slot_out[row] = slot;
            a.prio_out[row] = a.st->max_priority;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.st->replay_size = a.new_size;
        a.st->replay_head = a.new_head;
        a.st->inserted_total += a.n;
    }
}

D4PG_API int d4pg_replay_insert(const InsertArgs* a, hipStream_t stream) {
    if (a->n < 1 || a->n > a->rp.capacity || (a->rp.row_ld & 3)) return -1;
    if (a->head < 0 || a->head >= a->rp.capacity) return -2;
    const long long total = (long long)a->n * (a->rp.row_ld >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(replay_insert_kernel, dim3(blocks), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Tree update as a stand-alone launch (one workgroup of 16 waves); the body lives in
// replay_dev.h because the fused bf16 pipeline runs it as an extra workgroup of its actor
// phase kernel.
__global__ __launch_bounds__(1024) void tree_update_kernel(const TreeUpdateArgs, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char tree_lds[];
    // (read in place from the kernel-argument segment: the level arrays are indexed at run time, which through a
    // by-value parameter means a copy of the arguments in scratch memory; the struct is the first parameter)
    const TreeUpdateArgs& a = *(const TreeUpdateArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    tree_update_body<16>(a, blockIdx.x, tree_lds, lds_bytes);
}

// parts > 1 (several workgroups, csrc/replay_dev.h) needs a tree of at least three levels and the ticket counter
D4PG_API int d4pg_tree_update_check(const TreeUpdateArgs* a) {
    if (a->n < 1 || a->t.L < 0 || a->t.L > MAXL || a->idx == nullptr || a->prio == nullptr) return -1;
    if (a->parts > 64 || (a->parts > 1 && (a->t.L < 2 || a->ticket == nullptr))) return -2;
    return 0;
}
D4PG_API int d4pg_tree_update(const TreeUpdateArgs* a, hipStream_t stream) {
    if (int rc = d4pg_tree_update_check(a)) return rc;
    // LDS for the list of owned items where it fits the default limit, else the body filters at every level
    int lds = tree_update_lds_bytes(a->n);
    if (lds > 64 * 1024) lds = 16;
    hipLaunchKernelGGL(tree_update_kernel, dim3(a->parts > 1 ? a->parts : 1), dim3(1024), lds, stream, *a, lds);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Sample a batch and gather it into the learner's input buffers: one wave per sample
// (SampleArgs, draw_slot and sample_row live in replay_dev.h: the fused bf16 pipeline also
// runs them in extra workgroups of its last launch).  Also performs the per-update "tick".
__global__ __launch_bounds__(256) void replay_sample_kernel(const SampleArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    const long long step = a.st->step_done;
    const long long size = a.st->replay_size;
    if (b < a.B && size > 0) sample_row(a, b, step, size, lane);
    if (a.do_tick && blockIdx.x == 0 && threadIdx.x == 0)
        devstate_tick(a.st, a.beta1, a.beta2, a.per_b0, a.per_b1, a.total_steps);
}

D4PG_API int d4pg_replay_sample(const SampleArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->ld_x < a->rp.S + a->rp.A) return -1;
    if (a->prioritized && (a->t.L < 0 || a->t.L > MAXL || a->t.n[a->t.L] > 64)) return -2;
    hipLaunchKernelGGL(replay_sample_kernel, dim3((a->B + 3) / 4), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Packed batches (replicated data parallelism, parallel/replicated.py): every learner rank
// draws its share of the next K global batches (a "chunk") from its own replay shard into ONE
// contiguous buffer  chunk[K][B][row_ld + PACK_EXTRA]
//     row: [ replay row (row_ld floats) | IS weight | slot bits 31..0 | slot bits 63..32 | 0 ]
// which is all-gathered across the ranks as is ([world][K][B][..]); batch_unpack then spreads
// the rows of the batch that belongs to the update in flight over the learner's input
// buffers (and performs the per-update tick).
constexpr int PACK_EXTRA = 4;     // floats appended to a replay row

struct SampleRowsArgs {
    ReplayDesc rp;
    TreeDesc t;
    DevState* st;
    float* out;          // [K * B, row_ld + PACK_EXTRA]
    int B;               // rows per batch (the strata of a prioritised draw)
    int K;               // batches in the chunk
    int prioritized;
    int pad_;
    long long step;      // update that consumes batch 0 of the chunk; batch j belongs to step + j (RNG
                         // stream position and beta schedule).  The chunk is drawn ahead of time, so
                         // the device step counter is not used.
    float per_b0, per_b1;
    long long total_steps;
};

__global__ __launch_bounds__(256) void replay_sample_rows_kernel(const SampleRowsArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    const long long size = a.st->replay_size;
    if (b >= a.B * a.K || size <= 0) return;
    const int j = b / a.B, i = b - j * a.B;
    float w;
    const long long slot = draw_slot(a.t, a.prioritized, i, a.B, a.step + j, size, a.st->seed, a.per_b0, a.per_b1,
                                     a.total_steps, lane, w);
    const int ld = a.rp.row_ld;
    const float* row = a.rp.rows + slot * ld;
    float* o = a.out + (size_t)b * (ld + PACK_EXTRA);
    for (int c = lane; c < ld; c += 64) o[c] = row[c];
    if (lane == 0) {
        o[ld] = w;
        o[ld + 1] = __int_as_float((int)(slot & 0xFFFFFFFFll));
        o[ld + 2] = __int_as_float((int)(slot >> 32));
        o[ld + 3] = 0.f;
    }
}

D4PG_API int d4pg_replay_sample_rows(const SampleRowsArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->K < 1 || a->out == nullptr) return -1;
    if (a->prioritized && (a->t.L < 0 || a->t.L > MAXL || a->t.n[a->t.L] > 64)) return -2;
    hipLaunchKernelGGL(replay_sample_rows_kernel, dim3((a->B * a->K + 3) / 4), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

// The chunk a batch is taken from, and the batch inside it, follow from the device step
// counter: s = step_done - step0, batch j = s % K of buffer (s / K) & 1.  The launch is
// therefore the same for every update and is captured with the rest of the update graph.
struct UnpackArgs {
    const float* in0;    // gathered chunk, even chunks  [world][K][Bl][row_ld + PACK_EXTRA]
    const float* in1;    // gathered chunk, odd chunks
    DevState* st;
    float* sa;
    float* s_pi;
    float* s2_a2;
    float* reward;
    float* done;
    float* gamma;
    float* weights;
    long long* idx;
    int B;               // global batch = world * Bl
    int Bl;              // rows per rank and batch
    int K;               // batches per chunk
    int S, A, row_ld, ld_x;
    int do_tick;
    long long step0;     // value of the device step counter at the first batch of chunk 0
    float beta1, beta2, per_b0, per_b1;
    long long total_steps;
};

__global__ __launch_bounds__(256) void batch_unpack_kernel(const UnpackArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b < a.B) {
        const long long s = a.st->step_done - a.step0;
        const int j = (int)(s % a.K);
        const float* base = ((s / a.K) & 1) ? a.in1 : a.in0;
        const int r = b / a.Bl, i = b - r * a.Bl;
        const float* row = base + ((size_t)(r * a.K + j) * a.Bl + i) * (a.row_ld + PACK_EXTRA);
        const int S = a.S, A = a.A;
        const size_t ob = (size_t)b * a.ld_x;
        for (int c = lane; c < S; c += 64) {
            const float sv = row[c];
            a.sa[ob + c] = sv;
            a.s_pi[ob + c] = sv;
            a.s2_a2[ob + c] = row[S + A + 3 + c];
        }
        for (int c = lane; c < A; c += 64) a.sa[ob + S + c] = row[S + c];
        if (lane == 0) {
            a.reward[b] = row[S + A];
            a.done[b] = row[S + A + 1];
            a.gamma[b] = row[S + A + 2];
            a.weights[b] = row[a.row_ld];
            const unsigned int lo = (unsigned int)__float_as_int(row[a.row_ld + 1]);
            const unsigned int hi = (unsigned int)__float_as_int(row[a.row_ld + 2]);
            a.idx[b] = (long long)(((unsigned long long)hi << 32) | lo);
        }
    }
    // the tick reads step_done (as every block above does) and writes step_cur / bc1 / bc2 / per_beta,
    // which nothing in this kernel reads
    if (a.do_tick && blockIdx.x == 0 && threadIdx.x == 0)
        devstate_tick(a.st, a.beta1, a.beta2, a.per_b0, a.per_b1, a.total_steps);
}

D4PG_API int d4pg_batch_unpack(const UnpackArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->Bl < 1 || a->K < 1 || a->B % a->Bl) return -1;
    if (a->ld_x < a->S + a->A || a->row_ld < 2 * a->S + a->A + 3 || a->in0 == nullptr || a->in1 == nullptr) return -2;
    hipLaunchKernelGGL(batch_unpack_kernel, dim3((a->B + 3) / 4), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

D4PG_API int d4pg_pack_extra() { return PACK_EXTRA; }

// ---------------------------------------------------------------------------
// Load an externally provided batch (host sampler / tests) into the learner's
// input buffers with the same column layout the sampling kernel produces.
struct ScatterBatchArgs {
    const float* state;       // [B, S]
    const float* action;      // [B, A]
    const float* next_state;  // [B, S]
    float* sa;
    float* s_pi;
    float* s2_a2;
    int B, S, A, ld_x;
};

__global__ __launch_bounds__(256) void scatter_batch_kernel(const ScatterBatchArgs a) {
    const int W = a.S + a.A;
    const long long total = (long long)a.B * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / W), c = (int)(i - (long long)b * W);
        const size_t o = (size_t)b * a.ld_x + c;
        if (c < a.S) {
            const float s = a.state[(size_t)b * a.S + c];
            a.sa[o] = s;
            a.s_pi[o] = s;
            a.s2_a2[o] = a.next_state[(size_t)b * a.S + c];
        } else {
            a.sa[o] = a.action[(size_t)b * a.A + (c - a.S)];
        }
    }
}

D4PG_API int d4pg_scatter_batch(const ScatterBatchArgs* a, hipStream_t stream) {
    if (a->B < 1 || a->ld_x < a->S + a->A) return -1;
    const long long total = (long long)a->B * (a->S + a->A);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(scatter_batch_kernel, dim3(blocks), dim3(256), 0, stream, *a);
    return (int)hipGetLastError();
}

D4PG_SIZEOF(TreeDesc)
D4PG_SIZEOF(ReplayDesc)
D4PG_SIZEOF(TreeUpdateArgs)
D4PG_SIZEOF(SampleArgs)
D4PG_SIZEOF(InsertArgs)
D4PG_SIZEOF(SampleRowsArgs)
D4PG_SIZEOF(UnpackArgs)
D4PG_SIZEOF(ScatterBatchArgs)

// Host-side native runtime of the actor-learner framework (plain C++17, C ABI).
//
//  * ShmRing     lock-free single-producer / single-consumer ring of fixed-size float
//                records in POSIX shared memory.  One per explorer; carries transitions
//                in the replay row format to the replay owner without pickling.  The
//                consumer can "peek" contiguous spans so the learner copies straight from
//                the (host-registered) ring into HBM with an async DMA.
//                Replaces the reference's multiprocessing.Queue(64) of pickled numpy
//                arrays with silent drops (models/agent.py:101-105, models/d4pg/engine.py:114).
//  * WeightSlab  seq-locked shared buffer holding the flat actor parameters: one writer,
//                any number of readers, every reader sees every published version
//                (the reference's weight queue hands each snapshot to exactly one explorer,
//                models/d4pg/d4pg.py:141-146, models/agent.py:37-49).
//  * ControlBlock shared array of 64-bit atomics: stop flag, update step, episode and
//                env-step counters, heartbeats (the reference's unsynchronised mp.Value ints,
//                models/d4pg/engine.py:115-117).
//  * NStep       vectorised n-step return assembler (models/agent.py:89-123 semantics).
//  * OUNoise     Ornstein-Uhlenbeck exploration noise (utils/utils.py:9-34), xoshiro256++ RNG.
//  * actor_forward  3-layer MLP inference on the padded flat parameter layout for CPU explorers.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <sys/mman.h>
#include <sched.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#define API extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint64_t kRingMagic = 0x44345047524E4731ull;   // "D4PGRNG1"
constexpr uint64_t kSlabMagic = 0x44345047534C4231ull;   // "D4PGSLB1"
constexpr uint64_t kCtrlMagic = 0x4434504743544C31ull;   // "D4PGCTL1"
constexpr size_t kHeaderBytes = 4096;                     // keeps the payload page aligned

struct Mapping {
    void* base = nullptr;
    size_t bytes = 0;
};

Mapping map_shm(const char* name, size_t bytes, bool create) {
    Mapping m;
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return m;
    if (create) {
        if (ftruncate(fd, (off_t)bytes) != 0) {
            close(fd);
            shm_unlink(name);
            return m;
        }
    } else {
        struct stat st;
        if (fstat(fd, &st) != 0) {
            close(fd);
            return m;
        }
        bytes = (size_t)st.st_size;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return m;
    m.base = p;
    m.bytes = bytes;
    return m;
}

inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// ------------------------------------------------------------------ ring
struct alignas(64) RingHeader {
    uint64_t magic;
    uint64_t capacity;     // rows
    uint64_t row_floats;
    uint64_t check;        // 1: every slot carries the index of the row it holds, verified on consumption
    alignas(64) std::atomic<uint64_t> write_idx;   // rows ever pushed   (producer-owned)
    alignas(64) std::atomic<uint64_t> read_idx;    // rows ever consumed (consumer-owned)
    alignas(64) std::atomic<uint64_t> dropped;     // rows refused because the ring was full
    std::atomic<uint64_t> closed;
    std::atomic<uint64_t> seq_errors;              // check mode: consumed slots whose index was not the expected one
};
static_assert(sizeof(RingHeader) <= kHeaderBytes, "header too large");

struct Ring {
    Mapping map;
    RingHeader* hdr;
    float* data;
    uint64_t* seq;         // [capacity] 1 + index of the row in the slot (behind the data area)
};

// Check mode (race detection): the producer stamps every slot with the running index of the row it
// writes, the consumer verifies on consumption that slot (index % capacity) holds exactly row `index`:
// a producer overrunning the consumer, a second producer or consumer, or a lost release/acquire pair
// all show up as a mismatch.
inline void ring_stamp(Ring* r, uint64_t first, uint64_t n) {
    if (!r->hdr->check) return;
    const uint64_t cap = r->hdr->capacity;
    for (uint64_t i = 0; i < n; ++i) r->seq[(first + i) % cap] = first + i + 1;
}
inline void ring_verify(Ring* r, uint64_t first, uint64_t n) {
    if (!r->hdr->check) return;
    const uint64_t cap = r->hdr->capacity;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) bad += r->seq[(first + i) % cap] != first + i + 1;
    if (bad) r->hdr->seq_errors.fetch_add(bad);
}
inline uint64_t* ring_seq_ptr(void* base, uint64_t capacity, uint64_t row_floats) {
    return reinterpret_cast<uint64_t*>((char*)base + kHeaderBytes + (size_t)capacity * row_floats * sizeof(float));
}

// ------------------------------------------------------------------ slab
struct alignas(64) SlabHeader {
    uint64_t magic;
    uint64_t n_floats;
    alignas(64) std::atomic<uint64_t> version;     // even: stable, odd: write in progress
    std::atomic<int64_t> update_step;
};

struct Slab {
    Mapping map;
    SlabHeader* hdr;
    float* data;
};

// ------------------------------------------------------------------ ctrl
struct CtrlHeader {
    uint64_t magic;
    uint64_t n_slots;
};
struct Ctrl {
    Mapping map;
    CtrlHeader* hdr;
    std::atomic<int64_t>* slots;
};

// ------------------------------------------------------------------ rng
struct Xoshiro {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void seed(uint64_t v) {
        for (auto& w : s) w = splitmix(v);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double normal() {   // Box-Muller, one value per call (spare discarded: simplicity over speed)
        double u1 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        const double u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct OU {
    int E, A;
    float theta, mu, max_sigma, min_sigma;
    int64_t decay_period;
    std::vector<float> x;
    Xoshiro rng;
};

struct NStep {
    int E, n, S, A, row_ld;
    float gamma;
    // per env: circular window of up to n pending (state, action, reward)
    std::vector<float> states, actions, rewards;
    std::vector<int> start, count;
};

inline int pad4(int v) { return (v + 3) / 4 * 4; }

}   // namespace

// =============================================================================== ring API
API void* ring_create(const char* name, uint64_t capacity, uint64_t row_floats) {
    if (capacity < 2 || row_floats < 1) return nullptr;
    const size_t bytes = kHeaderBytes + (size_t)capacity * row_floats * sizeof(float) + (size_t)capacity * sizeof(uint64_t);
    Mapping m = map_shm(name, bytes, true);
    if (!m.base) return nullptr;
    auto* hdr = new (m.base) RingHeader();
    hdr->capacity = capacity;
    hdr->row_floats = row_floats;
    hdr->check = 0;
    hdr->seq_errors.store(0);
    hdr->write_idx.store(0);
    hdr->read_idx.store(0);
    hdr->dropped.store(0);
    hdr->closed.store(0);
    std::atomic_thread_fence(std::memory_order_release);
    hdr->magic = kRingMagic;
    return new Ring{m, hdr, reinterpret_cast<float*>((char*)m.base + kHeaderBytes),
                    ring_seq_ptr(m.base, capacity, row_floats)};
}

API void* ring_attach(const char* name) {
    Mapping m = map_shm(name, 0, false);
    if (!m.base) return nullptr;
    auto* hdr = reinterpret_cast<RingHeader*>(m.base);
    if (hdr->magic != kRingMagic) {
        munmap(m.base, m.bytes);
        return nullptr;
    }
    return new Ring{m, hdr, reinterpret_cast<float*>((char*)m.base + kHeaderBytes),
                    ring_seq_ptr(m.base, hdr->capacity, hdr->row_floats)};
}

// Check mode on / off (set by the creator before any row is pushed) and its error count.
API void ring_set_check(void* h, int on) { static_cast<Ring*>(h)->hdr->check = on ? 1 : 0; }
API uint64_t ring_seq_errors(void* h) { return static_cast<Ring*>(h)->hdr->seq_errors.load(); }

API void ring_detach(void* h) {
    auto* r = static_cast<Ring*>(h);
    if (!r) return;
    munmap(r->map.base, r->map.bytes);
    delete r;
}

API int shm_remove(const char* name) { return shm_unlink(name); }

API uint64_t ring_capacity(void* h) { return static_cast<Ring*>(h)->hdr->capacity; }
API uint64_t ring_row_floats(void* h) { return static_cast<Ring*>(h)->hdr->row_floats; }
API void* ring_data_ptr(void* h) { return static_cast<Ring*>(h)->data; }
API uint64_t ring_data_bytes(void* h) {
    auto* r = static_cast<Ring*>(h);
    return r->hdr->capacity * r->hdr->row_floats * sizeof(float);
}
API uint64_t ring_size(void* h) {
    auto* r = static_cast<Ring*>(h);
    return r->hdr->write_idx.load(std::memory_order_acquire) - r->hdr->read_idx.load(std::memory_order_acquire);
}
API uint64_t ring_dropped(void* h) { return static_cast<Ring*>(h)->hdr->dropped.load(); }
API uint64_t ring_pushed_total(void* h) { return static_cast<Ring*>(h)->hdr->write_idx.load(); }
API void ring_close(void* h) { static_cast<Ring*>(h)->hdr->closed.store(1); }
API int ring_is_closed(void* h) { return (int)static_cast<Ring*>(h)->hdr->closed.load(); }

// Producer.  Pushes up to n rows.  If the ring is full: wait up to timeout_ms
// (lossless back-pressure) and then count the remainder as dropped.
// timeout_ms = 0 reproduces the reference's drop-when-full behaviour, but counted.
API int64_t ring_push(void* h, const float* rows, int64_t n, int64_t timeout_ms) {
    auto* r = static_cast<Ring*>(h);
    RingHeader* hd = r->hdr;
    const uint64_t cap = hd->capacity, rf = hd->row_floats;
    int64_t done = 0;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    while (done < n) {
        const uint64_t w = hd->write_idx.load(std::memory_order_relaxed);
        const uint64_t rd = hd->read_idx.load(std::memory_order_acquire);
        const uint64_t free_rows = cap - (w - rd);
        if (free_rows == 0) {
            if (hd->closed.load(std::memory_order_relaxed) || timeout_ms <= 0 ||
                std::chrono::steady_clock::now() >= deadline) {
                hd->dropped.fetch_add((uint64_t)(n - done));
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            continue;
        }
        uint64_t take = (uint64_t)(n - done) < free_rows ? (uint64_t)(n - done) : free_rows;
        const uint64_t pos = w % cap;
        if (take > cap - pos) take = cap - pos;   // contiguous span
        std::memcpy(r->data + pos * rf, rows + (size_t)done * rf, (size_t)take * rf * sizeof(float));
        ring_stamp(r, w, take);
        hd->write_idx.store(w + take, std::memory_order_release);
        done += (int64_t)take;
    }
    return done;
}

// Consumer: copy out up to max_rows rows.
API int64_t ring_pop(void* h, float* out, int64_t max_rows) {
    auto* r = static_cast<Ring*>(h);
    RingHeader* hd = r->hdr;
    const uint64_t cap = hd->capacity, rf = hd->row_floats;
    int64_t done = 0;
    while (done < max_rows) {
        const uint64_t rd = hd->read_idx.load(std::memory_order_relaxed);
        const uint64_t w = hd->write_idx.load(std::memory_order_acquire);
        uint64_t avail = w - rd;
        if (avail == 0) break;
        uint64_t take = (uint64_t)(max_rows - done) < avail ? (uint64_t)(max_rows - done) : avail;
        const uint64_t pos = rd % cap;
        if (take > cap - pos) take = cap - pos;
        std::memcpy(out + (size_t)done * rf, r->data + pos * rf, (size_t)take * rf * sizeof(float));
        ring_verify(r, rd, take);
        hd->read_idx.store(rd + take, std::memory_order_release);
        done += (int64_t)take;
    }
    return done;
}

// Consumer, zero copy: the first contiguous readable span (row offset + count).
// The rows stay owned by the ring until ring_advance() releases them.
API int64_t ring_peek(void* h, uint64_t* row_offset, int64_t max_rows) {
    auto* r = static_cast<Ring*>(h);
    RingHeader* hd = r->hdr;
    const uint64_t rd = hd->read_idx.load(std::memory_order_relaxed);
    const uint64_t w = hd->write_idx.load(std::memory_order_acquire);
    uint64_t avail = w - rd;
    const uint64_t pos = rd % hd->capacity;
    if (avail > hd->capacity - pos) avail = hd->capacity - pos;
    if ((int64_t)avail > max_rows) avail = (uint64_t)max_rows;
    *row_offset = pos;
    return (int64_t)avail;
}

// The same for a consumer that has spans in flight (handed to a DMA engine, not yet released): the first
// contiguous readable span that starts `ahead` rows past the read cursor.
API int64_t ring_peek_from(void* h, uint64_t ahead, uint64_t* row_offset, int64_t max_rows) {
    auto* r = static_cast<Ring*>(h);
    RingHeader* hd = r->hdr;
    const uint64_t rd = hd->read_idx.load(std::memory_order_relaxed) + ahead;
    const uint64_t w = hd->write_idx.load(std::memory_order_acquire);
    if (w <= rd) {
        *row_offset = 0;
        return 0;
    }
    uint64_t avail = w - rd;
    const uint64_t pos = rd % hd->capacity;
    if (avail > hd->capacity - pos) avail = hd->capacity - pos;
    if ((int64_t)avail > max_rows) avail = (uint64_t)max_rows;
    *row_offset = pos;
    return (int64_t)avail;
}

API void ring_advance(void* h, int64_t rows) {
    auto* r = static_cast<Ring*>(h);
    ring_verify(r, r->hdr->read_idx.load(std::memory_order_relaxed), (uint64_t)rows);
    r->hdr->read_idx.fetch_add((uint64_t)rows, std::memory_order_release);
}

// =============================================================================== slab API
API void* slab_create(const char* name, uint64_t n_floats) {
    const size_t bytes = kHeaderBytes + (size_t)n_floats * sizeof(float);
    Mapping m = map_shm(name, bytes, true);
    if (!m.base) return nullptr;
    auto* hdr = new (m.base) SlabHeader();
    hdr->n_floats = n_floats;
    hdr->version.store(0);
    hdr->update_step.store(-1);
    std::atomic_thread_fence(std::memory_order_release);
    hdr->magic = kSlabMagic;
    return new Slab{m, hdr, reinterpret_cast<float*>((char*)m.base + kHeaderBytes)};
}

API void* slab_attach(const char* name) {
    Mapping m = map_shm(name, 0, false);
    if (!m.base) return nullptr;
    auto* hdr = reinterpret_cast<SlabHeader*>(m.base);
    if (hdr->magic != kSlabMagic) {
        munmap(m.base, m.bytes);
        return nullptr;
    }
    return new Slab{m, hdr, reinterpret_cast<float*>((char*)m.base + kHeaderBytes)};
}

API void slab_detach(void* h) {
    auto* s = static_cast<Slab*>(h);
    if (!s) return;
    munmap(s->map.base, s->map.bytes);
    delete s;
}

API void* slab_data_ptr(void* h) { return static_cast<Slab*>(h)->data; }
API uint64_t slab_n_floats(void* h) { return static_cast<Slab*>(h)->hdr->n_floats; }
API uint64_t slab_version(void* h) { return static_cast<Slab*>(h)->hdr->version.load(std::memory_order_acquire); }
API int64_t slab_step(void* h) { return static_cast<Slab*>(h)->hdr->update_step.load(); }

// Writer protocol for an external copy engine (async D2H lands in slab_data_ptr()):
//   slab_begin_write -> copy -> slab_end_write.
API void slab_begin_write(void* h) {
    auto* s = static_cast<Slab*>(h);
    s->hdr->version.fetch_add(1, std::memory_order_acq_rel);   // odd
    std::atomic_thread_fence(std::memory_order_seq_cst);
}
API void slab_end_write(void* h, int64_t update_step) {
    auto* s = static_cast<Slab*>(h);
    s->hdr->update_step.store(update_step, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    s->hdr->version.fetch_add(1, std::memory_order_acq_rel);   // even
}
API void slab_publish(void* h, const float* src, int64_t update_step) {
    auto* s = static_cast<Slab*>(h);
    slab_begin_write(h);
    std::memcpy(s->data, src, (size_t)s->hdr->n_floats * sizeof(float));
    slab_end_write(h, update_step);
}

// Reader: copies a consistent snapshot if its version differs from *last_version.
// Returns 1 (new snapshot), 0 (unchanged or nothing published yet), -1 (gave up: writer stuck).
API int slab_read(void* h, float* dst, uint64_t* last_version) {
    auto* s = static_cast<Slab*>(h);
    // A writer that was descheduled in the middle of a publication holds the version odd for a scheduling
    // quantum or more on a loaded host, so "stuck" is decided by the wall clock (2 s), not by a spin count.
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
    for (uint64_t attempt = 0;; ++attempt) {
        if ((attempt & 1023) == 1023) {
            if (std::chrono::steady_clock::now() > deadline) break;
            sched_yield();
        }
        const uint64_t v1 = s->hdr->version.load(std::memory_order_acquire);
        if (v1 & 1) {
            cpu_relax();
            continue;
        }
        if (v1 == 0 || v1 == *last_version) return 0;
        std::memcpy(dst, s->data, (size_t)s->hdr->n_floats * sizeof(float));
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const uint64_t v2 = s->hdr->version.load(std::memory_order_acquire);
        if (v1 == v2) {
            *last_version = v1;
            return 1;
        }
    }
    return -1;
}

// =============================================================================== ctrl API
API void* ctrl_create(const char* name, uint64_t n_slots) {
    const size_t bytes = kHeaderBytes + (size_t)n_slots * sizeof(std::atomic<int64_t>);
    Mapping m = map_shm(name, bytes, true);
    if (!m.base) return nullptr;
    auto* hdr = reinterpret_cast<CtrlHeader*>(m.base);
    hdr->n_slots = n_slots;
    auto* slots = reinterpret_cast<std::atomic<int64_t>*>((char*)m.base + kHeaderBytes);
    for (uint64_t i = 0; i < n_slots; ++i) new (&slots[i]) std::atomic<int64_t>(0);
    std::atomic_thread_fence(std::memory_order_release);
    hdr->magic = kCtrlMagic;
    return new Ctrl{m, hdr, slots};
}
API void* ctrl_attach(const char* name) {
    Mapping m = map_shm(name, 0, false);
    if (!m.base) return nullptr;
    auto* hdr = reinterpret_cast<CtrlHeader*>(m.base);
    if (hdr->magic != kCtrlMagic) {
        munmap(m.base, m.bytes);
        return nullptr;
    }
    return new Ctrl{m, hdr, reinterpret_cast<std::atomic<int64_t>*>((char*)m.base + kHeaderBytes)};
}
API void ctrl_detach(void* h) {
    auto* c = static_cast<Ctrl*>(h);
    if (!c) return;
    munmap(c->map.base, c->map.bytes);
    delete c;
}
API uint64_t ctrl_slots(void* h) { return static_cast<Ctrl*>(h)->hdr->n_slots; }
API int64_t ctrl_load(void* h, uint64_t i) { return static_cast<Ctrl*>(h)->slots[i].load(std::memory_order_acquire); }
API void ctrl_store(void* h, uint64_t i, int64_t v) { static_cast<Ctrl*>(h)->slots[i].store(v, std::memory_order_release); }
API int64_t ctrl_add(void* h, uint64_t i, int64_t v) {
    return static_cast<Ctrl*>(h)->slots[i].fetch_add(v, std::memory_order_acq_rel) + v;
}
API int64_t ctrl_max(void* h, uint64_t i, int64_t v) {
    auto& s = static_cast<Ctrl*>(h)->slots[i];
    int64_t cur = s.load();
    while (cur < v && !s.compare_exchange_weak(cur, v)) {
    }
    return cur < v ? v : cur;
}

// =============================================================================== OU noise
API void* ou_create(int E, int A, float theta, float mu, float max_sigma, float min_sigma,
                    int64_t decay_period, uint64_t seed) {
    auto* o = new OU();
    o->E = E; o->A = A; o->theta = theta; o->mu = mu;
    o->max_sigma = max_sigma; o->min_sigma = min_sigma;
    o->decay_period = decay_period > 0 ? decay_period : 1;
    o->x.assign((size_t)E * A, mu);
    o->rng.seed(seed);
    return o;
}
API void ou_destroy(void* h) { delete static_cast<OU*>(h); }
API void ou_reset(void* h, int env) {   // env < 0: all
    auto* o = static_cast<OU*>(h);
    if (env < 0) {
        std::fill(o->x.begin(), o->x.end(), o->mu);
    } else {
        std::fill(o->x.begin() + (size_t)env * o->A, o->x.begin() + (size_t)(env + 1) * o->A, o->mu);
    }
}
API const float* ou_state(void* h) { return static_cast<OU*>(h)->x.data(); }
// x <- x + theta (mu - x) + sigma N(0, I);  out = clip(action + x, low, high)
// t[e] = step index inside the episode (drives the sigma schedule).
API void ou_sample(void* h, const float* action, const int64_t* t, float low, float high, float* out) {
    auto* o = static_cast<OU*>(h);
    for (int e = 0; e < o->E; ++e) {
        double frac = (double)t[e] / (double)o->decay_period;
        if (frac > 1.0) frac = 1.0;
        const float sigma = o->max_sigma - (o->max_sigma - o->min_sigma) * (float)frac;
        for (int a = 0; a < o->A; ++a) {
            float& x = o->x[(size_t)e * o->A + a];
            x = x + o->theta * (o->mu - x) + sigma * (float)o->rng.normal();
            float v = action[(size_t)e * o->A + a] + x;
            v = v < low ? low : (v > high ? high : v);
            out[(size_t)e * o->A + a] = v;
        }
    }
}

// =============================================================================== n-step
API void* nstep_create(int E, int n, int S, int A, float gamma) {
    if (E < 1 || n < 1 || S < 1 || A < 1) return nullptr;
    auto* s = new NStep();
    s->E = E; s->n = n; s->S = S; s->A = A; s->gamma = gamma;
    s->row_ld = pad4(2 * S + A + 3);
    s->states.assign((size_t)E * n * S, 0.f);
    s->actions.assign((size_t)E * n * A, 0.f);
    s->rewards.assign((size_t)E * n, 0.f);
    s->start.assign(E, 0);
    s->count.assign(E, 0);
    return s;
}
API void nstep_destroy(void* h) { delete static_cast<NStep*>(h); }
API int nstep_row_ld(void* h) { return static_cast<NStep*>(h)->row_ld; }
API void nstep_reset(void* h, int env) {
    auto* s = static_cast<NStep*>(h);
    if (env < 0) {
        std::fill(s->count.begin(), s->count.end(), 0);
        std::fill(s->start.begin(), s->start.end(), 0);
    } else {
        s->count[env] = 0;
        s->start[env] = 0;
    }
}

namespace {
// Emit the oldest pending entry of env e with the horizon currently in the window.
void nstep_emit(NStep* s, int e, const float* next_state, float done, float* row) {
    const int n = s->n, S = s->S, A = s->A;
    const int first = s->start[e];
    float R = 0.f, g = 1.f;
    for (int i = 0; i < s->count[e]; ++i) {
        R += g * s->rewards[(size_t)e * n + (first + i) % n];
        g *= s->gamma;
    }
    std::memset(row, 0, sizeof(float) * s->row_ld);
    std::memcpy(row, &s->states[((size_t)e * n + first) * S], sizeof(float) * S);
    std::memcpy(row + S, &s->actions[((size_t)e * n + first) * A], sizeof(float) * A);
    row[S + A] = R;
    row[S + A + 1] = done;
    row[S + A + 2] = g;                  // gamma ** (number of rewards summed)
    std::memcpy(row + S + A + 3, next_state, sizeof(float) * S);
    s->start[e] = (first + 1) % n;
    s->count[e] -= 1;
}
}   // namespace

// One environment step for all E envs.  active[e] == 0 skips env e (already finished).
// Emits up to E * n rows (row format of the replay buffer) into out; returns the row count.
API int64_t nstep_push(void* h, const float* states, const float* actions, const float* rewards,
                       const float* next_states, const uint8_t* dones, const uint8_t* truncs,
                       const uint8_t* active, float* out, int64_t max_rows) {
    auto* s = static_cast<NStep*>(h);
    const int n = s->n, S = s->S, A = s->A;
    int64_t rows = 0;
    for (int e = 0; e < s->E; ++e) {
        if (active != nullptr && !active[e]) continue;
        const int slot = (s->start[e] + s->count[e]) % n;
        std::memcpy(&s->states[((size_t)e * n + slot) * S], states + (size_t)e * S, sizeof(float) * S);
        std::memcpy(&s->actions[((size_t)e * n + slot) * A], actions + (size_t)e * A, sizeof(float) * A);
        s->rewards[(size_t)e * n + slot] = rewards[e];
        s->count[e] += 1;
        const float d = dones[e] ? 1.f : 0.f;
        const float* ns = next_states + (size_t)e * S;
        if (s->count[e] >= n) {
            if (rows >= max_rows) return -1;
            nstep_emit(s, e, ns, d, out + rows * s->row_ld);
            ++rows;
        }
        if (dones[e] || (truncs != nullptr && truncs[e])) {
            while (s->count[e] > 0) {   // flush the tail with shrinking horizons
                if (rows >= max_rows) return -1;
                nstep_emit(s, e, ns, d, out + rows * s->row_ld);
                ++rows;
            }
            s->start[e] = 0;
        }
    }
    return rows;
}

// =============================================================================== CPU actor
// flat parameter layout (floats), leading dimensions padded to 4:
//   W1 [H, pad4(S)] | b1 [pad4(H)] | W2 [H, pad4(H)] | b2 [pad4(H)] | W3 [A, pad4(H)] | b3 [pad4(A)]
API void actor_forward(const float* flat, int S, int Hd, int A, const float* x, int E, float* out,
                       float action_scale) {
    const int ld1 = pad4(S), ldh = pad4(Hd);
    const float* W1 = flat;
    const float* b1 = W1 + (size_t)Hd * ld1;
    const float* W2 = b1 + pad4(Hd);
    const float* b2 = W2 + (size_t)Hd * ldh;
    const float* W3 = b2 + pad4(Hd);
    const float* b3 = W3 + (size_t)A * ldh;
    std::vector<float> h1(Hd), h2(Hd);
    for (int e = 0; e < E; ++e) {
        const float* xe = x + (size_t)e * S;
        for (int o = 0; o < Hd; ++o) {
            const float* w = W1 + (size_t)o * ld1;
            float acc = b1[o];
#pragma omp simd reduction(+ : acc)
            for (int k = 0; k < S; ++k) acc += w[k] * xe[k];
            h1[o] = acc > 0.f ? acc : 0.f;
        }
        for (int o = 0; o < Hd; ++o) {
            const float* w = W2 + (size_t)o * ldh;
            float acc = 0.f;
#pragma omp simd reduction(+ : acc)
            for (int k = 0; k < Hd; ++k) acc += w[k] * h1[k];
            acc += b2[o];
            h2[o] = acc > 0.f ? acc : 0.f;
        }
        for (int o = 0; o < A; ++o) {
            const float* w = W3 + (size_t)o * ldh;
            float acc = 0.f;
#pragma omp simd reduction(+ : acc)
            for (int k = 0; k < Hd; ++k) acc += w[k] * h2[k];
            out[(size_t)e * A + o] = std::tanh(acc + b3[o]) * action_scale;
        }
    }
}

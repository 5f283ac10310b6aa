// fw_engine.h -- the host engine's own records and the functions its translation units share (internal: nothing here is ABI).
//
// Division of labour:
//   host   - owns the control block of every spawner: emission clocks, enabled flags,
//            the OnDemand queue (ParticleSpawnerData minus `particles`, reference
//            src/core.rs:261-303) and evaluates compute_emission_count for Global
//            entries (src/core.rs:396-428, 553-575) in bit-exact fp32;
//   device - owns all particle state (fw_device.h) and runs spawn + update +
//            compaction; Nested entries are counted per parent on the device.
// A frame is: [params H2D on the copy stream] -> spawn kernel(s) -> update kernel, all
// asynchronous; live counts come back through a pinned snapshot ring that the update
// kernel writes directly (no memcpy in the frame).
//
// There is no CPU simulation path in this library.
//
// The engine is five translation units (round 5; one 4 300-line file until then), along the seams of the work:
//   fw_engine_mem.cpp    device / pinned memory of a context, exact counts, the error words of the update kernels
//   fw_engine_paths.cpp  which update path a particle type is on (FIFO ring / range ring / small / compacting): the transitions
//                        between them with live particles, the capacity policy, the tile table of the compacting launch
//   fw_engine_build.cpp  sync_spawner_data (core.rs:343-365): descriptors -> device tables of one spawner, and back
//   fw_engine_step.cpp   fw_step: lifetime windows, emission clocks, cohort replay, launch assembly
//   fw_engine_api.cpp    every other entry point of include/firework_hip.h (+ the debug hooks)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include <sys/mman.h>
#include <immintrin.h>

#include "../../include/firework_hip.h"
#include "../../include/firework_hip_debug.h"
#include "fw_kernels.h"
#include "fw_math.h"

namespace fwh {

// hipMemset runs on the null stream and may return before the fill has happened; the streams of a context are
// non-blocking ones, which the null stream does not order itself against: a kernel enqueued right after would race the fill
// (seen with two processes on the GPU: a fresh context's first frames read the forecast entries a previous context had left
// in the recycled allocation).  Every fill of this file is part of an allocation (rare): wait for it.
inline hipError_t fw_memset_done(void *p, int v, size_t bytes) {
    hipError_t e = hipMemset(p, v, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

constexpr int kParamRing = 8;    // per-frame parameter buffers in flight
#ifndef FW_BAR_PARAM_KB
#define FW_BAR_PARAM_KB 64
#endif
constexpr size_t kBarParamBytes = (size_t)FW_BAR_PARAM_KB << 10;  // fw_ctx::b_param
constexpr int kSnapRing = 8;     // live-count snapshots in flight
constexpr int kSnapEvery = 4;    // frames between snapshots
constexpr int kTabRing = 4;      // staging buffers for tile-table uploads
constexpr uint64_t kResidentSlots = 1024;  // fw_k_update workgroups resident at once on MI355X (256 CUs x 4)
constexpr uint32_t kMinCapacity = 4096;
constexpr uint32_t kNoSeg = 0xFFFFFFFFu;  // SpawnerHost::seg entry not built yet
constexpr uint64_t kMaxSpawnPerOp = 1ull << 30;
constexpr uint32_t kTimingEvents = 4096;
constexpr uint32_t kMaxFifoSegs = FW_FIFO_PER_LAUNCH;  // FIFO segments per context: one launch (beyond: the general path)
constexpr size_t kMaxCohorts = 16384;    // spawn cohorts a FIFO segment tracks before it gives the mode up (tiny dt)
constexpr uint32_t kReportRing = 32768;  // pinned per-frame cohort sizes of a ring that receives Nested children (> kMaxCohorts)

extern thread_local std::string g_create_error;  // (fw_last_error(nullptr): per calling thread, like errno; fw_engine_api.cpp)

// Host memory the per-frame loops of fw_step stream through -- the segment and spawner records, every segment's lifetime
// window.  As ordinary heap blocks they sit on thousands of 4 KB pages (one window ring per emitter alone): past ~3000
// emitters the loops missed the TLB on most records (2048 emitters: 28 ns each, 4096: 39 ns).  They come from 2 MB-aligned
// chunks instead, which the kernel is asked to back with huge pages (madvise: a no-op where transparent huge pages are off).
// One process-wide pool: blocks of 1536 << k bytes (a window ring of 64 << k entries) on free lists, larger requests mapped
// on their own; chunks are never returned (destroyed contexts leave their blocks on the lists).  Thread-safe: contexts may be
// driven from different threads.
class HugePool {
public:
    static void *alloc(size_t bytes) {
        if (bytes > kMaxBlock) return map(bytes);
        const int c = cls(bytes);
        std::lock_guard<std::mutex> lk(mu());
        Node *&head = lists()[c];
        if (head) {
            Node *n = head;
            head = n->next;
            return n;
        }
        const size_t sz = kBase << c;
        char *&cur = chunk_cur(), *&end = chunk_end();
        if (cur == nullptr || (size_t)(end - cur) < sz) {
            // (what is left of the old chunk goes to the lists of the classes it still fits)
            while (cur && (size_t)(end - cur) >= kBase) {
                int k = 0;
                while (k + 1 < kClasses && (kBase << (k + 1)) <= (size_t)(end - cur)) k++;
                Node *n = reinterpret_cast<Node *>(cur);
                n->next = lists()[k], lists()[k] = n;
                cur += kBase << k;
            }
            cur = static_cast<char *>(map(kChunk));
            if (!cur) return nullptr;
            end = cur + kChunk;
        }
        void *p = cur;
        cur += sz;
        return p;
    }
    static void free(void *p, size_t bytes) {
        if (!p) return;
        if (bytes > kMaxBlock) {
            munmap(p, (bytes + kHuge - 1) / kHuge * kHuge);
            return;
        }
        std::lock_guard<std::mutex> lk(mu());
        Node *n = static_cast<Node *>(p);
        n->next = lists()[cls(bytes)], lists()[cls(bytes)] = n;
    }

private:
    struct Node {
        Node *next;
    };
    static constexpr size_t kBase = 1536, kHuge = 2u << 20, kChunk = 8u << 20;
    static constexpr int kClasses = 11;  // 1.5 KB ... 1.5 MB
    static constexpr size_t kMaxBlock = kBase << (kClasses - 1);
    static int cls(size_t bytes) {
        int c = 0;
        while ((kBase << c) < bytes) c++;
        return c;
    }
    static std::mutex &mu() {
        static std::mutex m;
        return m;
    }
    static Node **lists() {
        static Node *l[kClasses] = {};
        return l;
    }
    static char *&chunk_cur() {
        static char *p = nullptr;
        return p;
    }
    static char *&chunk_end() {
        static char *p = nullptr;
        return p;
    }
    // `bytes` rounded up to 2 MB, aligned to 2 MB, huge pages requested
    static void *map(size_t bytes) {
        const size_t len = (bytes + kHuge - 1) / kHuge * kHuge;
        char *raw = static_cast<char *>(mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (raw == MAP_FAILED) return nullptr;
        char *al = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) / kHuge * kHuge);
        if (al > raw) munmap(raw, (size_t)(al - raw));
        if (al + len < raw + len + kHuge) munmap(al + len, (size_t)(raw + len + kHuge - (al + len)));
        madvise(al, len, MADV_HUGEPAGE);
        return al;
    }
};

template <typename T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <typename U>
    HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        void *p = HugePool::alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n) { HugePool::free(p, n * sizeof(T)); }
    template <typename U>
    bool operator==(const HugeAlloc<U> &) const { return true; }
    template <typename U>
    bool operator!=(const HugeAlloc<U> &) const { return false; }
};

struct CurveCopy {
    int32_t kind = 0, n = 0;
    std::vector<float> times, values;  // values: n (curve) or 4n (gradient)
};

struct TypeHost {
    fw_particle_settings ps{};
    CurveCopy scale, base, emis;
    // every particle of this type outlives a step of dt < life_lo_safe: lifetime = u * (max - min) + min, u in [0, 1),
    // two ulps of margin for the rounding of the lerp; NaN when the range is not finite (never provably safe)
    float life_lo_safe = 0.f;
};

struct EmissionHost {
    // ---- what fw_step's spawner loop touches: this state and the head of `es` (pacing, counts, mode: its first 40 bytes),
    // next to each other (with thousands of emitters the loop is bound by the cache lines it streams, not its arithmetic)
    // EmissionData (reference src/core.rs:261-267)
    float last_emission = 0.f, time_passed_in_cycle = 0.f;
    uint64_t serial = 0;      // RNG stream position (Global entries; Nested ones live on the device)
    uint32_t emit_idx = 0;    // -> FwEmit
    uint32_t emit_slot = 0;   // -> device serial counter (Nested)
    uint32_t dst_seg = 0;     // segment of es.particle_index (cached: the frame loop then touches only this record)
    float life_lo_safe = 0.f; // TypeHost::life_lo_safe of es.particle_index
    float between = 0.f;      // (es.offset_end - es.offset_start) / es.count: fw_emission_count_unit
    bool enabled = false, emits_on_other_particles = false;
    bool assigned = false;    // emit_idx / emit_slot are owned by this entry
    fw_emission_settings es{};
};

// The emission entries of a spawner.  ONE entry -- by far the most common spawner -- lives inside the spawner's own record
// (no heap block, no pointer to chase: the spawner array is then all the frame loop streams); more entries live in a vector.
struct EmVec {
    uint32_t n_ = 0;
    EmissionHost one_;
    std::vector<EmissionHost> more_;
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    EmissionHost *data() { return n_ > 1 ? more_.data() : &one_; }
    const EmissionHost *data() const { return n_ > 1 ? more_.data() : &one_; }
    EmissionHost &operator[](size_t i) { return data()[i]; }
    const EmissionHost &operator[](size_t i) const { return data()[i]; }
    EmissionHost *begin() { return data(); }
    EmissionHost *end() { return data() + n_; }
    const EmissionHost *begin() const { return data(); }
    const EmissionHost *end() const { return data() + n_; }
    void assign(size_t n, const EmissionHost &v) {
        more_.clear();
        one_ = v;
        if (n > 1) more_.assign(n, v);
        n_ = (uint32_t)n;
    }
    void clear() { assign(0, EmissionHost{}); }
};

// A queue in ONE pooled allocation (HugePool), a power of two of entries: in the steady state every frame pops one entry and
// pushes one per segment; a std::deque pays its chunk bookkeeping and two dependent pointer hops for each -- 30 us per frame
// with 2048 emitters.
template <typename T>
struct Ring {
    std::vector<T, HugeAlloc<T>> v;
    uint32_t head = 0, n = 0;
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    T &front() { return v[head]; }
    T &back() { return v[(head + n - 1) & (uint32_t)(v.size() - 1)]; }
    void pop_front() { head = (head + 1) & (uint32_t)(v.size() - 1), n--; }
    void pop_back() { n--; }
    void clear() { head = n = 0; }
    void push_back(const T &x) {
        if (n == v.size()) {  // grow to the next power of two, oldest entry first
            std::vector<T, HugeAlloc<T>> w(v.empty() ? 64 : v.size() * 2);
            for (uint32_t i = 0; i < n; i++) w[i] = v[(head + i) & (uint32_t)(v.size() - 1)];
            v.swap(w), head = 0;
        }
        v[(head + n) & (uint32_t)(v.size() - 1)] = x, n++;
    }
};

struct alignas(64) SegHost {
    // Lifetime window: a particle is destroyed by the update in which age >= lifetime (core.rs:590-592), and
    // lifetime <= life_bound, so everything alive was spawned less than life_bound of simulated time ago.  The sum of
    // the Global spawn counts inside that window bounds the live count without any device feedback.
    struct Spawned {
        double t;        // simulated time at the spawn
        uint64_t n;
        uint64_t frame;  // frame of the spawn (the device's ages are fp32 sums: the error grows with the steps taken)
    };
    using Window = Ring<Spawned>;
    // ---- what the per-frame loops of fw_step touch, in ONE cache line (with thousands of segments those loops are
    // bound by how many lines they stream, not by their arithmetic)
    bool in_use = false;
    bool nested_fed = false;    // receives Nested children: count not host-predictable
    bool collides = false;      // the type has collision settings (core.rs:137-138): frames run the collision path
    bool coll_inplace = false;  // ... without destroy_on_collision: a bounce changes neither age, lifetime nor order
                                // (core.rs:607-643), so the type may live in a ring (the COLL instantiations of the ring kernels)
                                // (also set for a type whose curve keys exceed the LDS staging area: the same feature
                                // kernels read them from device memory -- SegHost::bigkeys)
    bool auto_capacity = false; // capacity was derived (fw_particle_settings.capacity == 0): the library may grow it
    bool win_ok = false;
    bool dead_at_end = false;   // the last step left this type's destroyed records at the END of its buffer (a range ring's
                                // update fills them from there, the youngest dead first: fw_k_update_range)
    bool solo = false;          // a small type ONE Global entry feeds: its frame-begin work (lifetime window, bound) happens where
                                // fw_step's spawner loop makes its op -- the record is streamed once per frame, not twice (fw_ctx::n_solo)
    uint32_t capacity = 0;
    uint32_t ub = 0;            // upper bound of the device count (after this frame's spawns)
    uint32_t frame_spawn = 0;   // Global particles appended this frame
    uint32_t dev_count = 0;     // nested_fed: live count of the latest snapshot row (growth trigger)
    uint32_t dev_epoch = 0;     //   ... the frame that row describes, and
    float dev_rate = 0.f;       //   ... how fast the count was growing between the last two rows (particles per frame, >= 0)
    uint32_t snap_count = 0;    // nested_fed: ... the same count, kept together with
    uint64_t snap_cum = 0;      //   cum_spawn of the frame that row describes: Global particles since then are host-known
    uint64_t cum_spawn = 0;     // Global particles ever appended (host-known)
    uint64_t win_sum = 0;
    // ---- second line: the window itself and the segment's mode (both per-frame loops look at them)
    double life_bound = 0.0;
    char *inst = nullptr;       // caller-owned device buffer of ParticleInstance records (fw_spawner_attach_instances)
    Window win;
    bool fifo = false;          // FIFO ring (below)
    bool range = false;         // range ring (below)
    bool fifo_mat = false, fifo_dev = false, range_mat = false, range_dev = false, virt_parent = false;  // (below)
    // ---- the rest
    int spawner = -1, type = -1;
    uint32_t type_idx = 0, n_lplanes = 0;
    uint32_t keys_off = 0, keys_len = 0;  // key pool window of the segment's type
    uint32_t keys_cap = 0;                // ... and the floats reserved for it (returned to fw_ctx::free_keys with the type)
    bool bigkeys = false;                 // more keys than the streaming kernels stage in LDS (FW_KEYS_MAX floats)
    std::vector<int32_t> lplane_emission;  // [n_lplanes] the emission entry each last_emitted_age plane belongs to
    char *buf[2] = {nullptr, nullptr};
    char *destroyed = nullptr;
    uint32_t inst_cap = 0;
    // colours of the type at age 0: what both colour planes are filled with when the buffers are allocated, so that a
    // constant gradient's plane never has to be written by the update (FwOutWin::wr5 / wr6)
    float fill_bc[4] = {0, 0, 0, 0}, fill_em[4] = {0, 0, 0, 0};
    bool colors_dirty = false;  // the caller wrote particles (any colours) into the current buffer
    // FIFO ring (fw_kernels.h: FwFifoSeg): a type with a single lifetime value, Global emission only, no collisions.
    // ONE buffer (buf[0] == buf[1]); logical particle i sits in slot (head + i) mod capacity; `ub` is the EXACT live
    // count.  The host replays the fp32 age of every spawn cohort (same additions as the device), which tells it how
    // many particles each update destroys -- always the oldest ones.
    uint32_t head = 0;
    float fifo_life = 0.f;  // the lifetime every particle of the type gets (core.rs:455 with min == max)
    int32_t fifo_wm = 0;    // FwFifoArgs::write_mask of the type
    // FW_TYPE_NOSPIN (fw_device.h): no particle of the type can turn; the rotation plane is neither read nor written
    bool nospin = false;
    float const_rot[4] = {0.f, 0.f, 0.f, 1.f};
    // ... and keeps its lifetimes in one more 4-byte plane behind the n_lplanes last_emitted_age planes instead of in Q3
    // (allocated with the type, kept when the type leaves the mode)
    uint32_t n_xplanes = 0;
    struct Cohort {
        uint32_t n;
        float age;
        uint64_t frame = 0;  // frame the cohort was added in
        bool known = true;   // false: a cohort of Nested children whose size the device has not been asked for yet
    };
    std::deque<Cohort> coh;  // oldest first
    // A ring in a spawner WITH Nested entries: in frames that run the Nested pass its new particles are materialised in
    // the ring before the update (fw_k_spawn / fw_k_nest address it through the head) and fw_k_update_fifo gives them
    // their first update (FwFifoSeg::mat).  fifo_dev: the type receives Nested children -- its live count is known to the
    // device only, and the size of each frame's cohort reaches the host through a pinned ring (h_report[frame %
    // kReportRing] = {epoch, added}) long before the host needs it: when the cohort's age reaches the lifetime.
    unsigned long long *h_report = nullptr;
    // Range ring (fw_kernels.h: FwRangeRec): a type whose lifetime is a RANGE, Global emission only, no collisions, in a
    // spawner without Nested entries.  ONE buffer (buf[0] == buf[1]) used as a ring: [old survivors | young]; the young
    // part -- slot of its first particle, its size, its spawn cohorts -- is host-known exactly (the host made every spawn
    // count and replays the fp32 age of every cohort: fw_ctx::birth_age); the size of the old part is the device's
    // count minus young_n.  `ub` bounds the total as for any segment (lifetime window, snapshots).
    // (what the range pass of every frame reads and writes, next to each other)
    uint32_t young_lo = 0, young_n = 0;
    float range_life_lo = 0.f;  // every particle outlives an update that leaves its age below this (TypeHost::life_lo_safe)
    struct YCohort {
        uint64_t frame;  // frame of the spawn
        uint32_t n;
    };
    Ring<YCohort> ycoh;  // the young cohorts, oldest first
    bool few_ring = false;  // a range ring below fw_ctx::range_min: only because the context holds few segments (fw_ctx::range_few)
    bool spilled = false;   // a range ring that qualifies for a FIFO ring: the context holds more such types than one FIFO launch (fw_ctx::n_spilled)
    // A SMALL type (fw_k_small.hip, round 5): a few hundred particles, updated by ONE WAVE (four types per workgroup) instead of a
    // workgroup of the compacting kernels -- no tile table entry, no forecast.  Same buffers and layout as a compacting segment:
    // entering and leaving the mode is this flag (fw_ctx::n_small, small_eligible / leave_small).
    bool small = false;
    bool small_ok = false;    // ... qualifies for it (small_eligible); `small` follows the context's mode (fw_ctx::small_on)
    bool wide = false;        // ... as a WIDE type: up to a few thousand particles, a workgroup instead of a wave (fw_ctx::wide_max)
    bool wide_big = false;    // ... one that sustains more than fw_ctx::wide_mid particles: on the kernel only among wide_min types
    bool one_feeder = false;  // exactly one emission entry (a Global one) spawns into the type (SegHost::solo)
    float expect_live = 0.f;  // live particles the emitters that feed the type sustain (what derive_capacity derives the capacity from)
    uint32_t r_old = 0, r_new = 0, r_young = 0;  // workgroups of each role the device table provides for the segment
    uint32_t r_low[3] = {0, 0, 0};               // frames in a row a role's need has been far below what is provided
    uint32_t r_need[3] = {0, 0, 0};              // what each role needed in the latest frame (the table keeps more: fit())
    uint32_t r_status_base = 0;                  // first look-back word of its OLD workgroups in the current table
    uint32_t ticket_base = 0;                    // value of FwGlobals::range_ticket[segment] at the start of the next launch (START tickets)
    // ... in a spawner WITH Nested entries (core.rs:471-546):
    //   range_mat  other particles' entries emit FROM this type: in frames that run a Nested pass its Global particles are
    //              materialised behind the young part by fw_k_spawn before the pass (core.rs:488) and fw_k_update_range
    //              gives them their first update (FW_RREC_MAT);
    //   range_dev  the type RECEIVES Nested children: its live count -- hence the size of its young part -- is known to the
    //              device only (FW_RREC_DEV).  The host still knows where the young part STARTS: cohorts join the old part a
    //              lifetime.min after they were added, and by then the update of their frame has long left their size in the
    //              pinned ring h_report (as for a FIFO ring that receives children).
    // A ring type other particles' entries emit from whose Global particles need NOT be in memory for the frame's Nested pass:
    // every Nested entry on it is a CountOverDuration with count > 0 and 0 <= offset_start <= offset_end, and the type's
    // lifetimes are positive -- then compute_emission_count(age 0, last f32::MIN, ..) emits nothing for a particle born this
    // frame (core.rs:553-575: since = min(0, end) - start <= 0) and only leaves `next` in its last_emitted_age, which the lane
    // that spawns the particle inside the ring's update kernel computes itself (fw_init_last_emitted).  Such a type is spawned
    // in its update kernel in EVERY frame: a steady Nested frame is fw_k_nest + the update, without fw_k_spawn.
    struct DCohort {
        uint64_t frame;
        uint32_t n;
        bool known;
    };
    std::deque<DCohort> dcoh;   // range_dev: the young cohorts, oldest first (sizes unknown until they are needed)
    std::deque<YCohort> gcoh;   // range_dev: cohorts that have joined the old part and may still hold survivors (their bound)
    uint64_t gcoh_sum = 0;
    uint32_t rold_seen = 0;     // the old part's size as of the last exact read (refresh_counts_exact): FwGlobals::rold
    uint32_t r_young_main = 0;  // range_dev: young tiles the current table keeps in front (the rest: probably idle, at its end)
    bool ring() const { return fifo || range; }  // one buffer, particle 0 not in slot 0
    // FW_TYPE_DERIVED (fw_device.h): the planes S4 / Q5 / Q6 are not stored by the update; every reader evaluates scale and
    // colours from age / lifetime / initial_scale (an attached instance buffer receives them in its records).  Every type but
    // colliding ones and those whose curve keys exceed the LDS staging (fw_ctx::derive_all, wants_derived)
    bool derived = false;
    // ... but not yet: the caller wrote particles (any scale, any colours), and those that die in the very next step carry
    // what was written in their destroyed records -- the planes are read for one more frame, then the mode starts
    bool derive_pending = false;
    bool derive_ready = false;  // ... that frame has been enqueued: the flag flips at the start of the next fw_step
    // the attached buffer is a WINDOWED one (fw_spawner_attach_instances_window): the caller draws d_out[first, first + count)
    // and asks for `first` -- which lets a range ring keep its path (its tiles know a record's index counted from the
    // particles the update destroys, not from 0)
    bool inst_window = false;
    // where the lifetime of particle i is when the type cannot turn: a plane index (compacting / range segments), or
    // 0xFFFFFFFF = the one value fifo_life (a FIFO ring)
    uint32_t life_plane() const { return (nospin && !fifo) ? n_lplanes : 0xFFFFFFFFu; }
};

struct alignas(64) SpawnerHost {
    // ---- first line: what every frame reads of a spawner; the entries follow (EmVec: one entry is inline)
    bool alive = false;
    bool initialized = false, finished_notified = false;
    // An internal error of an update kernel (FwGlobals::err_host) named one of this spawner's particle types: its particle
    // state can no longer be trusted -- an in-place ring update that went wrong has overwritten its own input and cannot be
    // redone.  Sticky: fw_step refuses to run and every call that reads or writes the spawner's particles returns FW_EHIP
    // until fw_spawner_update_settings rebuilds it (which drops all particles, core.rs:343-365) or it is destroyed.
    bool poisoned = false;
    // ... and the rebuilt spawner keeps its particle types off the in-place ring paths
    bool no_rings = false;
    uint64_t manual_queued_count = 0;
    float origin_pos[3] = {0, 0, 0}, origin_rot[4] = {0, 0, 0, 1}, parent_vel[3] = {0, 0, 0};
    float mod_scale = 1.f, mod_speed = 1.f;
    EmVec em;
    // ---- the rest
    uint32_t uid = 0;
    int32_t starts_enabled = 1;
    std::vector<TypeHost> types;
    std::vector<uint32_t> seg;  // per type
};

template <typename T>
struct DevArray {
    T *d = nullptr;
    size_t cap = 0;
};

}  // namespace fwh
using namespace fwh;

// The Global ops of one emission index: a growable array that owns its storage -- or, in a frame that will most likely hand the
// kernels an op TABLE (small types: fw_k_update_small reads its ops from pinned host memory), writes straight into the pinned
// parameter slot the kernels will read (fw_step: `pre_slot`): 96 bytes per emitter that are written once instead of written,
// read and written again.
struct OpList {
    FwOp *p = nullptr;
    size_t n = 0, cap = 0;
    bool lent = false;  // p points into a parameter slot, not into `own`
    std::vector<FwOp> own;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    FwOp *begin() { return p; }
    FwOp *end() { return p + n; }
    const FwOp *begin() const { return p; }
    const FwOp *end() const { return p + n; }
    FwOp *data() { return p; }
    FwOp &operator[](size_t i) { return p[i]; }
    const FwOp &operator[](size_t i) const { return p[i]; }
    void clear() {
        n = 0;
        if (lent) lent = false, p = own.data(), cap = own.size();
    }
    void borrow(FwOp *mem, size_t c) { p = mem, cap = c, n = 0, lent = true; }  // (of an empty list)
    void reserve_own(size_t c) {  // contents move into (larger) storage of the list's own
        c = std::max<size_t>(std::max<size_t>(c, n), 64);
        if (!lent) {
            if (c <= cap) return;
            own.resize(c);  // (keeps the contents)
        } else {
            if (own.size() < c) own.resize(c);
            if (n) memcpy(own.data(), p, n * sizeof(FwOp));
            lent = false;
        }
        p = own.data(), cap = own.size();
    }
    FwOp &push_slot() {  // the next op, to be filled in place (NOT zeroed)
        if (n == cap) reserve_own(cap * 2);
        return p[n++];
    }
    void push_back(const FwOp &x) { push_slot() = x; }
    void append(const FwOp *b, const FwOp *e) {
        for (; b != e; ++b) push_back(*b);
    }
};

struct FwLevel {  // ops of one emission index (spawn order inside a frame: core.rs:377-428)
    OpList g;
    std::vector<FwNestOp> n;
};

struct fw_ctx {
    std::vector<FwLevel> levels;  // per-frame scratch of fw_step: one entry per emission index in use
    OpList ops_scratch;
    std::vector<uint32_t> grow_scratch;  // fw_step: Nested-fed segments past half their capacity
    // device staging of the record-format copies (read_particles / write_particles / pack_instances): ONE allocation that
    // only ever grows, instead of a hipMalloc + hipFree pair per call (each a device-wide synchronisation and an address-
    // space change; profiles/r02/shared_gpu.txt)
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    bool seg_kind_changed = false;       // a ring left its mode inside the current fw_step (realloc_segment)
    bool derive_ready_any = false;       // some SegHost::derive_ready is set
    // undo log of fw_step's host half: spawn_particles is all-or-nothing per frame in the reference, so a frame that
    // cannot be enqueued (limit exceeded, allocation failure) must leave clocks, queues and RNG serials untouched
    struct EmUndo {
        uint32_t spawner, entry;
        float last_emission, time_passed_in_cycle;
        bool enabled;
        uint64_t serial;
    };
    struct SpUndo {
        uint32_t spawner;
        uint64_t manual_queued_count;
    };
    std::vector<EmUndo> undo_em;
    std::vector<SpUndo> undo_sp;
    int device = 0;
    uint32_t seed = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    // Rings next to compacting segments: the ring launch and the general launch of a frame touch disjoint segments, so the
    // ring launch goes to a stream of its own and the two run concurrently (in one stream the second launch waits for the
    // first to drain: 1M ring particles + one small compacting emitter cost 39.8 us per frame, 29.4 with everything on the
    // general path).  Each chain is in order on its own stream; they are joined where the other stream (or the caller's
    // work on it) looks at ring data -- no event in a steady-state frame.  Not used while a ring has an attached instance
    // buffer, belongs to a spawner with Nested entries (fw_k_spawn / fw_k_nest on the main stream feed it) or a live-count
    // ring is registered.  FW_FIFO_STREAM=0: everything on the one stream.
    hipStream_t fifo_stream = nullptr;
    hipEvent_t ev_side = nullptr, ev_main = nullptr;
    bool use_fifo_stream = true;
    bool side_dirty = false;      // ring launches on the side stream that the main stream has not waited for
    bool fifo_last_side = false;  // where the previous frame's ring launch went
    bool main_reads_ring = false; // work enqueued on the main stream since then reads ring data (must finish first)
    bool own_stream = false;
    std::string err;
    int update_mode = FW_MODE_FUSED;
    uint32_t spin_limit = 1u << 16;
    uint32_t dbg = 0;  // FW_DEBUG: profiling-only kernel ablations (results are wrong when set)
    uint64_t recovered_rings = 0;  // rings sent to the compacting path because a cohort report was missing (fw_step: recoverable)

    std::vector<SpawnerHost, HugeAlloc<SpawnerHost>> spawners;  // (HugePool: huge pages)
    std::vector<SegHost, HugeAlloc<SegHost>> segs;
    uint32_t n_types = 0, n_emits = 0, n_emit_slots = 0;
    // table slots of destroyed / rebuilt spawners, reused by the next build (a type owns the key window
    // [type_idx * FW_KEYS_MAX, +FW_KEYS_MAX) of the key pool, so windows are recycled with their type)
    std::vector<uint32_t> free_types, free_emits, free_emit_slots;
    // the key pool: every type owns a window sized for its own curves (any number of samples, curve.rs:40-75)
    size_t keys_end = 0;                                      // floats handed out so far
    std::vector<std::pair<uint32_t, uint32_t>> free_keys;     // {offset, length} of windows of released types

    FwGlobals g{};
    DevArray<FwSeg> d_segs;
    DevArray<FwType> d_types;
    DevArray<FwTypeColl> d_type_coll;
    DevArray<float> d_keys;
    DevArray<FwEmit> d_emits;
    DevArray<unsigned long long> d_emit_serial;
    DevArray<uint32_t> d_nest_start;             // FwGlobals::nest_start: START tickets of the Nested entries (one per emit slot)
    std::vector<uint32_t> nest_ticket_base;      // ... and the value each has at the start of the next launch that uses it
    uint32_t max_seg = 0;
    size_t tiles_cap = 0, nest_tiles_cap = 0, nest_ops_cap = 0;

    // per-frame parameter ring (pinned host + device copies)
    size_t param_bytes = 0;
    char *h_param[kParamRing] = {};
    char *d_param[kParamRing] = {};
    hipEvent_t ev_copied[kParamRing] = {}, ev_consumed[kParamRing] = {};
    bool consumed_pending[kParamRing] = {};
    // Global-only frames with more ops than fit the kernel arguments: the kernel reads the op table straight from the
    // pinned ring slot (no copy, no events); a slot is free again once the launch after its frame has started, which
    // that launch reports through a pinned word (FwUpdateArgs::done_tag).  FW_OPS_ZEROCOPY=0: staged copy + events.
    bool ops_zerocopy = true;
    unsigned long long *h_done = nullptr;      // pinned; written by workgroup 0 of every update launch
    unsigned long long *h_err = nullptr;       // pinned; FwGlobals::err_host (h_done + 4: the same allocation); [1]: the device's
                                               // error flags are set (fw_flag)
    std::string poison_msg;                    // what poll_device_error saw
    uint32_t n_poisoned = 0;                   // spawners ever marked (fw_step scans for live ones only while non-zero)
    uint64_t slot_frame[kParamRing] = {};      // frame that last used the slot through the zero-copy path (+1; 0 = free)

    // live-count snapshots written by the update kernel into pinned host memory
    // Live-count snapshots: the update kernel stores {epoch, count} of each segment into a pinned row with one 8-byte
    // store; the host recognises a finished row by its tag -- no event, no packet between launches.
    unsigned long long *h_snap = nullptr;  // [kSnapRing][max_seg]
    bool snap_pending[kSnapRing] = {};
    bool snap_seen[kSnapRing] = {};
    uint32_t snap_epoch[kSnapRing] = {};
    std::vector<uint64_t> snap_cum[kSnapRing];  // cum_spawn of every segment when the frame was enqueued

    // device-resident segment -> tile table
    uint32_t *d_tile_first = nullptr;
    size_t tile_first_cap = 0;
    std::vector<uint32_t> tiles_dev;
    uint32_t total_tiles_dev = 0;
    uint32_t *h_tab[kTabRing] = {};
    uint2 *d_tile_keys = nullptr;  // per segment: {keys_off, keys_len}
    uint2 *h_keys[kTabRing] = {};
    uint4 *d_tile_desc = nullptr;  // per tile: {segment, first tile, tile count, 0}
    uint4 *h_desc[kTabRing] = {};
    size_t tile_desc_cap = 0;
    hipEvent_t ev_tab[kTabRing] = {};
    bool tab_pending[kTabRing] = {};
    uint64_t tab_seq = 0, ring_seq = 0;
    uint32_t vt_rounds = 1;  // new-particle tile size of the current frame (rounds of 256)
    bool tab_force = false;  // a segment was (re)built: re-send the descriptors even if the tile counts are equal

    // survivor forecast sums (update kernels)
    uint32_t fc_sums_prev = 0;    // format of the forecast the last forecast frame produced
    uint4 *d_fce = nullptr;       // [2][tiles_cap] forecast entries of small segments (double-buffered)
    unsigned long long *d_fc = nullptr;   // three rotating buffers of forecast sums: S[tiles_cap] | S2[tiles_cap / 64 + 1] | tag
    size_t fc_len = 0;          // elements per buffer
    uint64_t fc_seq = 0;        // forecast-producing frames so far (buffer rotation)
    bool fc_dirty = false;      // the tile table changed: clear all three buffers before the next forecast frame
    bool fc_ok = false;        // the previous frame left a forecast that still describes the device state
    uint32_t fc_dt_bits = 0;   // ... computed for this dt
    uint64_t fc_tab_seq = 0;   // ... under this tile table
    bool use_forecast = true;  // FW_FORECAST=0 disables (A/B, debugging)
    // Threshold forecast (round 6, fw_kernels.h: FwUpdateArgs::fc_theta, fw_k_fc_resolve): a dt that does NOT repeat used to send every
    // compacting frame through the decoupled look-back (configs[2] on this path: 400 us per frame against 247 with a forecast).
    // Once the host has seen dt change, forecast frames of tf_min_tiles tiles or more (per-tile entries only) also leave, per tile,
    // the survivors a step of theta = 1.25 dt would destroy; the next frame, whatever its dt below that theta, runs fw_k_fc_resolve
    // (a wave per tile: ~3 us) in front of the streaming kernel instead of the look-back schedule.  A dt at or beyond theta, a
    // changed tile table, sums-format forecasts and small frames (a look-back costs them less than a launch) take the look-back as before.
    // FW_TF=0 switches it off, FW_TF_MIN_TILES=n (the tests: 0).
    bool use_tf = true;
    uint32_t tf_min_tiles = 768;  // (tools/r06_tf_min_tiles.py: 977 tiles 30.1 -> 22.2 us per frame, 256 tiles 17.0 -> 18.2)
    uint4 *d_fct = nullptr;     // [2][tiles_cap] headers, double-buffered like d_fce
    float2 *d_fcl = nullptr;    // [2][tiles_cap][FW_TF_K] (age, lifetime) of the risky survivors
    size_t tf_cap = 0;          // tiles_cap the two arrays were allocated for (0: not yet)
    uint32_t tf_armed = 0;      // frames left in which producers write lists (re-armed whenever dt differs from the last frame's)
    float tf_prev_theta = 0.0f; // theta the PREVIOUS forecast frame's lists were made for (0: it made none)
    uint64_t tf_frames = 0;     // frames that ran fw_k_fc_resolve + the streaming schedule instead of the look-back
    bool use_static_new = true;  // static output slots for new particles when all of them survive (FW_STATIC_NEW)
    uint32_t snap_every = kSnapEvery;  // frames between live-count snapshots (FW_SNAP_EVERY)
    bool use_stream = true;    // FW_STREAM=0: forecast frames keep the count-park-store kernel (A/B)
    // AABB fused into the update (fw_ctx_track_aabbs): per-tile boxes of the last update, valid while nothing touched
    // the state or the tile table since
    bool track_aabb = false;
    uint32_t boxes_epoch = 0;  // epoch of the update that left valid boxes (0 = none)
    bool colors_dirty = false; // some SegHost::colors_dirty is set
    bool use_fifo = true;      // FW_FIFO=0: constant-lifetime types take the general (compacting) path too (A/B, tests)
    bool fifo_nested = true;   // FW_FIFO_NESTED=0: ... those of spawners with Nested entries do (A/B)
    bool use_nospin = true;    // FW_NOSPIN=0: every type keeps its rotation plane (A/B)
    bool use_derived = true;   // FW_DERIVED=0: every type stores its scale / colour planes, attached instance buffer or not (A/B)
    // Round 6: scale, base colour and emissive colour are pure functions of (age, lifetime, initial_scale) (core.rs:601-605,
    // 652-655) -- the update of EVERY type whose curves fit the LDS staging stops storing them (36 of the 100 B a configs[2]
    // particle moved), not only of types with an attached instance buffer: the renderer extracts visible spawners only
    // (render.rs:382-403), the planes were written for nobody in every other frame.  FW_DERIVED=1: only with a buffer attached
    // (rounds 3-5; a share of the test functions keeps that form under the suite).
    bool derive_all = true;
    // Smallest (derived or given) capacity that makes a type a FIFO ring (FW_FIFO_MIN; the tests set 0).  Below a few
    // tens of thousands of particles a frame is launch latency whatever the path.  Next to compacting segments the ring
    // launch runs on its own stream (fifo_stream) and wins at any size (tools/fifo_threshold.py, tools/mixed_context.py);
    // where it cannot -- attached instance buffers, Nested spawners, a registered live-count ring -- the two launches of
    // a mixed context run one after the other and a small ring costs a few microseconds more than it saves.
    uint32_t fifo_small_tiles = 384;  // FIFO launches of a context with fewer four-round tiles than this use one-round tiles (FW_FIFO_SMALL)
    uint32_t fifo_min = 32768;
    uint32_t n_fifo = 0;       // FIFO segments in use (at most kMaxFifoSegs: their records travel in kernel arguments)
    // A context with MORE one-lifetime types than one FIFO launch holds (round 5).  The ninth used to land on a range ring next to
    // eight FIFO rings: two kinds of launch per frame, one after the other -- 9 emitters of 22 000 particles 20.7 us per frame
    // where nine range rings take 13.2 (profiles/r04/few_small_emitters.txt).  Now the type that does not fit takes a range ring
    // AND every FIFO ring of the context becomes one where it stands (fifo_to_range: no copy, particles and order kept; build time,
    // the context is synchronised): one kind of launch again.  While such rings exist, further one-lifetime types join them.
    uint32_t n_spilled = 0;    // SegHost::spilled segments
    // ---- small types (SegHost::small): the wave-per-type kernel.  A type is one when it is built (or when it leaves a small ring:
    // drop_few_rings) if the emitters that feed it sustain at most small_max / 2 particles and nothing else claims it (no ring, no
    // Nested entry on or from it, no collisions, no instance buffer, no per-tile AABBs); it leaves for good when its live bound
    // passes small_max (it simply becomes a compacting segment: same buffers).  FW_SMALL=0 / FW_SMALL_MAX=n
    bool use_small = true;
    uint32_t small_max = 768;
    uint32_t n_small = 0;
    // The kernel pays from a few hundred small types on: a wave walks its type's list round by round, ~15.6 us per launch whatever the
    // number of types up to ~600, where a workgroup per type (the compacting kernels) takes 11.9 us for 96 types, 14.3 for 256, 18.9
    // for 512 (profiles/r05/mid_emitters_paths.txt).  The context runs its eligible types (SegHost::small_ok, n_small_ok of them) on
    // the kernel from small_min of them on and takes them off it again below three quarters of that (update_small_mode: when a
    // spawner is built or destroyed -- the context is synchronised then; the flag flips, nothing is copied).  FW_SMALL_MIN
    // WIDE types: emitters that sustain up to wide_max particles (hundreds of emitters of a thousand particles: two workgroups per
    // type on the compacting kernels, each a chain of tile table -> forecast -> count -> look-back) are walked by ONE workgroup of the
    // same kernel, the same launch; a narrow type whose bound passes small_max becomes a wide one, a wide one leaves the mode past
    // 2 x wide_max.  FW_WIDE_MAX; 0: no wide types
    // ... in contexts of wide_min eligible types or more (with the same hysteresis as small_min): 1024 x 600 / 1024 x 1000 particles 26 /
    // 31 us per launch against 29 / 53 on the compacting kernels, but 512 x 1500 24 against 20.5 (profiles/r05/wide_sweep.txt,
    // wide_ablations.txt): one workgroup walks a type's rounds one after the other, the compacting kernels spread them over several
    // workgroups -- which pays until those no longer fit the chip at once.  FW_WIDE_MIN
    // Types of up to wide_mid particles (four rounds of a workgroup) gain from a third of that on already: 256 x 600 14.9 against 16.8 us
    // per frame, 384 x 1000 19.3 against 23.1, 512 x 1000 20.3 against 25.6 (profiles/r05/mid_paths_sweep.txt); larger ones -- 1500
    // particles, six rounds -- lose up to 512 types (24.3 against 21.2) and win where the compacting launch no longer fits the chip.
    // Round 6 (tools/threshold_sweep.py, tools/r06_wave_vs_workgroup.py): from about as many types as the chip has SIMDs (1024) a WAVE per
    // type beats a WORKGROUP per type at EVERY size the kernel takes -- 1024 x 1000 particles 23.2 against 34.8 us per frame, 2048 x
    // 1000 39.1 against 58.0, 2048 x 2000 58.6 against 85.9, 1024 x 300 20.3 against 28.5 -- while at 512 types the workgroup wins from
    // 1000 particles on (16.4 against 19.1) and 768 types are a draw (profiles/r06/wave_vs_workgroup.txt): every SIMD holds a wave
    // of its own either way, and the workgroup form pays a barrier and an LDS exchange per round for parallelism nobody lacks.  From
    // wave_all_min eligible types on (off again below five sixths of it) the wide types of the context are laid out in the wave
    // role: the same kernel, the same launch, only the list's partition (FwSmallArgs::n_narrow) moves.  FW_WAVE_ALL_MIN; 0: never
    uint32_t wave_all_min = 896;
    bool wave_all_on = false;
    uint32_t wide_min = 768;
    bool wide_on = false;      // wide types of more than wide_mid particles run on the kernel
    bool wide_mid_on = false;  // ... those of up to wide_mid do
    uint32_t wide_mid = 1400;
    uint32_t wide_max = 2048;
    uint32_t n_narrow = 0;   // of small_list (narrow types first)
    uint32_t small_min = 352;
    uint32_t n_small_ok = 0;
    uint32_t n_inst = 0;     // segments with an instance buffer attached (SegHost::inst): which instantiation the small launch runs
    uint32_t n_small_coll = 0;  // types on the kernel that have collision settings: its COLL instantiation
    // In frames that run the collision passes (materialise -> count -> scan -> update: four dependent launches on the main stream, 49 us
    // for a destroy_on_collision type of 20 000 particles) the small launch -- which touches nothing those passes touch -- runs next to
    // them on the ring stream (fifo_stream): 512 / 2048 small emitters next to such a type 66.9 / 84.3 -> 53.1 / 67.3 us per frame
    // (profiles/r05/nested_among_small_side.txt).  Not in frames whose separate passes are a Nested entry's only: those are bound by the
    // host's launches, and the extra event costs more than the overlap gives (44.4 -> 48.0).  Under the rules of a ring launch there:
    // own stream only, no live-count ring, no colliding small type (a new collider set travels in the main stream), not in a frame that
    // re-sends the list (a type left the mode: the compacting launch of this very frame updates it).
    // Its op table is then recycled by an event on that stream, not by the main stream's "frame started" word.
    bool small_last_side = false;
    bool small_on = false;
    // ... and the host half of their frames (thousands of emitters: the frame is bound by the cache lines fw_step streams).
    // A SOLO segment (SegHost::solo: a small type with one Global feeder) is not visited by the per-segment pass at the start of a
    // frame: what that pass does for it -- expire the lifetime window, tighten the bound -- happens in the spawner loop, right
    // before its op is made, and its frame_spawn stays 0 (its one op of the frame carries the count).  The pass walks `big_list`
    // (every other segment in use) instead of the segment array.  A segment becomes solo in the first frame-begin pass after it
    // became small, never before (the pass still owes it one ordinary frame begin).  FW_HOST_FAST=0: no solo segments, no ops
    // written in place (OpList).
    bool host_fast = true;
    uint32_t n_solo = 0;
    std::vector<uint32_t> big_list;
    bool big_dirty = true;
    std::vector<uint32_t> small_list;   // the segments, ascending (rebuilt when small_dirty)
    bool small_dirty = true;
    uint32_t *d_small = nullptr, *h_small = nullptr;  // device list / pinned staging
    size_t small_cap = 0;
    hipEvent_t ev_small = nullptr;
    bool small_pending = false;
    std::vector<FwOp> fifo_ops;  // this frame's Global ops that feed FIFO segments (spawned inside fw_k_update_fifo)
    std::vector<std::pair<uint32_t, FwOp>> range_mat_ops;  // the same for range rings other particles' entries emit from
    std::vector<std::pair<uint32_t, FwOp>> fifo_mat_ops;  // {emission index, op}: rings of spawners with Nested entries -- the
                                                         // Nested pass of the frame, if there is one, must find them in memory
    uint64_t tev_frames = 0;   // frames timed so far (a frame may take several update launches)
    // ---- range rings (SegHost::range)
    bool use_range = true;       // FW_RANGE=0: lifetime-range types take the compacting path (A/B, tests)
    uint32_t range_min = 8192;   // smallest capacity that makes one (FW_RANGE_MIN; the tests use 0): a range ring costs a
                                 // small segment three workgroups where the compacting path needs one.  Swept on many equal
                                 // emitters (profiles/r04/range_min_sweep.txt): 8192 is where the one-round tiles start to win
    uint32_t n_range = 0;
    // A context with FEW segments -- the reference's own regime: examples/sparks.rs is one spawner of ~730 particles -- runs a small
    // type on a range ring too: one emitter of 733 particles 13.2 -> 9.1 us per frame, 8-64 such emitters 16 -> 10-11
    // (profiles/r04/few_small_emitters.txt), where with thousands of small emitters the compacting path wins (range_min above).
    // Up to range_few segments in use, no FIFO ring among them (a FIFO launch and a range launch run one after the other), such a
    // type becomes a range ring whatever its size (SegHost::few_ring); the spawner that takes the context past either condition
    // sends those rings to the compacting path (drop_few_rings: build time, the context is synchronised).  FW_RANGE_FEW; 0: off
    uint32_t range_few = 192;  // (64 until the range launch read its records from device memory: profiles/r05/mid_emitters_paths.txt)
    uint32_t n_few = 0;     // SegHost::few_ring segments
    bool few_blocked = false;  // the context has outgrown the rule: no new small rings until it is back at half of range_few (a
                               // context whose spawners come and go around the limit would convert rings at every crossing)
    uint32_t n_in_use = 0;  // SegHost::in_use segments
    // A range launch whose rings hold fewer than range_small_tiles four-round tiles in all (FW_RANGE_SMALL; not with a ring whose
    // count only the device knows), or with a colliding ring, runs on OLD / YOUNG tiles of ONE round (fw_k_update_range: TR): a
    // quarter of hysteresis, a change re-sends the table.
    uint32_t range_small_tiles = 384;
    bool range_small = false;
    // Rounds of the YOUNG workgroups of a four-round launch, chosen per launch (round 5): tiles of 512 slots (2) when the range
    // rings of the context hold range_young_big particles each or more on average (hysteresis of a quarter; a change re-sends the
    // table), 1024 (4) otherwise and always with an attached instance buffer.  FW_RANGE_YOUNG_BIG=n (0: never).
    // Round 6: never -- with 56 instead of 100 bytes per particle a 512-slot tile keeps too few bytes in flight (configs[2] 199 us per
    // frame on 512-slot tiles, 185 on 1024-slot ones; 2048-slot ones -- FW_RANGE_YR=8 -- lose 2 %); the rule of round 5 was 32768.
    uint32_t range_young_rounds = 4;
    uint32_t range_young_big = 0;
    std::vector<FwOp> range_ops;  // this frame's Global ops that feed range rings
    // age, BEFORE the current frame's update, of a particle born in frame f -- the same for every segment of the context:
    // born with age 0, then one fp32 addition per frame (core.rs:594), exactly the device's additions
    struct BirthAge {
        uint64_t frame;
        float age;
    };
    std::deque<BirthAge> birth_age;  // oldest first; only frames some range ring may still hold young particles of
    float range_life_max = 0.f;      // largest SegHost::range_life_lo in the context
    float range_age_keep = 0.f;      // largest lifetime.max of a range ring that receives Nested children (plus a margin)
    FwRangeDesc *d_rdesc = nullptr;  // device table: one descriptor per workgroup of the range launch
    FwRangeDesc *h_rdesc = nullptr;  // pinned staging of it
    size_t rdesc_cap = 0;
    uint32_t r_total = 0;            // workgroups of the range launch
    bool r_force = true;             // a range segment was (re)built: re-send the table
    hipEvent_t ev_rtab = nullptr;
    bool rtab_pending = false;
    unsigned long long *d_rstatus = nullptr;  // look-back words of the OLD workgroups
    char *h_rparam[kParamRing] = {};          // per-frame records + ops, written by the host, read by the kernel in place
    // ... in DEVICE memory the host writes through the large BAR (fine-grained, hipExtMallocWithFlags) when the platform maps it, in
    // pinned host memory otherwise: a workgroup's record is the first thing its loads depend on, and read over the bus it costs
    // every launch ~3 us of dead time (tools/barwrite.hip, profiles/r05/rparam_ab.txt: one sparks.rs emitter 8.9 -> 5.7 us per frame,
    // one GPU's share of configs[4] 90.3 -> 85.9, configs[2] 318 -> 311).  The host only ever WRITES such a buffer (write-combining
    // stores, a fence, one read-back of the last word before the launch: posted writes may not pass it).  FW_PARAM_BAR=0: pinned.
    bool param_bar = false;
    // ... and so do op TABLES of at most kBarParamBytes (a few hundred emitters on the compacting / wave-per-type launches: their
    // workgroups read a header and an op each before they can spawn); larger tables stay in pinned memory, written in place
    // (OpList): with thousands of emitters the frame is bound by the host, and write-combining stores cost it more than cached ones
    char *b_param[kParamRing] = {};
    bool range_spread_new = true;   // FW_RANGE_SPREAD_NEW=0: a segment's NEW workgroups all in front of its YOUNG ones (A/B)
    // An in-place ring launch (FIFO / range) that streams more than nt_bytes uses the fully non-temporal form of its kernel:
    // several times the 256 MiB Infinity Cache, where allocating lines that cannot survive until the next frame only costs.
    // One that streams more than nt_wo_bytes -- no longer all of it fits -- stores the planes no update reads back (scale,
    // colours) non-temporally, so that the cache keeps what the next frame reads (fw_dev.h: fw_ld4w; knobs FW_NT_MB=n,
    // FW_NT_WO_MB=n: 0 = always)
    uint64_t nt_bytes = 768ull << 20;  // (crossover measured at 0.65-0.85 GB for both kernels: profiles/r03/nt_sweep.txt)
    // (the write-only form measured over 50-970 MB, profiles/r03/nt_sweep_wo.txt: range rings gain from the smallest size on
    // (-1..4 % below 250 MB, -10..16 % at 320-425 MB); the FIFO kernel's non-temporal forms carry the generic write mask, which
    // costs 1-2 % where everything fits the cache, and gain from ~300 MB on (-10..16 % at 480-650 MB))
    uint64_t nt_wo_bytes = 280ull << 20, nt_wo_bytes_range = 64ull << 20;
    unsigned long long *d_rts = nullptr;  // FW_DEBUG & 8: per-workgroup timestamps of the last range launch
    bool range_idle_last = true;   // FW_RANGE_IDLE_LAST=0: provisioned-but-idle workgroups stay next to their segment's active ones
    std::vector<uint32_t> range_scratch;
    size_t rparam_bytes = 0;
    uint64_t rslot_frame[kParamRing] = {};    // frame that last used the slot (+1; 0 = free)
    uint64_t rring_seq = 0;
    uint64_t r_uploads = 0;  // times the range table was re-sent (FW_HOST_PROF prints it)

    uint32_t nest_seq = 0;  // launches of fw_k_nest so far (tag of their look-back words)
    // Nested entries whose two particle types live in FIFO rings of one launch run INSIDE that launch (fw_kernels.h: FwFifoNest):
    // a steady configs[3] frame is ONE launch instead of fw_k_nest + a gap + the update.  FW_NEST_FUSE=0: always the separate pass
    bool nest_fuse = true;
    uint64_t fused_nest_frames = 0, nest_pass_frames = 0;  // frames of either kind so far (fw_debug_nest_frames)
    // FW_HOST_PROF=1: time spent in the sections of fw_step's host half (printed when the context is destroyed)
    bool trace = false;  // FW_TRACE
    bool host_prof = false;
    uint64_t host_prof_skip = 0;  // FW_HOST_PROF=n (n > 1): frames to skip first (fill, table uploads)
    double prof_ns[10] = {};
    uint64_t prof_frames = 0;
    uint64_t frame = 0;
    double sim_time = 0.0;  // sum of the dt of every step so far (lifetime windows)
    uint32_t parity = 0;
    unsigned long long stats_before_last = 0;
    bool stats_valid = false;

    // kernel timing
    bool timing = false;
    std::vector<hipEvent_t> tev;
    size_t tev_used = 0;
    uint64_t timing_particles_start = 0;
    double tev_overhead_ms = 0;  // duration of an empty hipEvent pair on this stream

    unsigned long long *live_ring = nullptr;  // caller-owned device ring of per-frame live totals
    uint32_t live_ring_n = 0;
    uint64_t live_ring_frames = 0;            // frames written since the ring was registered

    FwCollider *d_colliders = nullptr;  // device-resident analytic colliders (fw_ctx_set_colliders)
    uint32_t n_colliders = 0;
    size_t coll_cap = 0;                // records the device table holds
    // a new set travels as ONE copy in the context's stream (ordered behind the frames that read the old set, in front of the
    // frames that will read the new one: no synchronisation); the pinned staging is double-buffered
    FwCollider *h_coll[2] = {nullptr, nullptr};
    size_t h_coll_cap[2] = {0, 0};
    hipEvent_t ev_coll[2] = {nullptr, nullptr};
    bool coll_pending[2] = {false, false};
    uint64_t coll_seq = 0;
    float *d_aabb = nullptr;   // 256 partial boxes of the AABB query
    float *h_aabb = nullptr;   // pinned result {min.xyz, any, max.xyz, -}
    unsigned long long *d_total = nullptr;
    uint32_t *d_segids = nullptr;
};

namespace fwh {


#define FW_HIP(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return FW_EHIP;                                                                          \
        }                                                                                            \
    } while (0)

inline fw_status fail(fw_ctx *ctx, fw_status s, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return s;
}

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

template <typename T>
fw_status dev_reserve(fw_ctx *ctx, DevArray<T> &a, size_t need, size_t used) {
    if (need <= a.cap) return FW_OK;
    size_t ncap = std::max<size_t>(need, a.cap * 2 + 64);
    T *nd = nullptr;
    FW_HIP(ctx, hipMalloc((void **)&nd, ncap * sizeof(T)));
    FW_HIP(ctx, fw_memset_done(nd, 0, ncap * sizeof(T)));
    if (a.d && used) FW_HIP(ctx, hipMemcpy(nd, a.d, used * sizeof(T), hipMemcpyDeviceToDevice));
    if (a.d) FW_HIP(ctx, hipFree(a.d));
    a.d = nd;
    a.cap = ncap;
    return FW_OK;
}

// ---- shared functions (definitions: see the list of translation units above)
fw_status sync(fw_ctx *ctx);
fw_status join_side(fw_ctx *ctx);
fw_status ensure_max_seg(fw_ctx *ctx, uint32_t need);
uint32_t seg_live_tiles(const SegHost &s);
uint32_t seg_tiles(const SegHost &s, uint32_t vt_rounds = 1);
fw_status ensure_tile_arrays(fw_ctx *ctx);
fw_status ensure_range_arrays(fw_ctx *ctx);
uint32_t ring_head_exact(const SegHost &S, uint32_t count);
fw_status ensure_param_ring(fw_ctx *ctx, size_t bytes);
fw_status upload_seg(fw_ctx *ctx, uint32_t si);
fw_status alloc_seg_buffers(fw_ctx *ctx, SegHost &s, uint32_t capacity, bool want_destroyed);
fw_status refresh_rold(fw_ctx *ctx);
fw_status refresh_counts_exact(fw_ctx *ctx);
fw_status check_device_errors(fw_ctx *ctx);
fw_status realloc_segment(fw_ctx *ctx, uint32_t si, uint32_t ncap, bool make_general);
fw_status grow_segment(fw_ctx *ctx, uint32_t si, uint32_t need, bool at_least_double = true);
bool nested_fed_wants_growth(const SegHost &S);
fw_status fifo_to_general(fw_ctx *ctx, uint32_t si);
bool small_eligible(const fw_ctx *ctx, const SegHost &S);
void enter_small(fw_ctx *ctx, SegHost &S);
void leave_small(fw_ctx *ctx, SegHost &S);
void update_small_mode(fw_ctx *ctx);
void small_suspend(fw_ctx *ctx, SegHost &S);  // off the kernel, still eligible (a narrow type that became wide while wide_on is off)
fw_status drop_few_rings(fw_ctx *ctx);
bool fifo_may_become_range(const fw_ctx *ctx, const SegHost &S);
fw_status fifo_to_range(fw_ctx *ctx, uint32_t si);
fw_status spill_fifo_rings(fw_ctx *ctx);
fw_status leave_nospin(fw_ctx *ctx, uint32_t si);
fw_status set_derived(fw_ctx *ctx, uint32_t si, bool on, bool refill = true);
// does the type's update leave scale and colours to its readers?  (colliding types -- and `bigkeys` ones, which run on their
// kernels -- stay as they are: the feature path reads the stored planes)
static inline bool wants_derived(const fw_ctx *ctx, const SegHost &S) {
    return ctx->use_derived && !S.collides && (ctx->derive_all || S.inst != nullptr);
}
fw_status grow_nested_children(fw_ctx *ctx, SpawnerHost &sp, uint32_t parent_type, int depth = 0);
void copy_curve(CurveCopy &dst, int32_t kind, int32_t n, const float *times, const float *values, int stride);
fw_status validate_desc(fw_ctx *ctx, const fw_spawner_desc *d);
uint32_t derive_capacity(const fw_spawner_desc *d, uint32_t t, const std::vector<uint32_t> &caps, double *expect_live = nullptr);
void fill_randvec3(const fw_rand_vec3 &r, float &mn, float &mx, float &spread, float dir[4], float arc[4]);
uint32_t pad4(uint32_t n);
fw_status build_spawner(fw_ctx *ctx, int h, const fw_spawner_desc *d, const std::vector<uint64_t> *carry_serial);
fw_status release_spawner_segments(fw_ctx *ctx, SpawnerHost &sp);
fw_status update_tile_table(fw_ctx *ctx);
bool poll_device_error(fw_ctx *ctx);
fw_status poisoned_status(fw_ctx *ctx);
fw_status poison_segment(fw_ctx *ctx, uint32_t si, const std::string &what);
SpawnerHost *get_spawner(fw_ctx *ctx, fw_spawner h);
void poll_snapshots(fw_ctx *ctx);
fw_status read_counts(fw_ctx *ctx, std::vector<uint32_t> &out);
bool spawner_active(const fw_ctx *ctx, const SpawnerHost &sp, const std::vector<uint32_t> &counts);

}  // namespace fwh

// fw_engine.cpp -- host side of libfirework_hip.so: the C ABI of include/firework_hip.h.
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

#include "../../include/firework_hip.h"
#include "../../include/firework_hip_debug.h"
#include "fw_kernels.h"
#include "fw_math.h"

namespace {

// hipMemset runs on the null stream and may return before the fill has happened; the streams of a context are
// non-blocking ones, which the null stream does not order itself against: a kernel enqueued right after would race the fill
// (seen with two processes on the GPU: a fresh context's first frames read the forecast entries a previous context had left
// in the recycled allocation).  Every fill of this file is part of an allocation (rare): wait for it.
hipError_t fw_memset_done(void *p, int v, size_t bytes) {
    hipError_t e = hipMemset(p, v, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

constexpr int kParamRing = 8;    // per-frame parameter buffers in flight
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

thread_local std::string g_create_error;  // (fw_last_error(nullptr): per calling thread, like errno)

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
    // FW_TYPE_DERIVED (fw_device.h): an instance buffer is attached -- its records carry scale and colours, the planes S4 / Q5 /
    // Q6 are not stored by the update; every reader evaluates them from age / lifetime / initial_scale
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

}  // namespace

struct FwLevel {  // ops of one emission index (spawn order inside a frame: core.rs:377-428)
    std::vector<FwOp> g;
    std::vector<FwNestOp> n;
};

struct fw_ctx {
    std::vector<FwLevel> levels;  // per-frame scratch of fw_step: one entry per emission index in use
    std::vector<FwOp> ops_scratch;
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
    bool use_derived = true;   // FW_DERIVED=0: types with an attached instance buffer keep storing scale / colour planes (A/B)
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
    uint32_t range_few = 64;
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
    // table), 1024 (4) otherwise and always with an attached instance buffer.  FW_RANGE_YOUNG_BIG=n (0: never)
    uint32_t range_young_rounds = 4;
    uint32_t range_young_big = 32768;
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
    char *h_rparam[kParamRing] = {};          // pinned per-frame records + ops, read by the kernel in place
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

namespace {

#define FW_HIP(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return FW_EHIP;                                                                          \
        }                                                                                            \
    } while (0)

fw_status fail(fw_ctx *ctx, fw_status s, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return s;
}

uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

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

fw_status sync(fw_ctx *ctx) {
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side_dirty) {
        FW_HIP(ctx, hipStreamSynchronize(ctx->fifo_stream));
        ctx->side_dirty = false;
    }
    ctx->main_reads_ring = false;
    return FW_OK;
}

// before work that reads ring data is enqueued on the main stream without a synchronisation: the main stream waits for
// the ring launches on the side stream (and the next ring launch will wait for that work)
fw_status join_side(fw_ctx *ctx) {
    if (ctx->side_dirty) {
        FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
        FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        ctx->side_dirty = false;
    }
    ctx->main_reads_ring = true;
    return FW_OK;
}

// grows the [2][max_seg] bookkeeping arrays and the snapshot ring
fw_status ensure_max_seg(fw_ctx *ctx, uint32_t need) {
    if (need <= ctx->max_seg) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    uint32_t nmax = std::max<uint32_t>(need, ctx->max_seg ? ctx->max_seg * 2 : 1024);
    auto regrow2 = [&](uint32_t *&p) -> fw_status {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, 2ull * nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, 2ull * nmax * sizeof(uint32_t)));
        if (p) {
            for (int r = 0; r < 2; r++)
                FW_HIP(ctx, hipMemcpy(np + (size_t)r * nmax, p + (size_t)r * ctx->max_seg,
                                      ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(p));
        }
        p = np;
        return FW_OK;
    };
    if ((st = regrow2(ctx->g.count))) return st;
    if ((st = regrow2(ctx->g.spawned))) return st;
    if ((st = regrow2(ctx->g.appended))) return st;
    if ((st = regrow2(ctx->g.rold))) return st;
    {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, (size_t)nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, (size_t)nmax * sizeof(uint32_t)));
        if (ctx->g.ndestroyed) {
            FW_HIP(ctx, hipMemcpy(np, ctx->g.ndestroyed, ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(ctx->g.ndestroyed));
        }
        ctx->g.ndestroyed = np;
    }
    {
        unsigned long long *nh = nullptr;
        FW_HIP(ctx, hipHostMalloc((void **)&nh, (size_t)kSnapRing * nmax * sizeof(unsigned long long), hipHostMallocDefault));
        memset(nh, 0, (size_t)kSnapRing * nmax * sizeof(unsigned long long));
        if (ctx->h_snap) FW_HIP(ctx, hipHostFree(ctx->h_snap));
        ctx->h_snap = nh;
        for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    }
    {
        uint32_t *np = nullptr;
        FW_HIP(ctx, hipMalloc((void **)&np, (size_t)nmax * sizeof(uint32_t)));
        FW_HIP(ctx, fw_memset_done(np, 0, (size_t)nmax * sizeof(uint32_t)));
        if (ctx->g.range_ticket) {
            FW_HIP(ctx, hipMemcpy(np, ctx->g.range_ticket, ctx->max_seg * sizeof(uint32_t), hipMemcpyDeviceToDevice));
            FW_HIP(ctx, hipFree(ctx->g.range_ticket));
        }
        ctx->g.range_ticket = np;
    }
    if (ctx->d_segids) FW_HIP(ctx, hipFree(ctx->d_segids));
    FW_HIP(ctx, hipMalloc((void **)&ctx->d_segids, (size_t)nmax * sizeof(uint32_t)));
    ctx->max_seg = nmax;
    ctx->g.max_seg = nmax;
    return FW_OK;
}

// tiles the update grid must cover for a segment: the live region in tiles of FW_TILE, plus this frame's new
// particles in tiles of FW_VTILE (fw_k_update's tiling of the index space)
uint32_t seg_live_tiles(const SegHost &s) {
    const uint32_t live_ub = s.nested_fed ? s.capacity : std::min(s.ub - std::min(s.ub, s.frame_spawn), s.capacity);
    return (live_ub + FW_TILE - 1) / FW_TILE;
}
uint32_t seg_tiles(const SegHost &s, uint32_t vt_rounds = 1) {
    if (!s.in_use || s.ring() || s.small) return 0;  // (rings and small types have their own launches: fw_k_update_fifo / _range / _small)
    const uint32_t vtile = vt_rounds * FW_VTILE;
    if (!s.nested_fed && s.frame_spawn <= FW_VTILE) {
        // At most one round of new particles: they ride in the last live tile whenever it has room for them (both
        // update kernels apply the same rule), otherwise they take one tile of their own right behind it -- either
        // way ceil((live + new) / FW_TILE) tiles, for any live count up to the bound.  Thousands of small emitters
        // then cost ONE workgroup each instead of three (every workgroup pays ~5 us of launch-time latencies).
        const uint64_t live_ub = std::min(s.ub - std::min(s.ub, s.frame_spawn), s.capacity);
        return std::max<uint32_t>(1, (uint32_t)((live_ub + s.frame_spawn + FW_TILE - 1) / FW_TILE));
    }
    return std::max<uint32_t>(1, seg_live_tiles(s) + (s.frame_spawn + vtile - 1) / vtile + 1);
}

// (size of the new-particle tiles of a frame, update_tile_table: the smallest -- most parallel -- of 1 or 2 rounds for
// which all ACTIVE tiles of the frame are resident at once (kResidentSlots workgroups: 4 per CU); a second, nearly
// empty round of workgroups would cost a full tile lifetime)

// tile scratch sized for every segment at full capacity
fw_status ensure_tile_arrays(fw_ctx *ctx) {
    size_t tiles = 0, nest_tiles = 0, nest_ops = 0;
    for (auto &s : ctx->segs)
        if (s.in_use && !s.ring()) tiles += (s.capacity + FW_VTILE - 1) / FW_VTILE + 2;  // worst case: every slot a new particle
    for (auto &sp : ctx->spawners) {
        if (!sp.alive) continue;
        for (auto &e : sp.em)
            if (e.es.mode == FW_MODE_NESTED) {
                nest_tiles += (ctx->segs[sp.seg[e.es.target_particle_type]].capacity + FW_NEST_TILE - 1) / FW_NEST_TILE + 1;
                nest_ops++;
            }
    }
    if (tiles > ctx->tiles_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = tiles * 2;
        if (ctx->g.tile_cnt) hipFree(ctx->g.tile_cnt), hipFree(ctx->g.tile_off), hipFree(ctx->g.tile_status);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_cnt, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_off, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_status, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.tile_status, 0, ncap * sizeof(unsigned long long)));
        if (ctx->g.tile_box) hipFree(ctx->g.tile_box);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.tile_box, ncap * 8 * sizeof(float)));
        FW_HIP(ctx, fw_memset_done(ctx->g.tile_box, 0, ncap * 8 * sizeof(float)));
        ctx->boxes_epoch = 0;
        if (ctx->g.dbg_ts) hipFree(ctx->g.dbg_ts);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.dbg_ts, (32768 + 8 * ncap) * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.dbg_ts, 0, (32768 + 8 * ncap) * sizeof(unsigned long long)));
        if (ctx->d_fce) hipFree(ctx->d_fce);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_fce, 2 * ncap * sizeof(uint4)));
        FW_HIP(ctx, fw_memset_done(ctx->d_fce, 0, 2 * ncap * sizeof(uint4)));
        if (ctx->d_fc) hipFree(ctx->d_fc);
        ctx->fc_len = ncap + (ncap / 64 + 2) * FW_FC_S2_STRIDE + 8;  // P | P2 | tag (64-bit words)
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_fc, 3 * ctx->fc_len * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->d_fc, 0, 3 * ctx->fc_len * sizeof(unsigned long long)));
        ctx->fc_ok = false, ctx->boxes_epoch = 0;
        ctx->fc_dirty = false;
        ctx->tiles_cap = ncap;
    }
    if (nest_tiles > ctx->nest_tiles_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = nest_tiles * 2;
        if (ctx->g.nest_status) hipFree(ctx->g.nest_status);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.nest_status, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.nest_status, 0, ncap * sizeof(unsigned long long)));
        ctx->nest_tiles_cap = ncap;
    }
    if (nest_ops > ctx->nest_ops_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        size_t ncap = nest_ops * 2 + 16;
        if (ctx->g.nest_ticket) hipFree(ctx->g.nest_ticket);
        FW_HIP(ctx, hipMalloc((void **)&ctx->g.nest_ticket, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->g.nest_ticket, 0, ncap * sizeof(unsigned long long)));
        ctx->nest_ops_cap = ncap;
    }
    return FW_OK;
}

// device table, look-back words and per-frame pinned records of the range launch, sized for the worst case of every range
// segment (old + young workgroups cover at most the ring, new ones at most a ring of new particles) when a segment is built
// or reallocated: fw_step itself never allocates for them
fw_status ensure_range_arrays(fw_ctx *ctx) {
    size_t tiles = 0;
    for (auto &s : ctx->segs)
        if (s.in_use && s.range)  // OLD tiles of FW_TILE, YOUNG tiles of the build's size, NEW workgroups of FW_BLOCK
            // (... or, a launch on one-round tiles -- fw_ctx::range_small -- OLD and YOUNG tiles of FW_BLOCK)
            tiles += 2 * ((size_t)s.capacity / FW_BLOCK + 4) + (size_t)s.capacity / FW_BLOCK + 2;
    if (tiles > ctx->rdesc_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = tiles + tiles / 2 + 64;
        if (ctx->d_rdesc) hipFree(ctx->d_rdesc);
        if (ctx->h_rdesc) hipHostFree(ctx->h_rdesc);
        if (ctx->d_rstatus) hipFree(ctx->d_rstatus);
        ctx->d_rdesc = nullptr, ctx->h_rdesc = nullptr, ctx->d_rstatus = nullptr, ctx->rdesc_cap = 0;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_rdesc, ncap * sizeof(FwRangeDesc)));
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_rdesc, ncap * sizeof(FwRangeDesc), hipHostMallocDefault));
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_rstatus, ncap * sizeof(unsigned long long)));
        FW_HIP(ctx, fw_memset_done(ctx->d_rstatus, 0, ncap * sizeof(unsigned long long)));
        if (ctx->dbg & 8u) {  // per-workgroup timestamps of the last range launch (tools/range_timeline.py)
            if (ctx->d_rts) hipFree(ctx->d_rts);
            ctx->d_rts = nullptr;
            FW_HIP(ctx, hipMalloc((void **)&ctx->d_rts, ncap * 8 * sizeof(unsigned long long)));
            FW_HIP(ctx, fw_memset_done(ctx->d_rts, 0, ncap * 8 * sizeof(unsigned long long)));
        }
        ctx->rdesc_cap = ncap;
        ctx->r_force = true, ctx->rtab_pending = false;
    }
    // one record per segment slot + one op per emission entry of the context
    const size_t need = round_up((uint32_t)(ctx->max_seg * sizeof(FwRangeRec)), 64) + (size_t)(ctx->n_emits + 8) * sizeof(FwOp) + 64;
    if (ctx->n_range && need > ctx->rparam_bytes) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t nb = need * 2;
        for (int i = 0; i < kParamRing; i++) {
            if (ctx->h_rparam[i]) hipHostFree(ctx->h_rparam[i]);
            ctx->h_rparam[i] = nullptr;
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_rparam[i], nb, hipHostMallocDefault));
            memset(ctx->h_rparam[i], 0, nb);
            ctx->rslot_frame[i] = 0;
        }
        ctx->rparam_bytes = nb;
    }
    return FW_OK;
}

// slot of particle 0 of a segment whose live count is `count` (exact): 0 unless the segment is a ring
// (a range ring: the old part sits right before the young part; its size is the device's -- rold_seen, refreshed together
// with the exact counts -- for a type that receives Nested children, and count - young_n, the same number, otherwise)
uint32_t ring_head_exact(const SegHost &S, uint32_t count) {
    if (S.fifo) return S.head;
    if (!S.range) return 0u;
    const uint32_t n_old = S.range_dev ? std::min(S.rold_seen, count) : (count > S.young_n ? count - S.young_n : 0u);
    return (uint32_t)(((uint64_t)S.young_lo + S.capacity - (n_old % S.capacity)) % S.capacity);
}

fw_status ensure_param_ring(fw_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->param_bytes) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    size_t nb = std::max<size_t>(bytes * 2, 1 << 16);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_param[i]) hipHostFree(ctx->h_param[i]), hipFree(ctx->d_param[i]);
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_param[i], nb, hipHostMallocDefault));
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_param[i], nb));
        ctx->consumed_pending[i] = false;
    }
    ctx->param_bytes = nb;
    return FW_OK;
}

fw_status upload_seg(fw_ctx *ctx, uint32_t si) {
    const SegHost &s = ctx->segs[si];
    FwSeg d{};
    d.buf[0] = s.buf[0], d.buf[1] = s.buf[1];
    d.destroyed = s.destroyed;
    d.capacity = s.capacity;
    d.type_idx = s.type_idx;
    d.n_lplanes = s.n_lplanes;
    d.inst = s.inst, d.inst_cap = s.inst_cap;
    d.lplane_emit[0] = d.lplane_emit[1] = 0xFFFFFFFFu;
    if (s.virt_parent && s.spawner >= 0)
        for (uint32_t k = 0; k < s.n_lplanes && k < 2u; k++) {
            const auto &em = ctx->spawners[s.spawner].em;
            const int32_t ei = s.lplane_emission[k];
            if (ei >= 0 && (size_t)ei < em.size() && em[ei].assigned) d.lplane_emit[k] = em[ei].emit_idx;
        }
    FW_HIP(ctx, hipMemcpy(ctx->d_segs.d + si, &d, sizeof d, hipMemcpyHostToDevice));
    return FW_OK;
}

fw_status alloc_seg_buffers(fw_ctx *ctx, SegHost &s, uint32_t capacity, bool want_destroyed) {
    const size_t bytes = FW_BUF_BYTES((size_t)capacity, s.n_lplanes + s.n_xplanes);
    char *b = nullptr;
    hipError_t e = hipMalloc((void **)&b, bytes * (s.ring() ? 1 : 2));  // a ring is updated in place: one buffer
    if (e != hipSuccess) return fail(ctx, FW_ENOMEM, std::string("hipMalloc particle buffers: ") + hipGetErrorString(e));
    s.buf[0] = b;
    s.buf[1] = s.ring() ? b : b + bytes;
    s.capacity = capacity;
    s.destroyed = nullptr;
    if (want_destroyed) {
        e = hipMalloc((void **)&s.destroyed, (size_t)capacity * sizeof(fw_particle));
        if (e != hipSuccess) return fail(ctx, FW_ENOMEM, "hipMalloc destroyed buffer");
    }
    FW_HIP(ctx, fw_launch_fill_colors(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], capacity, s.fill_bc, s.fill_em));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));  // callers go on with blocking copies on the null stream
    return FW_OK;
}

// the size of the old part of every range ring that receives Nested children, as the device has it (the stream has been
// waited for): what ring_head_exact derives such a ring's first slot from
fw_status refresh_rold(fw_ctx *ctx) {
    bool any = false;
    for (const SegHost &S : ctx->segs) any |= S.in_use && S.range && S.range_dev;
    if (!any) return FW_OK;
    const uint32_t n = (uint32_t)ctx->segs.size();
    std::vector<uint32_t> r(n);
    FW_HIP(ctx, hipMemcpy(r.data(), ctx->g.rold + (size_t)ctx->parity * ctx->max_seg, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
        if (ctx->segs[i].in_use && ctx->segs[i].range) ctx->segs[i].rold_seen = r[i];
    return FW_OK;
}

// exact device counts -> host upper bounds (synchronises)
fw_status refresh_counts_exact(fw_ctx *ctx) {
    fw_status st = sync(ctx);
    if (st) return st;
    const uint32_t n = (uint32_t)ctx->segs.size();
    if (!n) return FW_OK;
    std::vector<uint32_t> c(n);
    FW_HIP(ctx, hipMemcpy(c.data(), ctx->g.count + (size_t)ctx->parity * ctx->max_seg, n * sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++)
        if (ctx->segs[i].in_use) ctx->segs[i].ub = c[i];
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    return refresh_rold(ctx);
}

bool poll_device_error(fw_ctx *ctx);
fw_status poisoned_status(fw_ctx *ctx);
fw_status check_device_errors(fw_ctx *ctx) {
    // (the stream has been waited for: whatever a kernel reported is in the pinned word by now)
    const bool fresh = poll_device_error(ctx);
    // (every kernel that sets a flag says so in the pinned word next to err_host -- fw_flag: nothing there, nothing to fetch)
    if (!*(const volatile unsigned long long *)(ctx->h_err + 1) && !ctx->trace) return FW_OK;
    ctx->h_err[1] = 0ull;
    uint32_t ev[8] = {};
    FW_HIP(ctx, hipMemcpy(ev, ctx->g.err, sizeof ev, hipMemcpyDeviceToHost));
    const uint32_t e = ev[0];
    if (ctx->trace && ctx->d_tile_first) {
        uint32_t t[2] = {77, 77};
        hipMemcpy(t, ctx->d_tile_first, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[fw] check: flags=%u table=[%u,%u] ptr=%p fc=%p tiles_cap=%zu\n", e, t[0], t[1],
                (void *)ctx->d_tile_first, (void *)ctx->d_fc, ctx->tiles_cap);
    }
    if (!e) return FW_OK;
    uint32_t zero = 0;
    FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    // (the two bits are looked at independently -- a word with both set used to lose the capacity report when the internal
    // error was not news: ADVICE r04 -- and the more severe status wins)
    fw_status cap = FW_OK;
    if (e & FW_ERR_CAPACITY)
        cap = fail(ctx, FW_ECAPACITY,
                   "a particle type overflowed its device capacity; particles were dropped "
                   "(raise fw_particle_settings.capacity) [device flags " + std::to_string(e) + ", segment " +
                       std::to_string(ev[1]) + ": " + std::to_string(ev[2]) + " particles (" + std::to_string(ev[4]) +
                       " resident), " + std::to_string(ev[3]) + " tiles launched]");
    if ((e & FW_ERR_FORECAST) && fresh) {
        // An internal check of an update kernel failed.  The spawner it names is marked (poll_device_error): its own calls
        // refuse from now on.  Whoever synchronises first is told once; later synchronisations -- of healthy spawners -- are
        // not failed again for an error that has been reported and contained.
        ctx->poison_msg += " [device flags " + std::to_string(e) + ", check " + std::to_string(ev[5]) + ": " + std::to_string(ev[6]) +
                           " " + std::to_string(ev[7]) + "]";
        return poisoned_status(ctx);
    }
    return cap;  // FW_ERR_LOOKBACK_TIMEOUT is informational: the fallback path produced the same result
}

// moves a segment into freshly allocated buffers of `ncap` slots: its live particles, in order, from slot 0 (a FIFO ring
// is unwrapped); make_general: a FIFO segment leaves that mode (two buffers, compacting update) on the way
fw_status realloc_segment(fw_ctx *ctx, uint32_t si, uint32_t ncap, bool make_general) {
    SegHost &s = ctx->segs[si];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    fw_status st = refresh_counts_exact(ctx);
    if (st) return st;
    SegHost old = s;
    if (s.fifo && ncap >= 0x40000000u) make_general = true;  // ring slots are computed in 32 bits: head + index < 2^32
    if (s.range && ncap > FW_RANGE_MAX_CAPACITY) make_general = true;  // (32-bit byte offsets into a plane)
    if (make_general && s.fifo) {
        s.fifo = false, s.fifo_mat = s.fifo_dev = false, s.coh.clear();
        ctx->n_fifo--;
        ctx->tab_force = true;
        ctx->seg_kind_changed = true;
    }
    if (make_general && s.range) {
        s.range = false, s.ycoh.clear(), s.dcoh.clear(), s.gcoh.clear(), s.gcoh_sum = 0, s.young_lo = s.young_n = 0;
        s.range_mat = s.range_dev = false;
        ctx->n_range--;
        if (s.few_ring) s.few_ring = false, ctx->n_few--;
        if (s.spilled) s.spilled = false, ctx->n_spilled--;
        ctx->tab_force = true, ctx->r_force = true;
        ctx->seg_kind_changed = true;
    }
    st = alloc_seg_buffers(ctx, s, ncap, old.destroyed != nullptr);
    if (st) {
        if (old.fifo && !s.fifo) ctx->n_fifo++;
        if (old.range && !s.range) ctx->n_range++;
        if (old.few_ring && !s.few_ring) ctx->n_few++;
        if (old.spilled && !s.spilled) ctx->n_spilled++;
        s = old;
        return st;
    }
    if (((old.fifo && !s.fifo) || (old.range && !s.range)) && s.h_report) hipHostFree(s.h_report), s.h_report = nullptr;
    s.head = 0;
    const uint32_t p = ctx->parity;
    const uint32_t n = old.ub;  // exact after the refresh
    const uint32_t h = ring_head_exact(old, n);
    if (s.range) {  // the list now starts in slot 0: old part first, the young part right behind it
        s.young_lo = old.range_dev ? std::min(old.rold_seen, n) : (n > old.young_n ? n - old.young_n : 0u);
        ctx->r_force = true;
    }
    const uint32_t n1 = std::min<uint32_t>(n, old.capacity - h);  // up to the end of the old buffer, then from its slot 0
    auto cp = [&](size_t noff, size_t ooff, size_t elem) -> hipError_t {
        hipError_t e = hipSuccess;
        if (n1) e = hipMemcpy(s.buf[p] + noff, old.buf[p] + ooff + (size_t)h * elem, (size_t)n1 * elem, hipMemcpyDeviceToDevice);
        if (e == hipSuccess && n > n1)
            e = hipMemcpy(s.buf[p] + noff + (size_t)n1 * elem, old.buf[p] + ooff, (size_t)(n - n1) * elem, hipMemcpyDeviceToDevice);
        return e;
    };
    const size_t OC = old.capacity, NC = ncap;
    FW_HIP(ctx, cp(FW_OFF_Q0(NC), FW_OFF_Q0(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q1(NC), FW_OFF_Q1(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q2(NC), FW_OFF_Q2(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q3(NC), FW_OFF_Q3(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q5(NC), FW_OFF_Q5(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_Q6(NC), FW_OFF_Q6(OC), 16));
    FW_HIP(ctx, cp(FW_OFF_S4(NC), FW_OFF_S4(OC), 4));
    for (uint32_t k = 0; k < s.n_lplanes + s.n_xplanes; k++) FW_HIP(ctx, cp(FW_OFF_L(NC, k), FW_OFF_L(OC, k), 4));
    if (old.destroyed) {  // the records of the last update stay readable (fw_spawner_read_destroyed)
        const size_t m = std::min(old.capacity, ncap);  // (from the start of the buffer, or -- a range ring's -- up to its end)
        const size_t so = old.dead_at_end ? (size_t)old.capacity - m : 0, dof = old.dead_at_end ? (size_t)ncap - m : 0;
        FW_HIP(ctx, hipMemcpy(s.destroyed + dof * sizeof(fw_particle), old.destroyed + so * sizeof(fw_particle), m * sizeof(fw_particle),
                              hipMemcpyDeviceToDevice));
    }
    FW_HIP(ctx, hipFree(old.buf[0]));
    if (old.destroyed) FW_HIP(ctx, hipFree(old.destroyed));
    if (old.fifo && !s.fifo) {
        s.win_ok = false;  // no lifetime window was kept: the bound follows the snapshots from here on
        if (s.nospin) {  // a ring keeps no lifetime plane (one value); the compacting kernels read it
            FW_HIP(ctx, fw_launch_fill_plane1(ctx->stream, s.buf[0], s.buf[1], FW_OFF_L((size_t)s.capacity, s.n_lplanes), s.capacity,
                                              s.fifo_life));
            FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if ((st = upload_seg(ctx, si))) return st;
    if ((st = ensure_range_arrays(ctx))) return st;
    return ensure_tile_arrays(ctx);
}

// (at_least_double: the amortised growth of a Vec -- `need` with a quarter of slack, never less than twice the capacity;
// false: exactly what the caller asks for)
fw_status grow_segment(fw_ctx *ctx, uint32_t si, uint32_t need, bool at_least_double = true) {
    const SegHost &s = ctx->segs[si];
    const uint32_t ncap = round_up(at_least_double ? std::max<uint32_t>((uint32_t)std::min<uint64_t>((uint64_t)need * 5 / 4, 0xFFFF0000ull),
                                                                        s.capacity * 2)
                                                   : std::max<uint32_t>(need, s.capacity),
                                   std::max<uint32_t>(FW_TILE, fw_range_young_tile()));
    return realloc_segment(ctx, si, ncap, false);
}

// Types fed by Nested entries cannot be bounded by the host (children are counted per parent on the device), so they cannot
// grow exactly when needed the way Global-fed ones do (the reference's Vec::push, core.rs:523).  Their DERIVED capacity
// (parent capacity x children per parent x lifetime ratio x 1.25) is an upper estimate of the steady state already: such a
// segment grows when the count seen in the snapshot rows passes 85 % of it, or when at the rate it was last seen growing
// it would fill up within 64 frames (the rows a free-running host looks at are up to a dozen frames old) -- long before the
// device-side clamp (FW_ECAPACITY) could drop a particle, and without the 2x over-allocation and the ~30 ms reallocation
// the old "half full" rule cost a steady configs[3].
bool nested_fed_wants_growth(const SegHost &S) {
    if (!S.nested_fed || !S.auto_capacity || S.capacity >= 0x70000000u) return false;
    return (double)S.dev_count > 0.85 * (double)S.capacity || (double)S.dev_count + 64.0 * (double)S.dev_rate > (double)S.capacity;
}

// a FIFO segment whose premise no longer holds (the caller wrote particles, dt went negative or non-finite, ...)
// continues as an ordinary segment
fw_status fifo_to_general(fw_ctx *ctx, uint32_t si) {
    if (!ctx->segs[si].ring()) return FW_OK;
    return realloc_segment(ctx, si, ctx->segs[si].capacity, true);
}

// SegHost::small: may this compacting segment be updated by the wave-per-type kernel?
bool small_eligible(const fw_ctx *ctx, const SegHost &S) {
    return ctx->use_small && S.in_use && !S.ring() && !S.nested_fed && S.n_lplanes == 0 && !S.collides && S.inst == nullptr && !ctx->track_aabb &&
           !S.colors_dirty && S.expect_live * 2.0f <= (float)ctx->small_max;
}
void enter_small(fw_ctx *ctx, SegHost &S) {
    if (S.small) return;
    S.small = true, ctx->n_small++, ctx->small_dirty = true;
    ctx->tab_force = true, ctx->fc_ok = false, ctx->boxes_epoch = 0;
}
// ... and back: the segment is a compacting segment again (the same buffers; the tile table is re-sent)
void leave_small(fw_ctx *ctx, SegHost &S) {
    if (!S.small) return;
    S.small = false, ctx->n_small--, ctx->small_dirty = true;
    ctx->tab_force = true, ctx->fc_ok = false, ctx->boxes_epoch = 0;
}

// every SegHost::few_ring segment leaves its ring (fw_ctx::range_few), particles and order kept
fw_status drop_few_rings(fw_ctx *ctx) {
    for (uint32_t si = 0; si < ctx->segs.size() && ctx->n_few; si++) {
        if (!ctx->segs[si].in_use || !ctx->segs[si].few_ring) continue;
        const fw_status st = fifo_to_general(ctx, si);  // (realloc_segment clears the flag and the count)
        if (st) return st;
        if (small_eligible(ctx, ctx->segs[si])) enter_small(ctx, ctx->segs[si]);  // (a small type: the wave-per-type kernel from here on)
    }
    return FW_OK;
}

// can this FIFO ring continue as a RANGE ring (build_spawner's rule for range rings, for a type that is a FIFO ring already)
bool fifo_may_become_range(const fw_ctx *ctx, const SegHost &S) {
    if (!S.in_use || !S.fifo || !ctx->use_range || S.spawner < 0) return false;
    const SpawnerHost &sp = ctx->spawners[S.spawner];
    return !sp.no_rings && S.n_lplanes <= 2 && sp.types[S.type].life_lo_safe > 0.0f && S.capacity <= FW_RANGE_MAX_CAPACITY &&
           S.capacity % std::max<uint32_t>(FW_TILE, fw_range_young_tile()) == 0u && !(S.inst != nullptr && !S.inst_window);
}

// A FIFO ring becomes a RANGE ring where it stands (fw_ctx::n_spilled): the same buffer, the same slots, nothing copied.  A FIFO
// ring is a range ring whose particles are all "young" -- nobody has been told yet that it may die -- with an empty old part: the
// slot of the first young particle is the head, the young cohorts are the FIFO cohorts (frame of birth + size; sizes the device
// alone knows stay in the pinned report ring), their ages -- one table for all range rings of the context, fw_ctx::birth_age -- are
// the ages the FIFO replay kept per cohort: the same fp32 additions, bit for bit.  The next fw_step moves the cohorts that may
// die in it to the old part as for any range ring.  The context is synchronised (build time).
fw_status fifo_to_range(fw_ctx *ctx, uint32_t si) {
    if (!fifo_may_become_range(ctx, ctx->segs[si])) return FW_OK;
    fw_status st = refresh_counts_exact(ctx);  // (a ring that receives Nested children: only the device knows its count)
    if (st) return st;
    SegHost &S = ctx->segs[si];
    const TypeHost &T = ctx->spawners[S.spawner].types[S.type];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    // ---- the ages of the frames its cohorts were born in: fw_ctx::birth_age holds one entry per frame, contiguous up to the
    // current frame; it is extended backwards to the oldest cohort.  Frames in which this ring received nothing get the age of
    // the next older cohort (never asked for by it; at least as old as the true one, so the pruning order holds) -- and are
    // overwritten with the exact value by whichever ring does hold a cohort of that frame.
    if (!S.coh.empty()) {
        auto &B = ctx->birth_age;
        const uint64_t have_from = B.empty() ? ctx->frame : B.front().frame;
        if (S.coh.front().frame < have_from) {
            std::vector<fw_ctx::BirthAge> pre;
            size_t ci = 0;
            float age = S.coh.front().age;
            for (uint64_t f = S.coh.front().frame; f < have_from; f++) {
                while (ci < S.coh.size() && S.coh[ci].frame < f) ci++;
                if (ci < S.coh.size() && S.coh[ci].frame == f) age = S.coh[ci].age;
                pre.push_back(fw_ctx::BirthAge{f, age});
            }
            B.insert(B.begin(), pre.begin(), pre.end());
        }
        for (const SegHost::Cohort &c : S.coh)
            if (!B.empty() && c.frame >= B.front().frame && c.frame - B.front().frame < B.size()) B[(size_t)(c.frame - B.front().frame)].age = c.age;
    }
    // ---- the ring's own bookkeeping
    S.fifo = false, ctx->n_fifo--;
    S.range = true, ctx->n_range++;
    S.spilled = true, ctx->n_spilled++;
    S.young_lo = S.head, S.head = 0;
    S.range_life_lo = T.life_lo_safe;
    ctx->range_life_max = std::max(ctx->range_life_max, S.range_life_lo);
    S.range_mat = S.n_lplanes != 0, S.range_dev = S.nested_fed;
    S.ycoh.clear(), S.dcoh.clear(), S.gcoh.clear(), S.gcoh_sum = 0, S.rold_seen = 0, S.r_young_main = 0;
    S.r_old = S.r_new = S.r_young = 0;
    S.win.clear(), S.win_sum = 0, S.win_ok = false;
    if (S.range_dev) {
        S.young_n = 0;
        ctx->range_age_keep = std::max(ctx->range_age_keep, (float)(S.life_bound * 1.01 + 1e-3));
        for (const SegHost::Cohort &c : S.coh) S.dcoh.push_back(SegHost::DCohort{c.frame, c.known ? c.n : 0u, c.known});
        // (h_report stays: the update of a range ring that receives children leaves each frame's cohort size in the same ring)
    } else {
        uint64_t sum = 0;
        for (const SegHost::Cohort &c : S.coh) {
            if (!c.n) continue;
            S.ycoh.push_back(SegHost::YCohort{c.frame, c.n});
            // the lifetime window (the bound of the old part follows it): the time of the spawn from the cohort's age -- an fp32
            // running sum, whose distance from the exact time the window's horizon allows for (fw_step)
            S.win.push_back(SegHost::Spawned{ctx->sim_time - (double)c.age, c.n, c.frame});
            sum += c.n;
        }
        S.young_n = (uint32_t)sum;  // (= the exact live count: everybody is young)
        S.win_sum = sum, S.win_ok = std::isfinite(S.life_bound);
        if (S.h_report) hipHostFree(S.h_report), S.h_report = nullptr;
    }
    S.coh.clear();
    S.fifo_mat = S.fifo_dev = false;
    // ---- the device's share: the old part is empty; a type that cannot turn keeps its lifetimes in a plane of their own on
    // this path (a FIFO ring has one value and no plane)
    const uint32_t zero = 0;
    for (int r = 0; r < 2; r++) FW_HIP(ctx, hipMemcpy(ctx->g.rold + (size_t)r * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
    if (S.nospin) {
        FW_HIP(ctx, fw_launch_fill_plane1(ctx->stream, S.buf[0], nullptr, FW_OFF_L((size_t)S.capacity, S.n_lplanes), S.capacity, S.fifo_life));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->tab_force = true, ctx->r_force = true;
    return ensure_range_arrays(ctx);
}

// every FIFO ring of the context that may becomes a range ring (fw_ctx::n_spilled)
fw_status spill_fifo_rings(fw_ctx *ctx) {
    for (uint32_t si = 0; si < ctx->segs.size() && ctx->n_fifo; si++) {
        const fw_status st = fifo_to_range(ctx, si);
        if (st) return st;
    }
    return FW_OK;
}

// A type stops being FW_TYPE_NOSPIN (the caller rewrites its particles, a non-finite dt is stepped): the rotation plane,
// which nobody maintained, gets the constant rotation in every slot, then the flag goes.
fw_status leave_nospin(fw_ctx *ctx, uint32_t si) {
    SegHost &s = ctx->segs[si];
    if (!s.nospin) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, fw_launch_fill_rotation(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], s.capacity, s.const_rot));
    FW_HIP(ctx, fw_launch_restore_q3(ctx->stream, s.buf[0], s.ring() ? nullptr : s.buf[1], s.capacity,
                                     s.fifo ? 0xFFFFFFFFu : s.n_lplanes, s.fifo_life));
    FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t flags = s.derived ? FW_TYPE_DERIVED : 0u;
    FW_HIP(ctx, hipMemcpy((char *)(ctx->d_types.d + s.type_idx) + offsetof(FwType, flags), &flags, sizeof flags, hipMemcpyHostToDevice));
    s.nospin = false;
    ctx->tab_force = true, ctx->r_force = true;  // (the tile descriptors carry the flag)
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    return FW_OK;
}

// FW_TYPE_DERIVED on / off (SegHost::derived).  Off: the planes nobody maintained are filled from age / lifetime / initial_scale
// first (`refill` false when the caller is about to overwrite the particles anyway).
fw_status set_derived(fw_ctx *ctx, uint32_t si, bool on, bool refill = true) {
    SegHost &s = ctx->segs[si];
    if (s.derived == on) return FW_OK;
    fw_status st = sync(ctx);
    if (st) return st;
    if (!on && refill) {
        FW_HIP(ctx, fw_launch_rederive(ctx->stream, s.buf[ctx->parity], s.capacity, ctx->d_types.d + s.type_idx, ctx->d_keys.d, s.nospin,
                                       s.life_plane(), s.fifo_life));
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    uint32_t flags = (s.nospin ? FW_TYPE_NOSPIN : 0u) | (on ? FW_TYPE_DERIVED : 0u);
    FW_HIP(ctx, hipMemcpy((char *)(ctx->d_types.d + s.type_idx) + offsetof(FwType, flags), &flags, sizeof flags, hipMemcpyHostToDevice));
    s.derived = on;
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    return FW_OK;
}

// A parent type grew: the types its particles emit onto (Nested entries targeting it) were sized from the parent's
// capacity (derive_capacity) and cannot grow on demand themselves -- their counts are only known on the device -- so
// they follow the parent now, by the same rule.  Types with a caller-given capacity are left alone.
fw_status grow_nested_children(fw_ctx *ctx, SpawnerHost &sp, uint32_t parent_type, int depth = 0) {
    if (depth > (int)sp.types.size()) return FW_OK;
    const double pcap = ctx->segs[sp.seg[parent_type]].capacity;
    for (const EmissionHost &E : sp.em) {
        const fw_emission_settings &es = E.es;
        if (es.mode != FW_MODE_NESTED || (uint32_t)es.target_particle_type != parent_type) continue;
        if (es.pacing_kind != FW_PACING_COUNT_OVER_DURATION || !(es.count > 0)) continue;
        const uint32_t ct = (uint32_t)es.particle_index;
        if (ct == parent_type) continue;
        const fw_particle_settings &cp = sp.types[ct].ps, &pp = sp.types[parent_type].ps;
        if (cp.capacity) continue;
        const double life = std::max(0.0, (double)std::max(cp.lifetime.min, cp.lifetime.max));
        const double plife = std::max(1e-3, (double)std::min(pp.lifetime.min, pp.lifetime.max));
        double need = pcap * (double)es.count * std::max(1.0, life / plife + 0.1) * 1.25 + kMinCapacity;
        if (need > 3.0e9) need = 3.0e9;
        SegHost &C = ctx->segs[sp.seg[ct]];
        if (need > (double)C.capacity) {
            fw_status st = grow_segment(ctx, sp.seg[ct], (uint32_t)need);
            if (st) return st;
            if ((st = grow_nested_children(ctx, sp, ct, depth + 1))) return st;
        }
    }
    return FW_OK;
}

void copy_curve(CurveCopy &dst, int32_t kind, int32_t n, const float *times, const float *values, int stride) {
    dst.kind = kind;
    dst.n = n;
    dst.values.assign(values, values + (size_t)n * stride);
    dst.times.assign((size_t)n, 0.f);
    if (times && kind == FW_CURVE_UNEVEN) dst.times.assign(times, times + n);
    if (kind == FW_CURVE_UNEVEN && n >= 2) {
        // bevy_math UnevenCore::new: drop non-finite times, stable sort by time, dedup keeping the first
        std::vector<int> idx;
        for (int i = 0; i < n; i++)
            if (std::isfinite(dst.times[i])) idx.push_back(i);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return dst.times[a] < dst.times[b]; });
        std::vector<float> t, v;
        for (int i : idx) {
            if (!t.empty() && t.back() == dst.times[i]) continue;
            t.push_back(dst.times[i]);
            v.insert(v.end(), dst.values.begin() + (size_t)i * stride, dst.values.begin() + (size_t)(i + 1) * stride);
        }
        dst.times = t;
        dst.values = v;
        dst.n = (int32_t)t.size();
    }
    if (dst.n == 1) dst.kind = FW_CURVE_CONSTANT;  // curve.rs:46-49: one sample -> ConstantCurve
}

fw_status validate_desc(fw_ctx *ctx, const fw_spawner_desc *d) {
    if (!d) return fail(ctx, FW_EINVAL, "null descriptor");
    // (Vec<ParticleSettings> / Vec<EmissionSettings> of any length, core.rs:178-185; the counts are 32-bit here)
    if ((d->n_particle_settings && !d->particle_settings) || (d->n_emission_settings && !d->emission_settings))
        return fail(ctx, FW_EINVAL, "null settings array");
    for (uint32_t i = 0; i < d->n_particle_settings; i++) {
        const fw_particle_settings &p = d->particle_settings[i];
        const int32_t ns[3] = {p.scale_curve.n, p.base_color.n, p.emissive_color.n};
        const int32_t ks[3] = {p.scale_curve.kind, p.base_color.kind, p.emissive_color.kind};
        const void *vs[3] = {p.scale_curve.values, p.base_color.rgba, p.emissive_color.rgba};
        const void *ts[3] = {p.scale_curve.times, p.base_color.times, p.emissive_color.times};
        for (int k = 0; k < 3; k++) {
            if (ns[k] < 1) return fail(ctx, FW_EINVAL, "Cannot create curve from 0 samples");  // curve.rs:45,61,211,227
            if (ks[k] < 0 || ks[k] > 2 || !vs[k]) return fail(ctx, FW_EINVAL, "bad curve kind / null values");
            if (ks[k] == FW_CURVE_UNEVEN && !ts[k]) return fail(ctx, FW_EINVAL, "uneven curve without times");
            if (ks[k] == FW_CURVE_UNEVEN && ns[k] >= 2) {
                // UnevenCore::new drops non-finite times and duplicates; with fewer than two left it returns
                // Err(NotEnoughSamples) and the reference's `.unwrap()` panics (curve.rs:50,67,217,232)
                const float *tt = (const float *)ts[k];
                int distinct = 0;
                for (int a = 0; a < ns[k]; a++) {
                    if (!std::isfinite(tt[a])) continue;
                    bool dup = false;
                    for (int b = 0; b < a; b++) dup |= std::isfinite(tt[b]) && tt[b] == tt[a];
                    distinct += dup ? 0 : 1;
                }
                if (distinct < 2) return fail(ctx, FW_EINVAL, "uneven curve needs at least 2 distinct finite times");
            }
        }
    }
    for (uint32_t i = 0; i < d->n_emission_settings; i++) {
        const fw_emission_settings &e = d->emission_settings[i];
        if (e.particle_index < 0 || (uint32_t)e.particle_index >= d->n_particle_settings)
            return fail(ctx, FW_EINVAL, "emission_settings.particle_index out of range");  // index panic core.rs:392
        if (e.mode == FW_MODE_NESTED &&
            (e.target_particle_type < 0 || (uint32_t)e.target_particle_type >= d->n_particle_settings))
            return fail(ctx, FW_EINVAL, "target_particle_type out of range");  // index panic core.rs:488
        if (e.mode != FW_MODE_GLOBAL && e.mode != FW_MODE_NESTED) return fail(ctx, FW_EINVAL, "bad emission mode");
        if (e.pacing_kind < 0 || e.pacing_kind > 2) return fail(ctx, FW_EINVAL, "bad pacing kind");
        if (e.shape_kind < 0 || e.shape_kind > 2) return fail(ctx, FW_EINVAL, "bad shape kind");
    }
    return FW_OK;
}

// capacity heuristic: expected live count from the emitters feeding a type, x1.25 + slack
uint32_t derive_capacity(const fw_spawner_desc *d, uint32_t t, const std::vector<uint32_t> &caps, double *expect_live = nullptr) {
    const fw_particle_settings &p = d->particle_settings[t];
    const uint32_t FW_CAP_ROUND = std::max<uint32_t>(FW_TILE, fw_range_young_tile());  // every kernel's tile divides a capacity
    if (p.capacity && !expect_live) return round_up(std::max<uint32_t>(p.capacity, FW_TILE), FW_CAP_ROUND);
    const double life = std::max(0.0, (double)std::max(p.lifetime.min, p.lifetime.max));
    double need = 0;
    for (uint32_t i = 0; i < d->n_emission_settings; i++) {
        const fw_emission_settings &e = d->emission_settings[i];
        if ((uint32_t)e.particle_index != t) continue;
        if (e.mode == FW_MODE_GLOBAL) {
            if (e.pacing_kind == FW_PACING_ONESHOT)
                need += (double)e.oneshot_count;
            else if (e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.duration > 0 && e.count > 0)
                need += (double)e.count / e.duration * (life + 0.05) + 2 * (double)e.count / e.duration / 30.0;
        } else if (e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.count > 0) {
            const fw_particle_settings &pp = d->particle_settings[e.target_particle_type];
            const double plife = std::max(1e-3, (double)std::min(pp.lifetime.min, pp.lifetime.max));
            const double pcap = caps[e.target_particle_type] ? caps[e.target_particle_type] : kMinCapacity;
            need += pcap * (double)e.count * std::max(1.0, life / plife + 0.1);
        }
    }
    if (expect_live) *expect_live = need;  // (what the emitters sustain: SegHost::expect_live)
    if (p.capacity) return round_up(std::max<uint32_t>(p.capacity, FW_TILE), FW_CAP_ROUND);
    need = need * 1.25 + kMinCapacity;
    if (need > 3.0e9) need = 3.0e9;
    return round_up((uint32_t)need, FW_CAP_ROUND);
}

void fill_randvec3(const fw_rand_vec3 &r, float &mn, float &mx, float &spread, float dir[4], float arc[4]) {
    mn = r.magnitude.min, mx = r.magnitude.max, spread = r.spread;
    dir[0] = r.direction[0], dir[1] = r.direction[1], dir[2] = r.direction[2], dir[3] = 0.f;
    fw_q4 q = fw_quat_from_rotation_arc(fw_v3{0.f, 1.f, 0.f}, fw_v3{r.direction[0], r.direction[1], r.direction[2]});
    arc[0] = q.x, arc[1] = q.y, arc[2] = q.z, arc[3] = q.w;
}

uint32_t pad4(uint32_t n) { return (n + 3u) & ~3u; }

// builds the device tables (types, keys, emits, segments) of one spawner
fw_status build_spawner(fw_ctx *ctx, int h, const fw_spawner_desc *d, const std::vector<uint64_t> *carry_serial) {
    SpawnerHost &sp = ctx->spawners[h];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    ctx->tab_force = true;
    const uint32_t nt = d->n_particle_settings, ne = d->n_emission_settings;
    sp.uid = d->uid;
    sp.starts_enabled = d->starts_enabled;
    if (ctx->levels.size() < d->n_emission_settings) ctx->levels.resize(d->n_emission_settings);
    sp.types.assign(nt, TypeHost{});
    sp.em.assign(ne, EmissionHost{});
    sp.seg.assign(nt, kNoSeg);

    fw_status st;
    if ((st = dev_reserve(ctx, ctx->d_types, ctx->n_types + nt, ctx->n_types))) return st;
    if ((st = dev_reserve(ctx, ctx->d_type_coll, ctx->n_types + nt, ctx->n_types))) return st;
    if ((st = dev_reserve(ctx, ctx->d_emits, ctx->n_emits + ne, ctx->n_emits))) return st;
    if ((st = dev_reserve(ctx, ctx->d_emit_serial, ctx->n_emit_slots + ne, ctx->n_emit_slots))) return st;
    if ((st = dev_reserve(ctx, ctx->d_nest_start, ctx->n_emit_slots + ne, ctx->n_emit_slots))) return st;
    if (ctx->nest_ticket_base.size() < ctx->n_emit_slots + ne) ctx->nest_ticket_base.resize(ctx->n_emit_slots + ne, 0u);
    if ((st = dev_reserve(ctx, ctx->d_segs, ctx->segs.size() + nt, ctx->segs.size()))) return st;
    if ((st = ensure_max_seg(ctx, (uint32_t)ctx->segs.size() + nt))) return st;
    ctx->g.type_coll = ctx->d_type_coll.d;
    ctx->g.types = ctx->d_types.d, ctx->g.emits = ctx->d_emits.d, ctx->g.keys = ctx->d_keys.d;
    ctx->g.segs = ctx->d_segs.d, ctx->g.emit_serial = ctx->d_emit_serial.d, ctx->g.nest_start = ctx->d_nest_start.d;

    std::vector<uint32_t> caps(nt, 0);
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t t = 0; t < nt; t++) caps[t] = derive_capacity(d, t, caps);

    for (uint32_t t = 0; t < nt; t++) {
        TypeHost &T = sp.types[t];
        const fw_particle_settings &p = d->particle_settings[t];
        T.ps = p;
        copy_curve(T.scale, p.scale_curve.kind, p.scale_curve.n, p.scale_curve.times, p.scale_curve.values, 1);
        copy_curve(T.base, p.base_color.kind, p.base_color.n, p.base_color.times, p.base_color.rgba, 4);
        copy_curve(T.emis, p.emissive_color.kind, p.emissive_color.n, p.emissive_color.times, p.emissive_color.rgba, 4);
        T.life_lo_safe = (std::isfinite(p.lifetime.min) && std::isfinite(p.lifetime.max))
                             ? std::nextafterf(std::nextafterf(std::min(p.lifetime.min, p.lifetime.max), -INFINITY), -INFINITY)
                             : NAN;
        T.ps.scale_curve.times = T.ps.scale_curve.values = nullptr;  // descriptors are copied, never kept
        T.ps.base_color.times = T.ps.base_color.rgba = nullptr;
        T.ps.emissive_color.times = T.ps.emissive_color.rgba = nullptr;

        FwType dt{};
        memcpy(dt.acc, p.acceleration, sizeof dt.acc);
        memcpy(dt.angacc, p.angular_acceleration, sizeof dt.angacc);
        dt.lin_drag = p.linear_drag, dt.ang_drag = p.angular_drag;
        dt.sc_kind = T.scale.kind, dt.sc_n = T.scale.n;
        dt.bc_kind = T.base.kind, dt.bc_n = T.base.n;
        dt.em_kind = T.emis.kind, dt.em_n = T.emis.n;
        dt.pbr = p.pbr, dt.report_destroyed = p.report_destroyed;
        // (angular_drag must be finite: 0 * inf = NaN, core.rs:648-650 would turn a zero angular velocity into NaN)
        bool nospin = ctx->use_nospin && p.angular_acceleration[0] == 0.f && p.angular_acceleration[1] == 0.f &&
                      p.angular_acceleration[2] == 0.f && std::isfinite(p.angular_drag);
        {  // FW_TYPE_NOSPIN: every entry that feeds the type spawns with zero angular velocity and the same rotation
            const fw_emission_settings *first = nullptr;
            for (uint32_t i = 0; i < ne && nospin; i++) {
                const fw_emission_settings &e = d->emission_settings[i];
                if ((uint32_t)e.particle_index != t) continue;
                nospin = e.initial_angular_velocity.magnitude.min == 0.f && e.initial_angular_velocity.magnitude.max == 0.f &&
                         (!first || memcmp(first->initial_rotation, e.initial_rotation, sizeof e.initial_rotation) == 0);
                if (!first) first = &e;
            }
            nospin = nospin && first != nullptr;
            if (nospin) {
                dt.flags |= FW_TYPE_NOSPIN;
                // (+ 0.0f: a negative zero component becomes +0, which is what from_scaled_axis(0) * rotation makes of it in
                // all but contrived cases; every reader then sees the same bits)
                for (int c = 0; c < 4; c++) dt.const_rot[c] = first->initial_rotation[c] + 0.0f;
            }
        }
        FwTypeColl dc{};
        dc.coll_flags = (p.collision.enabled ? FW_COLL_ENABLED : 0u) |
                        (p.collision.enabled && p.collision.destroy_on_collision ? FW_COLL_DESTROY : 0u);
        dc.coll_mask = p.collision.filter_mask;
        dc.coll_restitution = p.collision.restitution, dc.coll_friction = p.collision.friction;
        std::vector<float> keys;
        auto put = [&](const std::vector<float> &v, uint32_t padded) {
            uint32_t off = (uint32_t)keys.size();
            keys.insert(keys.end(), v.begin(), v.end());
            keys.resize(off + padded, 0.f);
            return off;
        };
        put(T.scale.times, pad4(T.scale.n));
        dt.o_sc_v = put(T.scale.values, pad4(T.scale.n));
        dt.o_bc_t = put(T.base.times, pad4(T.base.n));
        dt.o_bc_v = put(T.base.values, 4 * T.base.n);
        dt.o_em_t = put(T.emis.times, pad4(T.emis.n));
        dt.o_em_v = put(T.emis.values, 4 * T.emis.n);
        if (keys.size() > 0x3FFFFFFFu) return fail(ctx, FW_EINVAL, "curve keys exceed 2^30 floats");
        const bool bigkeys = keys.size() > FW_KEYS_MAX;  // beyond the LDS staging area of the streaming kernels
        uint32_t type_idx;
        if (!ctx->free_types.empty()) {
            type_idx = ctx->free_types.back();
            ctx->free_types.pop_back();
        } else {
            type_idx = ctx->n_types++;
        }
        // a window of the key pool: the first released one that is large enough, or fresh floats at the end
        uint32_t kwin_off = 0, kwin_cap = 0;
        for (size_t fi = 0; fi < ctx->free_keys.size(); fi++)
            if (ctx->free_keys[fi].second >= keys.size()) {
                kwin_off = ctx->free_keys[fi].first, kwin_cap = ctx->free_keys[fi].second;
                ctx->free_keys.erase(ctx->free_keys.begin() + (long)fi);
                break;
            }
        if (!kwin_cap) {
            kwin_cap = std::max<uint32_t>(64u, pad4((uint32_t)keys.size()));
            if ((st = dev_reserve(ctx, ctx->d_keys, ctx->keys_end + kwin_cap, ctx->keys_end))) return st;
            ctx->g.keys = ctx->d_keys.d;
            kwin_off = (uint32_t)ctx->keys_end;
            ctx->keys_end += kwin_cap;
        }
        dt.keys_off = kwin_off;
        dt.keys_len = (uint32_t)keys.size();
        // segment: the slot, its type index and the spawner's reference to it are recorded BEFORE anything that can
        // fail, so that release_spawner_segments undoes a build that stops half-way (nothing leaks, nothing dangles)
        uint32_t si = (uint32_t)ctx->segs.size();
        for (uint32_t k = 0; k < ctx->segs.size(); k++)
            if (!ctx->segs[k].in_use) {
                si = k;
                break;
            }
        if (si == ctx->segs.size()) ctx->segs.push_back(SegHost{});
        SegHost &S = ctx->segs[si];
        S = SegHost{};
        S.in_use = true, S.spawner = h, S.type = (int)t, S.type_idx = type_idx;
        ctx->n_in_use++;
        S.keys_off = dt.keys_off, S.keys_len = dt.keys_len, S.keys_cap = kwin_cap, S.bigkeys = bigkeys;
        S.nospin = nospin, S.n_xplanes = nospin ? 1u : 0u;
        memcpy(S.const_rot, dt.const_rot, sizeof S.const_rot);
        sp.seg[t] = si;
        FW_HIP(ctx, hipMemcpy(ctx->d_keys.d + dt.keys_off, keys.data(), keys.size() * sizeof(float),
                              hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->d_types.d + type_idx, &dt, sizeof dt, hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->d_type_coll.d + type_idx, &dc, sizeof dc, hipMemcpyHostToDevice));
        S.lplane_emission.clear();
        for (uint32_t i = 0; i < ne; i++) {
            const fw_emission_settings &e = d->emission_settings[i];
            if (e.mode == FW_MODE_NESTED && (uint32_t)e.target_particle_type == t)
                S.lplane_emission.push_back((int32_t)i), S.n_lplanes++;
            if (e.mode == FW_MODE_NESTED && (uint32_t)e.particle_index == t) S.nested_fed = true;
        }
        S.auto_capacity = p.capacity == 0;
        S.collides = p.collision.enabled != 0 || bigkeys;
        S.coll_inplace = p.collision.enabled != 0 && p.collision.destroy_on_collision == 0 && !bigkeys;
        S.life_bound = (double)std::max(p.lifetime.min, p.lifetime.max);  // lifetime = lerp(min, max, u), u in [0, 1)
        S.win_ok = !S.nested_fed && std::isfinite(S.life_bound);
        {  // FIFO ring (SegHost::fifo): one lifetime value, no collisions; spawners whose particles emit onto their own
            // type stay on the general path (a parent would see this frame's children as parents)
            // ... and so does a type that receives Nested children AND Global particles (its Global particles would have to be
            // placed behind a live count only the device knows)
            bool any_nested = false, self_nested = false, mixed_feed = false;
            uint32_t n_global_feed = 0;  // Global entries that feed the type: each may add one op to a frame
            for (uint32_t i = 0; i < ne; i++) {
                const fw_emission_settings &e = d->emission_settings[i];
                n_global_feed += (e.mode == FW_MODE_GLOBAL && (uint32_t)e.particle_index == t) ? 1u : 0u;
                any_nested |= e.mode == FW_MODE_NESTED;
                self_nested |= e.mode == FW_MODE_NESTED && e.target_particle_type == e.particle_index;
                mixed_feed |= S.nested_fed && e.mode == FW_MODE_GLOBAL && (uint32_t)e.particle_index == t;
            }
            // (a ring's spawn ops of a frame travel in the kernel arguments of its launch -- FwInlineOps, FW_INLINE_OPS of
            // them: a type fed by more Global entries than that takes the range or the compacting path, whose tiles read op
            // tables from memory)
            S.fifo = ctx->use_fifo && !sp.no_rings && !self_nested && !mixed_feed && (!S.collides || S.coll_inplace) && p.lifetime.min == p.lifetime.max &&
                     n_global_feed <= FW_INLINE_OPS &&
                     std::isfinite(p.lifetime.min) &&
                     caps[t] >= ctx->fifo_min && caps[t] < 0x40000000u &&  // (head + index stays far from 2^32)
                     (!any_nested || ctx->fifo_nested);
            // more such types than one FIFO launch holds (fw_ctx::n_spilled): this one takes a range ring -- if it qualifies for
            // one: the rule below -- and the FIFO rings of the context follow it at the end of the build
            const bool spill = S.fifo && (ctx->n_spilled != 0 || ctx->n_fifo >= kMaxFifoSegs);
            if (spill) S.fifo = false;
            // (fw_ctx::range_few) the capacity of a type that receives Nested children is derived from its parents' CAPACITY -- the
            // host cannot bound their number -- and passes fifo_min for a handful of parents already (examples/textures.rs: 55
            // bullet cases, 110 puffs, 32 768 slots): in a context of few segments such a type stays with its small parent type on
            // range rings (one kind of launch per frame) unless its derived capacity is really large
            bool few_nested = false;  // ... a range ring only because of that: it leaves with the other small rings (drop_few_rings)
            if (S.fifo && S.nested_fed && ctx->use_range && ctx->range_few != 0 && !ctx->few_blocked && ctx->n_in_use <= ctx->range_few && ctx->n_fifo == 0 &&
                caps[t] < 8u * ctx->fifo_min && caps[t] < ctx->range_min * 32u &&
                S.n_lplanes <= 2 && T.life_lo_safe > 0.0f && caps[t] <= FW_RANGE_MAX_CAPACITY)  // (it does qualify for a range ring)
                S.fifo = false, few_nested = true;
            if (S.fifo) {
                ctx->n_fifo++;
                S.win_ok = false;
                S.fifo_mat = any_nested;
                S.fifo_dev = S.nested_fed;
                if (S.fifo_dev) {
                    FW_HIP(ctx, hipHostMalloc((void **)&S.h_report, (size_t)kReportRing * sizeof(unsigned long long), hipHostMallocDefault));
                    memset(S.h_report, 0, (size_t)kReportRing * sizeof(unsigned long long));
                }
                S.fifo_life = 0.0f * (p.lifetime.max - p.lifetime.min) + p.lifetime.min;  // u * (max - min) + min, any u
                S.fifo_wm = (T.base.kind != 0 ? 1 : 0) | (T.emis.kind != 0 ? 2 : 0) | (T.scale.kind != 0 ? 4 : 0);
            }
            // Range ring (SegHost::range): any finite lifetime range -- a single value included, for the types the eight
            // FIFO records of a launch have no room for -- in a spawner without Nested entries; the young part of the
            // list is updated in place, only the part that can lose particles this frame is compacted
            // In a spawner WITH Nested entries (round 4): types other particles' entries emit FROM (range_mat: fw_k_spawn /
            // fw_k_nest address their particles by list index through the size of the old part, which the device keeps --
            // FwGlobals::rold) and types that RECEIVE children (range_dev: the device alone knows their count) qualify too;
            // as for FIFO rings, not a type that emits onto itself, nor one that receives children AND Global particles; at
            // most two last_emitted_age planes (the old tiles carry them in registers).
            S.range = ctx->use_range && !sp.no_rings && !S.fifo && !self_nested && !mixed_feed && S.n_lplanes <= 2 &&
                      (!S.collides || S.coll_inplace) && std::isfinite(p.lifetime.min) &&
                      std::isfinite(p.lifetime.max) && T.life_lo_safe > 0.0f &&
                      (caps[t] >= ctx->range_min || (ctx->range_few != 0 && !ctx->few_blocked && ctx->n_in_use <= ctx->range_few && ctx->n_fifo == 0)) &&
                      caps[t] <= FW_RANGE_MAX_CAPACITY;
            if (S.range) {
                if (caps[t] < ctx->range_min || few_nested) S.few_ring = true, ctx->n_few++;  // (fw_ctx::range_few)
                if (spill) S.spilled = true, ctx->n_spilled++;
                ctx->n_range++;
                S.range_life_lo = T.life_lo_safe;
                ctx->range_life_max = std::max(ctx->range_life_max, S.range_life_lo);
                ctx->r_force = true;
                S.range_mat = S.n_lplanes != 0;
                S.range_dev = S.nested_fed;
                if (S.range_dev) {
                    S.win_ok = false;
                    ctx->range_age_keep = std::max(ctx->range_age_keep, (float)(S.life_bound * 1.01 + 1e-3));
                    FW_HIP(ctx, hipHostMalloc((void **)&S.h_report, (size_t)kReportRing * sizeof(unsigned long long), hipHostMallocDefault));
                    memset(S.h_report, 0, (size_t)kReportRing * sizeof(unsigned long long));
                }
            }
        }
        for (int c = 0; c < 4; c++) {  // the first key is the colour at age 0 (and, for one key, at every age)
            S.fill_bc[c] = T.base.values.empty() ? 0.f : T.base.values[c];
            S.fill_em[c] = T.emis.values.empty() ? 0.f : T.emis.values[c];
        }
        S.colors_dirty = false;
        {
            double expect = 0.0;
            derive_capacity(d, t, caps, &expect);
            S.expect_live = (float)std::min(expect, 3.0e9);
        }
        if ((st = alloc_seg_buffers(ctx, S, caps[t], p.report_destroyed != 0))) return st;
        if (small_eligible(ctx, S)) enter_small(ctx, S);  // (fw_ctx::n_small: the wave-per-type kernel)
        if ((st = upload_seg(ctx, si))) return st;
        const uint32_t zero2[2] = {0, 0};
        for (int r = 0; r < 2; r++) {
            FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.spawned + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.appended + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
            FW_HIP(ctx, hipMemcpy(ctx->g.rold + (size_t)r * ctx->max_seg + si, zero2, 4, hipMemcpyHostToDevice));
        }
        FW_HIP(ctx, hipMemcpy(ctx->g.ndestroyed + si, zero2, 4, hipMemcpyHostToDevice));
        FW_HIP(ctx, hipMemcpy(ctx->g.range_ticket + si, zero2, 4, hipMemcpyHostToDevice));  // (S.ticket_base is 0: a fresh SegHost)
    }

    for (uint32_t i = 0; i < ne; i++) {
        EmissionHost &E = sp.em[i];
        const fw_emission_settings &e = d->emission_settings[i];
        E.es = e;
        E.last_emission = 0.f, E.time_passed_in_cycle = 0.f;  // sync_spawner_data core.rs:350-358
        E.enabled = d->starts_enabled != 0;
        E.emits_on_other_particles = e.mode == FW_MODE_NESTED;
        E.dst_seg = sp.seg[e.particle_index];
        E.life_lo_safe = sp.types[e.particle_index].life_lo_safe;
        E.serial = carry_serial && i < carry_serial->size() ? (*carry_serial)[i] : 0;
        const fw_particle_settings &p = d->particle_settings[e.particle_index];
        FwEmit de{};
        de.shape_kind = e.shape_kind, de.shape_radius = e.shape_radius;
        de.uid = d->uid, de.emission_index = i;
        fw_q4 sa = fw_quat_from_rotation_arc(fw_v3{0.f, 1.f, 0.f},
                                             fw_v3{e.shape_normal[0], e.shape_normal[1], e.shape_normal[2]});
        de.shape_arc[0] = sa.x, de.shape_arc[1] = sa.y, de.shape_arc[2] = sa.z, de.shape_arc[3] = sa.w;
        fill_randvec3(e.initial_velocity, de.v_mag_min, de.v_mag_max, de.v_spread, de.v_dir, de.v_arc);
        fill_randvec3(e.initial_angular_velocity, de.w_mag_min, de.w_mag_max, de.w_spread, de.w_dir, de.w_arc);
        de.inherit = e.inherit_parent_velocity;
        de.type_idx = ctx->segs[sp.seg[e.particle_index]].type_idx;
        memcpy(de.init_rot, e.initial_rotation, sizeof de.init_rot);
        de.radial_min = e.initial_velocity_radial.min, de.radial_max = e.initial_velocity_radial.max;
        de.iscale_min = p.initial_scale.min, de.iscale_max = p.initial_scale.max;
        de.life_min = p.lifetime.min, de.life_max = p.lifetime.max;
        de.n_count = e.count, de.n_start = e.offset_start, de.n_end = e.offset_end;
        de.n_lplane = 0;
        if (e.mode == FW_MODE_NESTED) {
            const SegHost &P = ctx->segs[sp.seg[e.target_particle_type]];
            for (uint32_t k = 0; k < P.n_lplanes; k++)
                if (P.lplane_emission[k] == (int32_t)i) de.n_lplane = k;
        }
        if (!ctx->free_emits.empty()) {
            E.emit_idx = ctx->free_emits.back();
            ctx->free_emits.pop_back();
        } else {
            E.emit_idx = ctx->n_emits++;
        }
        if (!ctx->free_emit_slots.empty()) {
            E.emit_slot = ctx->free_emit_slots.back();
            ctx->free_emit_slots.pop_back();
        } else {
            E.emit_slot = ctx->n_emit_slots++;
        }
        E.assigned = true;
        FW_HIP(ctx, hipMemcpy(ctx->d_emits.d + E.emit_idx, &de, sizeof de, hipMemcpyHostToDevice));
        const unsigned long long s0 = E.serial;
        FW_HIP(ctx, hipMemcpy(ctx->d_emit_serial.d + E.emit_slot, &s0, sizeof s0, hipMemcpyHostToDevice));
        const uint32_t t0 = 0u;
        FW_HIP(ctx, hipMemcpy(ctx->d_nest_start.d + E.emit_slot, &t0, sizeof t0, hipMemcpyHostToDevice));
        ctx->nest_ticket_base[E.emit_slot] = 0u;
    }
    // ring types other particles' entries emit from that need no materialisation (SegHost::virt_parent)
    for (uint32_t t = 0; t < nt; t++) {
        SegHost &S = ctx->segs[sp.seg[t]];
        if (!S.ring() || S.nested_fed || S.n_lplanes > 2) continue;
        if (S.n_lplanes == 0) {  // (no entry emits from it or onto it: nothing in a Nested pass ever looks at its particles)
            S.virt_parent = true;
            continue;
        }
        const fw_particle_settings &p = d->particle_settings[t];
        bool ok = std::min(p.lifetime.min, p.lifetime.max) > 0.0f;
        for (uint32_t k = 0; k < S.n_lplanes && ok; k++) {
            const fw_emission_settings &e = d->emission_settings[S.lplane_emission[k]];
            ok = e.pacing_kind == FW_PACING_COUNT_OVER_DURATION && e.count > 0.0f && e.offset_start >= 0.0f && e.offset_end >= e.offset_start &&
                 std::isfinite(e.count) && std::isfinite(e.offset_end);
        }
        S.virt_parent = ok;
        if (ok && (st = upload_seg(ctx, sp.seg[t]))) return st;
    }
    sp.initialized = true;
    if ((st = ensure_range_arrays(ctx))) return st;
    if ((st = ensure_tile_arrays(ctx))) return st;
    if (ctx->segs.size() > ctx->small_cap) {  // the small-type list (fw_ctx::d_small): room for every segment slot; fw_step never allocates
        if ((st = sync(ctx))) return st;
        const size_t ncap = ctx->segs.size() * 2 + 256;
        if (ctx->d_small) hipFree(ctx->d_small);
        if (ctx->h_small) hipHostFree(ctx->h_small);
        ctx->d_small = nullptr, ctx->h_small = nullptr, ctx->small_cap = 0, ctx->small_pending = false;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_small, ncap * sizeof(uint32_t)));
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_small, ncap * sizeof(uint32_t), hipHostMallocDefault));
        ctx->small_cap = ncap, ctx->small_dirty = true;
    }
    // the context is no longer one of few segments without a FIFO ring: its small range rings continue on the compacting path
    // (fw_ctx::range_few; callers of build_spawner have synchronised the context)
    if (ctx->n_spilled && ctx->n_fifo && (st = spill_fifo_rings(ctx))) return st;  // (fw_ctx::n_spilled)
    if (ctx->n_in_use > ctx->range_few) ctx->few_blocked = true;
    if (ctx->n_few && (ctx->n_fifo != 0 || ctx->n_in_use > ctx->range_few)) {
        // (the hysteresis covers the arrival of a FIFO ring as well: a context in which one comes and goes would otherwise
        // convert its small rings at every arrival -- ADVICE r04)
        ctx->few_blocked = true;
        return drop_few_rings(ctx);
    }
    return FW_OK;
}

fw_status release_spawner_segments(fw_ctx *ctx, SpawnerHost &sp) {
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    ctx->tab_force = true;
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;  // rows in flight describe the old segments
    for (const EmissionHost &e : sp.em) {
        if (!e.assigned) continue;  // a build that failed half-way
        ctx->free_emits.push_back(e.emit_idx);
        ctx->free_emit_slots.push_back(e.emit_slot);
    }
    sp.em.clear();
    for (uint32_t si : sp.seg) {
        if (si == kNoSeg) continue;
        SegHost &S = ctx->segs[si];
        if (!S.in_use) continue;
        ctx->free_types.push_back(S.type_idx);
        if (S.keys_cap) ctx->free_keys.push_back({S.keys_off, S.keys_cap});
        if (S.fifo) ctx->n_fifo--;
        if (S.range) ctx->n_range--, ctx->r_force = true;
        if (S.few_ring) ctx->n_few--;
        if (S.spilled) ctx->n_spilled--;
        if (S.small) ctx->n_small--, ctx->small_dirty = true;
        ctx->n_in_use--;
        if (ctx->n_in_use <= ctx->range_few / 2) ctx->few_blocked = false;
        if (S.h_report) hipHostFree(S.h_report);
        if (S.buf[0]) FW_HIP(ctx, hipFree(S.buf[0]));
        if (S.destroyed) FW_HIP(ctx, hipFree(S.destroyed));
        S = SegHost{};
        const uint32_t zero = 0;
        for (int r = 0; r < 2; r++)
            FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)r * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
    }
    sp.seg.clear();
    return FW_OK;
}


// The update grid covers ceil(bound / FW_TILE) tiles per segment, where `bound` is the host's upper
// bound of the live count.  The table lives on the device and is re-sent only when a segment's
// need leaves the band [need, need * 5/4 + 8], so steady-state frames upload nothing.
fw_status update_tile_table(fw_ctx *ctx) {
    const uint32_t n_seg = (uint32_t)ctx->segs.size();
    // (a context of rings and small types only -- thousands of small emitters: nothing for the compacting launch, whose table is
    // empty already: not a pass over every segment record per frame)
    if (!ctx->tab_force && ctx->total_tiles_dev == 0 && ctx->d_tile_first && ctx->tiles_dev.size() == n_seg &&
        ctx->n_in_use == ctx->n_fifo + ctx->n_range + ctx->n_small) {
        ctx->vt_rounds = 1u;
        return FW_OK;
    }
    bool dirty = ctx->tiles_dev.size() != n_seg || ctx->tab_force;  // descriptors carry per-segment type indices
    ctx->tab_force = false;
    ctx->tiles_dev.resize(n_seg, 0);
    // (the same pass picks the size of the new-particle tiles: one round of FW_VTILE, or two -- at most FW_TILE / 2, Q1/Q2
    // of new particles live in the upper half of the LDS planes -- when only that keeps the whole frame resident)
    uint64_t act1 = 0, act2 = 0;
    for (uint32_t i = 0; i < n_seg; i++) {
        const SegHost &S = ctx->segs[i];
        if (S.in_use && !S.ring() && !S.small) {
            const uint32_t live = seg_live_tiles(S);
            act1 += live + (S.frame_spawn + FW_VTILE - 1) / FW_VTILE;
            act2 += live + (S.frame_spawn + 2 * FW_VTILE - 1) / (2 * FW_VTILE);
        }
        // provision for one-round new-particle tiles whatever vt_rounds says: a lone segment picks its tile size on
        // the device from exact counts and may use the smaller tiles when the host, with looser bounds, would not
        const uint32_t need = seg_tiles(S, 1);
        uint32_t &have = ctx->tiles_dev[i];
        if (!S.in_use || S.ring() || S.small) {
            if (have) have = 0, dirty = true;
            continue;
        }
        const uint32_t cap_tiles =
            (S.capacity + FW_TILE - 1) / FW_TILE + (S.frame_spawn + FW_VTILE - 1) / FW_VTILE + 1;
        // slack: an eighth for large segments; a small segment (thousands of small emitters) gets one spare tile --
        // idle workgroups are cheap one by one, but two per segment doubled such a grid
        if (need > have || have > need + need / 4 + (need >= 16 ? 8u : 2u) || have > cap_tiles) {
            have = std::min(cap_tiles, need + (need >= 16 ? std::max<uint32_t>(2, need / 8) : (need >= 4 ? 1u : 0u)));
            dirty = true;
        }
    }
    // neither size keeps the frame resident: the smaller (more parallel) one
    ctx->vt_rounds = (act2 > kResidentSlots || act1 <= kResidentSlots) ? 1u : 2u;
    if (ctx->trace)
        fprintf(stderr, "[fw] frame %llu tile table dirty=%d n_seg=%u have0=%u ub0=%u\n",
                (unsigned long long)ctx->frame, (int)dirty, n_seg, n_seg ? ctx->tiles_dev[0] : 0u,
                n_seg ? ctx->segs[0].ub : 0u);
    if (!dirty && ctx->d_tile_first) return FW_OK;
    ctx->fc_dirty = true;  // forecast sums are indexed by global tile: a new table invalidates whatever they hold
    if (n_seg + 1 > ctx->tile_first_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = (size_t)(n_seg + 1) * 2 + 64;
        if (ctx->d_tile_first) hipFree(ctx->d_tile_first);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_first, ncap * sizeof(uint32_t)));
        if (ctx->d_tile_keys) hipFree(ctx->d_tile_keys);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_keys, ncap * sizeof(uint2)));
        for (int i = 0; i < kTabRing; i++) {
            if (ctx->h_tab[i]) hipHostFree(ctx->h_tab[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_tab[i], ncap * sizeof(uint32_t), hipHostMallocDefault));
            if (ctx->h_keys[i]) hipHostFree(ctx->h_keys[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_keys[i], ncap * sizeof(uint2), hipHostMallocDefault));
            ctx->tab_pending[i] = false;
        }
        ctx->tile_first_cap = ncap;
    }
    uint32_t total_new = 0;
    for (uint32_t i = 0; i < n_seg; i++) total_new += ctx->tiles_dev[i];
    if (total_new > ctx->tile_desc_cap) {
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = (size_t)total_new * 2 + 256;
        if (ctx->d_tile_desc) hipFree(ctx->d_tile_desc);
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_tile_desc, ncap * sizeof(uint4)));
        for (int i = 0; i < kTabRing; i++) {
            if (ctx->h_desc[i]) hipHostFree(ctx->h_desc[i]);
            FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_desc[i], ncap * sizeof(uint4), hipHostMallocDefault));
            ctx->tab_pending[i] = false;
        }
        ctx->tile_desc_cap = ncap;
    }
    const int slot = (int)(ctx->tab_seq++ % kTabRing);
    if (ctx->tab_pending[slot]) {
        FW_HIP(ctx, hipEventSynchronize(ctx->ev_tab[slot]));
        ctx->tab_pending[slot] = false;
    }
    uint32_t *h = ctx->h_tab[slot];
    uint32_t total = 0;
    for (uint32_t i = 0; i < n_seg; i++) {
        h[i] = total;
        total += ctx->tiles_dev[i];
    }
    h[n_seg] = total;
    ctx->total_tiles_dev = total;
    uint2 *hk = ctx->h_keys[slot];
    for (uint32_t i = 0; i < n_seg; i++) hk[i] = make_uint2(ctx->segs[i].keys_off, ctx->segs[i].keys_len);
    if (n_seg)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_keys, hk, (size_t)n_seg * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    uint4 *hd = ctx->h_desc[slot];
    for (uint32_t i = 0; i < n_seg; i++)
        for (uint32_t t = 0; t < ctx->tiles_dev[i]; t++)
            hd[h[i] + t] = make_uint4(i, h[i], ctx->tiles_dev[i],
                                      ctx->segs[i].type_idx | (ctx->segs[i].nospin ? FW_TYPE_IDX_NOSPIN : 0u));
    FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_first, h, (size_t)(n_seg + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                               ctx->stream));
    if (total)
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_tile_desc, hd, (size_t)total * sizeof(uint4), hipMemcpyHostToDevice,
                                   ctx->stream));
    FW_HIP(ctx, hipEventRecord(ctx->ev_tab[slot], ctx->stream));
    ctx->tab_pending[slot] = true;
    return FW_OK;
}

// The pinned error word (FwGlobals::err_host): a kernel's internal check failed.  No synchronisation, no HIP call: fw_step and
// every reader look here first.  The spawner the segment belongs to (every spawner, when the error names none) is marked.
bool poll_device_error(fw_ctx *ctx) {
    const volatile unsigned long long *w = ctx->h_err;
    const unsigned long long v = w ? *w : 0ull;
    if (!v) return false;
    *ctx->h_err = 0ull;
    const uint32_t check = (uint32_t)(v >> 32) & 0x7FFFFFFFu, seg = (uint32_t)v;
    // (frames enqueued before the host looked here repeat the report: only a spawner that was healthy so far is news)
    bool one = false, news = false;
    if (seg < ctx->segs.size() && ctx->segs[seg].in_use && ctx->segs[seg].spawner >= 0 &&
        (size_t)ctx->segs[seg].spawner < ctx->spawners.size()) {
        SpawnerHost &sp = ctx->spawners[ctx->segs[seg].spawner];
        news = !sp.poisoned;
        sp.poisoned = true;
        one = true;
        ctx->n_poisoned++;
    }
    if (!one)
        for (auto &sp : ctx->spawners) news |= sp.alive && !sp.poisoned, sp.poisoned |= sp.alive, ctx->n_poisoned++;
    if (!news) return false;
    ctx->poison_msg = "internal error: check " + std::to_string(check) + " of an update kernel failed" +
                      (one ? " for segment " + std::to_string(seg) : std::string()) +
                      "; the particle state of the spawner is invalid -- rebuild it with fw_spawner_update_settings (drops its "
                      "particles) or destroy it";
    return true;
}
fw_status poisoned_status(fw_ctx *ctx) { return fail(ctx, FW_EHIP, ctx->poison_msg.empty() ? "spawner poisoned by an earlier internal error" : ctx->poison_msg); }
// A check of the HOST half of fw_step failed after the frame's bookkeeping was committed (clocks, RNG serials, cohorts, ring heads
// have advanced; a launch may be out): nothing can be rolled back, so the spawner the segment belongs to (every spawner, when no
// segment is named) is treated like one a kernel's check failed for -- sticky until rebuilt or destroyed
fw_status poison_segment(fw_ctx *ctx, uint32_t si, const std::string &what) {
    bool one = false;
    if (si < ctx->segs.size() && ctx->segs[si].in_use && ctx->segs[si].spawner >= 0 && (size_t)ctx->segs[si].spawner < ctx->spawners.size()) {
        SpawnerHost &sp = ctx->spawners[ctx->segs[si].spawner];
        if (!sp.poisoned) sp.poisoned = true, ctx->n_poisoned++;
        one = true;
    }
    if (!one)
        for (auto &sp : ctx->spawners)
            if (sp.alive && !sp.poisoned) sp.poisoned = true, ctx->n_poisoned++;
    ctx->poison_msg = "internal error: " + what + (one ? " (segment " + std::to_string(si) + ")" : std::string()) +
                      "; the particle state of the spawner is invalid -- rebuild it with fw_spawner_update_settings (drops its "
                      "particles) or destroy it";
    return poisoned_status(ctx);
}

SpawnerHost *get_spawner(fw_ctx *ctx, fw_spawner h) {
    if (!ctx || h < 0 || (size_t)h >= ctx->spawners.size() || !ctx->spawners[h].alive) {
        if (ctx) ctx->err = "invalid spawner handle";
        return nullptr;
    }
    return &ctx->spawners[h];
}

// consume finished live-count snapshots to tighten the host upper bounds (no sync, no HIP call)
void poll_snapshots(fw_ctx *ctx) {
    for (int k = 0; k < kSnapRing; k++) {
        if (!ctx->snap_pending[k]) continue;
        const volatile unsigned long long *snap = ctx->h_snap + (size_t)k * ctx->max_seg;
        const auto &cum = ctx->snap_cum[k];
        const size_t n = std::min(ctx->segs.size(), cum.size());
        if (!ctx->snap_seen[k]) {
            // the row is complete once every segment's last tile has stored; look at one segment first and give
            // the rest one more step
            size_t probe = n;
            for (size_t i = 0; i < n && probe == n; i++)
                if (ctx->segs[i].in_use) probe = i;
            if (probe == n) {
                ctx->snap_pending[k] = false;
            } else if ((uint32_t)(snap[probe] >> 32) == ctx->snap_epoch[k]) {
                ctx->snap_seen[k] = true;
            }
            continue;
        }
        ctx->snap_pending[k] = false;
        for (size_t i = 0; i < n; i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use) continue;
            const unsigned long long v = snap[i];
            if ((uint32_t)(v >> 32) != ctx->snap_epoch[k]) continue;  // that segment's store has not landed yet
            if (S.fifo && !S.fifo_dev) continue;  // the host's count is exact
            if (S.nested_fed) {
                {   // no host-side bound exists; the count and its growth rate only drive capacity growth
                    const uint32_t ep = ctx->snap_epoch[k], c = (uint32_t)v;
                    if (S.dev_epoch && ep > S.dev_epoch) S.dev_rate = c > S.dev_count ? (float)(c - S.dev_count) / (float)(ep - S.dev_epoch) : 0.f;
                    S.dev_count = c, S.dev_epoch = ep;
                }
                S.snap_count = (uint32_t)v, S.snap_cum = cum[i];
                continue;
            }
            const uint64_t b = (uint64_t)(uint32_t)v + (S.cum_spawn - cum[i]);
            if (b < S.ub) S.ub = (uint32_t)b;
        }
    }
}

fw_status read_counts(fw_ctx *ctx, std::vector<uint32_t> &out) {
    fw_status st = sync(ctx);
    if (st) return st;
    out.assign(ctx->segs.size(), 0);
    if (!out.empty())
        FW_HIP(ctx, hipMemcpy(out.data(), ctx->g.count + (size_t)ctx->parity * ctx->max_seg,
                              out.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if ((st = refresh_rold(ctx))) return st;
    return check_device_errors(ctx);
}

// ParticleSpawnerData::active (core.rs:288-302) with exact device counts
bool spawner_active(const fw_ctx *ctx, const SpawnerHost &sp, const std::vector<uint32_t> &counts) {
    bool any = false;
    for (uint32_t si : sp.seg) any |= counts[si] != 0;
    bool enabled = false;
    for (const EmissionHost &e : sp.em) enabled |= e.emits_on_other_particles ? (e.enabled && any) : e.enabled;
    (void)ctx;
    return enabled;
}

}  // namespace

// =====================================================================================
// C ABI
// =====================================================================================

extern "C" {

int fw_abi_version(void) { return FW_ABI_VERSION; }

const char *fw_last_error(const fw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

uint64_t fw_compute_emission_count(float t, float last, float dur, float start, float end, float per_cycle,
                                   float *next_last) {
    float nl = 0.f;
    const uint64_t n = fw_emission_count(t, last, dur, start, end, per_cycle, &nl);
    if (next_last) *next_last = nl;
    return n;
}

fw_status fw_ctx_create(int device, uint32_t seed, void *stream, fw_ctx **out) {
    if (!out) return FW_EINVAL;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("no HIP device available (") + hipGetErrorString(e) +
                         "); this backend has no CPU fallback";
        return FW_ENODEV;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return FW_EINVAL;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return FW_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName;
        return FW_ENODEV;
    }
    fw_ctx *ctx = new fw_ctx();
    ctx->device = device;
    ctx->seed = seed;
    auto bail = [&](const char *what, hipError_t he) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(he);
        delete ctx;
        return FW_EHIP;
    };
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e);
        ctx->own_stream = true;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate(copy)", e);
    if ((e = hipStreamCreateWithFlags(&ctx->fifo_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate(rings)", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_rtab, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_small, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming)) != hipSuccess)
        return bail("hipEventCreate", e);
    for (int i = 0; i < kParamRing; i++) {
        if ((e = hipEventCreateWithFlags(&ctx->ev_copied[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&ctx->ev_consumed[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    }
    for (int i = 0; i < kTabRing; i++)
        if ((e = hipEventCreateWithFlags(&ctx->ev_tab[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    if ((e = hipHostMalloc((void **)&ctx->h_done, 64, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    *ctx->h_done = 0ull;
    ctx->h_err = ctx->h_done + 4, ctx->h_err[0] = ctx->h_err[1] = 0ull, ctx->g.err_host = ctx->h_err;
    if ((e = hipMalloc((void **)&ctx->g.err, 64)) != hipSuccess) return bail("hipMalloc", e);
    fw_memset_done(ctx->g.err, 0, 64);
    if ((e = hipMalloc((void **)&ctx->g.stats, 64)) != hipSuccess) return bail("hipMalloc", e);
    fw_memset_done(ctx->g.stats, 0, 64);
    if ((e = hipMalloc((void **)&ctx->d_aabb, 256 * 8 * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_total, 64)) != hipSuccess) return bail("hipMalloc", e);
    ctx->g.seed = seed;
    // A/B and debugging switches (firework_hip_debug.h): read only when FW_ENABLE_KNOBS=1 -- a product process does not change
    // behaviour because of a stray environment variable
    const char *knobs_on = getenv("FW_ENABLE_KNOBS");
    auto getenv = [&](const char *name) -> const char * { return (knobs_on && atoi(knobs_on) != 0) ? ::getenv(name) : nullptr; };
    // -- path selectors: every one of them names a path the product takes by itself under some workload; the tests force each
    if (const char *m = getenv("FW_UPDATE_MODE")) ctx->update_mode = !strcmp(m, "split") ? FW_MODE_SPLIT : FW_MODE_FUSED;
    if (const char *m = getenv("FW_FORECAST")) ctx->use_forecast = atoi(m) != 0;
    if (const char *m = getenv("FW_STREAM")) ctx->use_stream = atoi(m) != 0;
    if (const char *m = getenv("FW_STATIC_NEW")) ctx->use_static_new = atoi(m) != 0;  // 0: always count + look back
    if (const char *m = getenv("FW_SPIN_LIMIT")) ctx->spin_limit = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO")) ctx->use_fifo = atoi(m) != 0;
    if (const char *m = getenv("FW_FIFO_MIN")) ctx->fifo_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO_SMALL")) ctx->fifo_small_tiles = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_FIFO_STREAM")) ctx->use_fifo_stream = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE")) ctx->use_range = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_MIN")) ctx->range_min = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_FEW")) ctx->range_few = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_SMALL")) ctx->range_small_tiles = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_RANGE_YOUNG_BIG")) ctx->range_young_big = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_NOSPIN")) ctx->use_nospin = atoi(m) != 0;
    if (const char *m = getenv("FW_NEST_FUSE")) ctx->nest_fuse = atoi(m) != 0;
    if (const char *m = getenv("FW_SMALL")) ctx->use_small = atoi(m) != 0;
    if (const char *m = getenv("FW_SMALL_MAX")) ctx->small_max = (uint32_t)strtoul(m, nullptr, 10);
    if (const char *m = getenv("FW_NT_MB")) ctx->nt_bytes = (uint64_t)atoll(m) << 20;
    if (const char *m = getenv("FW_NT_WO_MB")) ctx->nt_wo_bytes = ctx->nt_wo_bytes_range = (uint64_t)atoll(m) << 20;
#ifdef FW_AB
    // -- the experiment surface: only in the `make ab` build (libfirework_hip_ab.so; the tools load it through FW_LIB_PATH)
    if (const char *m = getenv("FW_DEBUG")) ctx->dbg = (uint32_t)atoi(m);
    if (const char *m = getenv("FW_FIFO_NESTED")) ctx->fifo_nested = atoi(m) != 0;
    if (const char *m = getenv("FW_DERIVED")) ctx->use_derived = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_IDLE_LAST")) ctx->range_idle_last = atoi(m) != 0;
    if (const char *m = getenv("FW_RANGE_SPREAD_NEW")) ctx->range_spread_new = atoi(m) != 0;
    if (const char *m = getenv("FW_AABB")) ctx->track_aabb = atoi(m) != 0;  // same as fw_ctx_track_aabbs(ctx, 1)
    if (const char *m = getenv("FW_OPS_ZEROCOPY")) ctx->ops_zerocopy = atoi(m) != 0;
    if (const char *m = getenv("FW_SNAP_EVERY")) ctx->snap_every = std::max(1, atoi(m));
    ctx->trace = getenv("FW_TRACE") != nullptr;
    if (const char *m = getenv("FW_HOST_PROF")) ctx->host_prof = atoi(m) != 0, ctx->host_prof_skip = atoi(m) > 1 ? (uint64_t)atoi(m) : 0;
#endif
    if (ensure_max_seg(ctx, 1024) != FW_OK) {
        g_create_error = ctx->err;
        delete ctx;
        return FW_EHIP;
    }
    *out = ctx;
    return FW_OK;
}

fw_status fw_ctx_destroy(fw_ctx *ctx) {
    if (!ctx) return FW_EINVAL;
    if (ctx->host_prof && ctx->prof_frames) {
        static const char *names[10] = {"windows+reset", "spawner loop", "tile table", "commit", "args", "op tables", "launch", "post", "", ""};
        fprintf(stderr, "[fw] host half of fw_step over %llu frames (ns per frame):", (unsigned long long)ctx->prof_frames);
        for (int i = 0; i < 8; i++) fprintf(stderr, "  %s %.0f", names[i], ctx->prof_ns[i] / (double)ctx->prof_frames);
        fprintf(stderr, "\n[fw] table uploads: general %llu, range %llu over %llu frames\n", (unsigned long long)ctx->tab_seq,
                (unsigned long long)ctx->r_uploads, (unsigned long long)ctx->frame);
        fprintf(stderr, "[fw] host records: SegHost %zu B, SpawnerHost %zu B (entry at %zu), EmissionHost %zu B, FwOp %zu B\n", sizeof(SegHost),
                sizeof(SpawnerHost), (size_t)((const char *)&ctx->spawners.data()->em.one_ - (const char *)ctx->spawners.data()),
                sizeof(EmissionHost), sizeof(FwOp));
    }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->copy_stream);
    if (ctx->fifo_stream) hipStreamSynchronize(ctx->fifo_stream);
    for (auto &S : ctx->segs) {
        if (S.buf[0]) hipFree(S.buf[0]);
        if (S.destroyed) hipFree(S.destroyed);
        if (S.h_report) hipHostFree(S.h_report);
    }
    void *frees[] = {ctx->d_type_coll.d, ctx->d_segs.d,       ctx->d_types.d,       ctx->d_keys.d,        ctx->d_emits.d,
                     ctx->d_emit_serial.d, ctx->d_nest_start.d, ctx->g.range_ticket, ctx->g.count,        ctx->g.spawned,       ctx->g.appended,     ctx->g.rold,
                     ctx->g.ndestroyed,   ctx->g.tile_cnt,      ctx->g.tile_off,      ctx->g.tile_status,
                     ctx->g.err,          ctx->g.stats,         ctx->g.nest_status,   ctx->g.nest_ticket,
                     ctx->d_aabb,         ctx->d_total,         ctx->d_segids,        ctx->g.dbg_ts,
                     ctx->d_colliders,   ctx->g.tile_box,      ctx->d_stage};
    for (void *p : frees)
        if (p) hipFree(p);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_param[i]) hipHostFree(ctx->h_param[i]);
        if (ctx->d_param[i]) hipFree(ctx->d_param[i]);
        hipEventDestroy(ctx->ev_copied[i]);
        hipEventDestroy(ctx->ev_consumed[i]);
    }
    for (int i = 0; i < kTabRing; i++) {
        hipEventDestroy(ctx->ev_tab[i]);
        if (ctx->h_tab[i]) hipHostFree(ctx->h_tab[i]);
        if (ctx->h_desc[i]) hipHostFree(ctx->h_desc[i]);
    }
    if (ctx->d_tile_first) hipFree(ctx->d_tile_first);
    if (ctx->d_tile_desc) hipFree(ctx->d_tile_desc);
    if (ctx->d_tile_keys) hipFree(ctx->d_tile_keys);
    for (int i = 0; i < kTabRing; i++)
        if (ctx->h_keys[i]) hipHostFree(ctx->h_keys[i]);
    if (ctx->d_fc) hipFree(ctx->d_fc);
    if (ctx->d_fce) hipFree(ctx->d_fce);
    if (ctx->h_snap) hipHostFree(ctx->h_snap);
    if (ctx->h_aabb) hipHostFree(ctx->h_aabb);
    if (ctx->h_done) hipHostFree(ctx->h_done);
    for (hipEvent_t ev : ctx->tev) hipEventDestroy(ev);
    hipStreamDestroy(ctx->copy_stream);
    if (ctx->fifo_stream) hipStreamDestroy(ctx->fifo_stream);
    if (ctx->ev_side) hipEventDestroy(ctx->ev_side);
    if (ctx->ev_rtab) hipEventDestroy(ctx->ev_rtab);
    if (ctx->ev_small) hipEventDestroy(ctx->ev_small);
    if (ctx->d_small) hipFree(ctx->d_small);
    if (ctx->h_small) hipHostFree(ctx->h_small);
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_coll[i]) hipEventDestroy(ctx->ev_coll[i]);
        if (ctx->h_coll[i]) hipHostFree(ctx->h_coll[i]);
    }
    if (ctx->d_rdesc) hipFree(ctx->d_rdesc);
    if (ctx->h_rdesc) hipHostFree(ctx->h_rdesc);
    if (ctx->d_rstatus) hipFree(ctx->d_rstatus);
    if (ctx->d_rts) hipFree(ctx->d_rts);
    for (int i = 0; i < kParamRing; i++) {
        if (ctx->h_rparam[i]) hipHostFree(ctx->h_rparam[i]);
    }
    if (ctx->ev_main) hipEventDestroy(ctx->ev_main);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return FW_OK;
}

void *fw_ctx_stream(const fw_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

fw_status fw_ctx_synchronize(fw_ctx *ctx) {
    if (!ctx) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    return check_device_errors(ctx);
}

fw_status fw_ctx_set_colliders(fw_ctx *ctx, const fw_collider *colliders, uint32_t n) {
    if (!ctx || (n && !colliders)) return fail(ctx, FW_EINVAL, "bad collider set");
    hipSetDevice(ctx->device);
    for (uint32_t i = 0; i < n; i++)
        if (colliders[i].kind < FW_COLLIDER_PLANE || colliders[i].kind > FW_COLLIDER_CONE)
            return fail(ctx, FW_EINVAL, "unknown collider kind");
    // The reference asks the live physics world every frame (core.rs:756-765): a set that changes every frame must not
    // stall the frames in flight.  The new set is staged in pinned memory and copied by the stream itself.
    const int slot = (int)(ctx->coll_seq++ & 1u);
    if (ctx->coll_pending[slot]) {  // the copy of two calls ago: long done unless the caller replaces the set in a tight loop
        FW_HIP(ctx, hipEventSynchronize(ctx->ev_coll[slot]));
        ctx->coll_pending[slot] = false;
    }
    if (!ctx->ev_coll[slot]) FW_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_coll[slot], hipEventDisableTiming));
    if (n > ctx->h_coll_cap[slot]) {
        const size_t ncap = std::max<size_t>(64, (size_t)n * 2);
        if (ctx->h_coll[slot]) FW_HIP(ctx, hipHostFree(ctx->h_coll[slot]));
        ctx->h_coll[slot] = nullptr, ctx->h_coll_cap[slot] = 0;
        FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_coll[slot], ncap * sizeof(FwCollider), hipHostMallocDefault));
        ctx->h_coll_cap[slot] = ncap;
    }
    if (n > ctx->coll_cap) {  // a larger world than ever before: the one case that waits (kernels in flight read the old table)
        fw_status st = sync(ctx);
        if (st) return st;
        const size_t ncap = std::max<size_t>(64, (size_t)n * 2);
        if (ctx->d_colliders) FW_HIP(ctx, hipFree(ctx->d_colliders));
        ctx->d_colliders = nullptr, ctx->coll_cap = 0;
        FW_HIP(ctx, hipMalloc((void **)&ctx->d_colliders, ncap * sizeof(FwCollider)));
        ctx->coll_cap = ncap;
    }
    for (uint32_t i = 0; i < n; i++) {
        const fw_collider &c = colliders[i];
        FwCollider &d = ctx->h_coll[slot][i];
        d = FwCollider{};
        d.kind = c.kind, d.layers = c.layers, d.radius = c.radius;
        // (the sphere around `position` that contains it: a wave skips a collider none of its rays can reach, fw_cast_ray)
        d.bound = c.kind == 1 ? c.radius
                  : c.kind == 2 ? std::sqrt(c.half_extents[0] * c.half_extents[0] + c.half_extents[1] * c.half_extents[1] + c.half_extents[2] * c.half_extents[2]) * 1.0001f
                  : (c.kind == 3 || c.kind == 4) ? std::sqrt(c.radius * c.radius + c.half_extents[1] * c.half_extents[1]) * 1.0001f  // (the rim of a cap / of the base)
                                : INFINITY;
        if (!(d.bound >= 0.0f)) d.bound = INFINITY;  // (NaN / negative extents: never skipped)
        memcpy(d.position, c.position, sizeof c.position);
        memcpy(d.rotation, c.rotation, sizeof c.rotation);
        memcpy(d.normal, c.normal, sizeof c.normal);
        memcpy(d.half_extents, c.half_extents, sizeof c.half_extents);
    }
    if (n) {
        FW_HIP(ctx, hipMemcpyAsync(ctx->d_colliders, ctx->h_coll[slot], n * sizeof(FwCollider), hipMemcpyHostToDevice, ctx->stream));
        FW_HIP(ctx, hipEventRecord(ctx->ev_coll[slot], ctx->stream));
        ctx->coll_pending[slot] = true;
    }
    ctx->n_colliders = n;
    ctx->g.colliders = ctx->d_colliders, ctx->g.n_colliders = n;
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    return FW_OK;
}

fw_status fw_spawner_create(fw_ctx *ctx, const fw_spawner_desc *desc, fw_spawner *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = validate_desc(ctx, desc);
    if (st) return st;
    if ((st = sync(ctx))) return st;
    int h = -1;
    for (size_t i = 0; i < ctx->spawners.size(); i++)
        if (!ctx->spawners[i].alive) {
            h = (int)i;
            break;
        }
    if (h < 0) {
        ctx->spawners.push_back(SpawnerHost{});
        h = (int)ctx->spawners.size() - 1;
    }
    ctx->spawners[h] = SpawnerHost{};
    ctx->spawners[h].alive = true;
    st = build_spawner(ctx, h, desc, nullptr);
    if (st) {
        release_spawner_segments(ctx, ctx->spawners[h]);
        ctx->spawners[h].alive = false;
        return st;
    }
    *out = h;
    return FW_OK;
}

fw_status fw_spawner_update_settings(fw_ctx *ctx, fw_spawner h, const fw_spawner_desc *desc) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = validate_desc(ctx, desc);
    if (st) return st;
    if ((st = sync(ctx))) return st;
    if (sp->poisoned) {
        // everything enqueued on top of the invalid state has finished: what those frames reported again is not news, and the
        // segment slots the report names are about to be reused by the rebuilt spawner
        // (... but the word may by now name ANOTHER spawner, healthy so far -- one word, the last writer wins: mark it before the
        // report is dropped)
        poll_device_error(ctx);
        ctx->h_err[0] = ctx->h_err[1] = 0ull;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
    // RNG streams never replay: carry the serials of surviving emission indices
    std::vector<uint64_t> serials;
    for (const EmissionHost &e : sp->em) {
        unsigned long long s = e.serial;
        if (e.es.mode == FW_MODE_NESTED)
            hipMemcpy(&s, ctx->d_emit_serial.d + e.emit_slot, sizeof s, hipMemcpyDeviceToHost);
        serials.push_back(s);
    }
    if ((st = release_spawner_segments(ctx, *sp))) return st;
    const bool finished_notified = sp->finished_notified;
    SpawnerHost keep = *sp;
    *sp = SpawnerHost{};
    sp->alive = true;
    sp->manual_queued_count = keep.manual_queued_count;
    memcpy(sp->origin_pos, keep.origin_pos, sizeof keep.origin_pos);
    memcpy(sp->origin_rot, keep.origin_rot, sizeof keep.origin_rot);
    memcpy(sp->parent_vel, keep.parent_vel, sizeof keep.parent_vel);
    sp->mod_scale = keep.mod_scale, sp->mod_speed = keep.mod_speed;
    sp->no_rings = keep.no_rings || keep.poisoned;  // (rebuilt after an internal error: compacting path only from now on)
    st = build_spawner(ctx, h, desc, &serials);
    if (st) {
        // the old particle types are gone and the new ones could not be built: the handle dies (every later call on
        // it returns FW_EINVAL) instead of pointing at segments that do not exist
        const std::string why = ctx->err;
        release_spawner_segments(ctx, ctx->spawners[h]);
        ctx->spawners[h] = SpawnerHost{};
        ctx->err = why + " (spawner destroyed: its settings could not be rebuilt)";
        return st;
    }
    ctx->spawners[h].finished_notified = finished_notified;
    ctx->n_poisoned = 0;  // (fw_step looks for invalid spawners only while some exist)
    for (const SpawnerHost &x : ctx->spawners) ctx->n_poisoned += (x.alive && x.poisoned) ? 1u : 0u;
    return st;
}

fw_status fw_spawner_destroy(fw_ctx *ctx, fw_spawner h) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if (sp->poisoned) {  // (as in fw_spawner_update_settings)
        poll_device_error(ctx);
        ctx->h_err[0] = ctx->h_err[1] = 0ull;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.err, &zero, sizeof zero, hipMemcpyHostToDevice));
    }
    if ((st = release_spawner_segments(ctx, *sp))) return st;
    *sp = SpawnerHost{};
    ctx->n_poisoned = 0;
    for (const SpawnerHost &x : ctx->spawners) ctx->n_poisoned += (x.alive && x.poisoned) ? 1u : 0u;
    return FW_OK;
}

fw_status fw_spawner_set_origin(fw_ctx *ctx, fw_spawner h, const float t[3], const float r[4]) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !t || !r) return FW_EINVAL;
    memcpy(sp->origin_pos, t, sizeof sp->origin_pos);
    memcpy(sp->origin_rot, r, sizeof sp->origin_rot);
    return FW_OK;
}

fw_status fw_ctx_set_origins(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *translations,
                             const float *rotations_xyzw) {
    if (!ctx || (n && (!handles || !translations || !rotations_xyzw))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_origins: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) {
        SpawnerHost &sp = ctx->spawners[(size_t)handles[i]];
        memcpy(sp.origin_pos, translations + (size_t)i * 3, sizeof sp.origin_pos);
        memcpy(sp.origin_rot, rotations_xyzw + (size_t)i * 4, sizeof sp.origin_rot);
    }
    return FW_OK;
}

fw_status fw_spawner_set_parent_velocity(fw_ctx *ctx, fw_spawner h, const float v[3]) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !v) return FW_EINVAL;
    memcpy(sp->parent_vel, v, sizeof sp->parent_vel);
    return FW_OK;
}

fw_status fw_spawner_set_modifier(fw_ctx *ctx, fw_spawner h, float scale, float speed) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    sp->mod_scale = scale, sp->mod_speed = speed;
    return FW_OK;
}

fw_status fw_spawner_queue(fw_ctx *ctx, fw_spawner h, uint64_t count) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp) return FW_EINVAL;
    sp->manual_queued_count += count;  // core.rs:284-286
    return FW_OK;
}

fw_status fw_ctx_set_parent_velocities(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *velocities) {
    if (!ctx || (n && (!handles || !velocities))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_parent_velocities: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) memcpy(ctx->spawners[(size_t)handles[i]].parent_vel, velocities + (size_t)i * 3, 3 * sizeof(float));
    return FW_OK;
}

fw_status fw_ctx_set_modifiers(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const float *scales, const float *speeds) {
    if (!ctx || (n && (!handles || !scales || !speeds))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_set_modifiers: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) {
        SpawnerHost &sp = ctx->spawners[(size_t)handles[i]];
        sp.mod_scale = scales[i], sp.mod_speed = speeds[i];
    }
    return FW_OK;
}

fw_status fw_ctx_queue(fw_ctx *ctx, uint32_t n, const fw_spawner *handles, const uint64_t *counts) {
    if (!ctx || (n && (!handles || !counts))) return FW_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (!get_spawner(ctx, handles[i])) return fail(ctx, FW_EINVAL, "fw_ctx_queue: invalid spawner handle");
    for (uint32_t i = 0; i < n; i++) ctx->spawners[(size_t)handles[i]].manual_queued_count += counts[i];  // core.rs:284-286
    return FW_OK;
}

// ---- the frame ---------------------------------------------------------------------------
fw_status fw_step(fw_ctx *ctx, float dt) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    auto prof_t = std::chrono::steady_clock::now();
    auto prof = [&](int i) {
        if (!ctx->host_prof) return;
        const auto now = std::chrono::steady_clock::now();
        if (ctx->frame < ctx->host_prof_skip) {
            prof_t = now;
            return;
        }
        ctx->prof_ns[i] += std::chrono::duration<double, std::nano>(now - prof_t).count();
        prof_t = now;
    };
    // a spawner whose particle state an internal error invalidated (SpawnerHost::poisoned): no further frame is enqueued on
    // top of it -- for anybody: the frame is all-or-nothing -- until it has been rebuilt or destroyed
    poll_device_error(ctx);
    if (ctx->n_poisoned)  // (counted, not looked for: with thousands of spawners a scan of their records would be a cost of every frame)
        for (const SpawnerHost &sp : ctx->spawners)
            if (sp.alive && sp.poisoned) return poisoned_status(ctx);
    poll_snapshots(ctx);
    if (ctx->derive_ready_any) {  // types whose caller-written particles have all been through an update (see the end of fw_step)
        ctx->derive_ready_any = false;
        for (uint32_t i = 0; i < ctx->segs.size(); i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use || !S.derive_ready) continue;
            S.derive_ready = false;
            if (S.inst == nullptr || S.colors_dirty || S.collides || !ctx->use_derived) continue;  // (detached / rewritten since)
            const fw_status dst = set_derived(ctx, i, true);
            if (dst) return dst;
        }
    }

    // per-frame scratch lives in the context: with thousands of emitters the allocations were a visible part of the
    // host's ~60 ns per emitter
    auto &levels = ctx->levels;
    for (auto &L : levels) L.g.clear(), L.n.clear();
    ctx->fifo_ops.clear(), ctx->fifo_mat_ops.clear(), ctx->range_mat_ops.clear();
    ctx->seg_kind_changed = false;
    if (!std::isfinite(dt))  // 0 * inf = NaN: an angular velocity of zero does not stay zero (core.rs:648-650)
        for (uint32_t si = 0; si < ctx->segs.size(); si++)
            if (ctx->segs[si].in_use && ctx->segs[si].nospin) {
                fw_status nst = leave_nospin(ctx, si);
                if (nst) return nst;
            }
    if (ctx->n_fifo) {
        // the FIFO order rests on ages that never decrease: a negative or non-finite dt ends the mode (as does a dt so
        // small that the cohort list grows without bound)
        const bool dt_ok = dt >= 0.0f && std::isfinite(dt);
        for (uint32_t si = 0; si < ctx->segs.size(); si++) {
            SegHost &S = ctx->segs[si];
            // (a type that receives Nested children: a step as long as its lifetime would destroy children whose number only
            // this frame's Nested pass will know)
            if (!S.in_use || !S.fifo || (dt_ok && S.coh.size() < kMaxCohorts && !(S.fifo_dev && dt >= S.fifo_life))) continue;
            fw_status cst = fifo_to_general(ctx, si);
            if (cst) return cst;
        }
    }
    ctx->range_ops.clear();
    if (ctx->n_range) {
        // The in-place young part of a range ring rests on ages that never decrease and on nobody -- a particle spawned this
        // frame included -- dying before its age reaches lifetime.min: a negative or non-finite dt, a step as long as the
        // shortest lifetime, or a cohort list that grows without bound (a dt thousands of times smaller than the
        // lifetimes) end the mode; the type continues on the compacting path.
        const bool flood = ctx->birth_age.size() > (1u << 22);
        for (uint32_t si = 0; si < ctx->segs.size(); si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.range) continue;
            if (!flood && dt >= 0.0f && dt < S.range_life_lo && S.ycoh.size() < kMaxCohorts && S.dcoh.size() < kMaxCohorts) continue;
            fw_status cst = fifo_to_general(ctx, si);
            if (cst) return cst;
        }
    }
    if (!ctx->n_range) ctx->birth_age.clear();
    // (frame_spawn is reset in the lifetime-window pass below: one pass over the segments instead of two)
    bool new_static = std::isfinite(dt);  // cleared by any Global op whose particles might not survive this step

    // lifetime windows: drop the spawns that must have expired by now and tighten the bounds with what is left.
    // (ages are fp32 sums of the same dt values on the device; the margin covers the rounding difference)
    if (!(dt >= 0.0f) || !std::isfinite(dt))
        for (auto &S : ctx->segs) S.win_ok = false;  // ages would not grow monotonically
    // (the same pass notes what the rest of the frame asks of every segment: which Nested-fed ones must grow, whether a
    // particle type collides, whether a compacting segment has an instance buffer attached)
    bool any_coll = false, any_inst_general = false;
    struct {
        uint64_t fifo_parts = 0, range_parts = 0;
        bool fifo_dev = false, fifo_coll = false, fifo_inst = false, range_dev = false, range_coll = false, range_inst = false;
    } ring_stats;
    ctx->grow_scratch.clear();
    for (size_t si = 0, ns = ctx->segs.size(); si < ns; si++) {
        SegHost &S = ctx->segs[si];
        if (si + 8 < ns) {  // the oldest window entry of a later segment: a heap line of its own, fetched ahead of time
            const SegHost &N = ctx->segs[si + 8];
            if (N.win.n) __builtin_prefetch(&N.win.v[N.win.head]);
        }
        S.frame_spawn = 0;
        if (!S.in_use) continue;
        S.dead_at_end = false;  // (set again below for the segments this frame updates as range rings)
        any_coll |= S.collides && !S.ring();  // (a colliding type in a ring is updated by its ring kernel)
        // (what decides the tile size of the ring launches -- fifo_small / range_small below -- gathered while the record is hot)
        if (S.fifo) ring_stats.fifo_parts += S.ub, ring_stats.fifo_dev |= S.fifo_dev, ring_stats.fifo_coll |= S.collides, ring_stats.fifo_inst |= S.inst != nullptr;
        if (S.range) ring_stats.range_parts += S.ub, ring_stats.range_dev |= S.range_dev, ring_stats.range_coll |= S.collides, ring_stats.range_inst |= S.inst != nullptr;
        any_inst_general |= !S.ring() && S.inst != nullptr;
        if (S.nested_fed && nested_fed_wants_growth(S)) ctx->grow_scratch.push_back((uint32_t)si);
        if (!S.win_ok) continue;
        // an age is an fp32 running sum of the dt values: up to half an ulp of the age per step taken, i.e. a relative
        // error below steps * 6e-8; the horizon carries that (with a factor 4) on top of a fixed 1e-3
        while (!S.win.empty() &&
               ctx->sim_time - S.win.front().t >=
                   S.life_bound * (1.0 + 1e-3 + 2.4e-7 * (double)(ctx->frame - S.win.front().frame)) + 1e-6) {
            S.win_sum -= S.win.front().n;
            S.win.pop_front();
        }
        if (S.win_sum < S.ub) S.ub = (uint32_t)S.win_sum;
    }

    // Nested-fed segments whose count (or its growth) nears their derived capacity: nested_fed_wants_growth.  By half again,
    // not by doubling.  Caller-given capacities are left alone.
    for (uint32_t si : ctx->grow_scratch) {
        SegHost &S = ctx->segs[si];
        // (growing one type's children may already have grown a later entry of the list)
        if (!S.in_use || !nested_fed_wants_growth(S)) continue;
        const uint32_t want = (uint32_t)std::min<uint64_t>((uint64_t)S.capacity * 3 / 2 + (uint64_t)(64.0f * S.dev_rate), 0x70000000ull);
        S.dev_count = 0, S.dev_rate = 0.f, S.dev_epoch = 0;
        fw_status gst = grow_segment(ctx, si, want, false);
        if (!gst) gst = grow_nested_children(ctx, ctx->spawners[S.spawner], (uint32_t)S.type);
        if (gst) return gst;
    }

    // Everything the host half changes before the frame is known to be enqueueable goes through this log; `rollback`
    // restores it (the per-segment live-count bounds only ever get looser, which is harmless).
    ctx->undo_em.clear(), ctx->undo_sp.clear();
    auto rollback = [&](fw_status why) {
        for (const auto &u : ctx->undo_em) {
            EmissionHost &E = ctx->spawners[u.spawner].em[u.entry];
            E.last_emission = u.last_emission, E.time_passed_in_cycle = u.time_passed_in_cycle;
            E.enabled = u.enabled, E.serial = u.serial;
        }
        for (const auto &u : ctx->undo_sp) ctx->spawners[u.spawner].manual_queued_count = u.manual_queued_count;
        // the spawn totals and lifetime windows the ops of this frame were entered into (note_spawned below)
        auto forget = [&](const FwOp &op) {
            SegHost &S = ctx->segs[op.seg];
            S.cum_spawn -= op.n;
            if (S.fifo || !S.win_ok || S.win.empty() || S.win.back().t != ctx->sim_time) return;
            S.win_sum -= op.n;
            if ((S.win.back().n -= op.n) == 0) S.win.pop_back();
        };
        for (auto &L : ctx->levels)
            for (const FwOp &op : L.g) forget(op);
        for (const FwOp &op : ctx->fifo_ops) forget(op);
        for (const FwOp &op : ctx->range_ops) forget(op);
        for (const auto &io : ctx->fifo_mat_ops) forget(io.second);
        for (const auto &io : ctx->range_mat_ops) forget(io.second);
        for (auto &S : ctx->segs) S.ub -= std::min(S.ub, S.frame_spawn), S.frame_spawn = 0;
        return why;
    };
    // a Global op enters its segment's spawn total and lifetime window as it is made (the segment record is in the cache
    // then; `rollback` takes it out again)
    auto note_spawned = [&](SegHost &S, uint64_t n) {
        S.cum_spawn += n;
        if (S.fifo || !S.win_ok) return;
        if (!S.win.empty() && S.win.back().t == ctx->sim_time) {
            S.win.back().n += n;
        } else {
            if (S.win.size() >= 8192) {  // very long lifetimes: fold the two oldest entries into the newer
                const uint64_t m = S.win.front().n;
                S.win.pop_front();
                S.win.front().n += m;
            }
            S.win.push_back(SegHost::Spawned{ctx->sim_time, n, ctx->frame});
        }
        S.win_sum += n;
    };

    prof(0);
    // spawn_particles, host half (core.rs:377-428): emission clocks and Global counts
    for (size_t h = 0; h < ctx->spawners.size(); h++) {
        SpawnerHost &sp = ctx->spawners[h];
        if (h + 8 < ctx->spawners.size()) {  // the entries of a later spawner live on the heap: fetch them ahead of time
            const SpawnerHost &nx = ctx->spawners[h + 8];
            if (nx.em.size() > 1) {  // (a single entry is part of the spawner's record: the array is streamed as it is)
                __builtin_prefetch(nx.em.data());
                __builtin_prefetch((const char *)nx.em.data() + 128);
            }
        }
        if (!sp.alive) continue;
        // `if data.active()` (core.rs:378): an entry that emits on other particles contributes only when
        // some particle exists; with no particle at all the Nested arm below is a no-op anyway, so the
        // host-side gate "any entry enabled" gives the same state transitions without a device round trip.
        bool any_enabled = false;
        for (const EmissionHost &e : sp.em) any_enabled |= e.enabled;
        if (!any_enabled) continue;
        for (size_t i = 0; i < sp.em.size(); i++) {
            EmissionHost &E = sp.em[i];
            if (!E.enabled) continue;
            const fw_emission_settings &es = E.es;
            const uint32_t dst = E.dst_seg;
            if (es.mode == FW_MODE_GLOBAL) {
                uint64_t n = 0;
                ctx->undo_em.push_back(fw_ctx::EmUndo{(uint32_t)h, (uint32_t)i, E.last_emission, E.time_passed_in_cycle,
                                                      E.enabled, E.serial});
                if (es.pacing_kind == FW_PACING_ONESHOT) {
                    E.enabled = false;  // core.rs:397-400
                    n = es.oneshot_count;
                } else if (es.pacing_kind == FW_PACING_ONDEMAND) {
                    n = sp.manual_queued_count;  // core.rs:401-405
                    if (n) ctx->undo_sp.push_back(fw_ctx::SpUndo{(uint32_t)h, n});
                    sp.manual_queued_count = 0;
                } else {
                    E.time_passed_in_cycle = fw_rem_euclid(E.time_passed_in_cycle + dt, es.duration);  // core.rs:412-414
                    float next = 0.f;
                    n = fw_emission_count(E.time_passed_in_cycle, E.last_emission, es.duration, es.offset_start,
                                          es.offset_end, es.count, &next);
                    E.last_emission = next;
                }
                if (!n) continue;
                if (n > kMaxSpawnPerOp)
                    return rollback(fail(ctx, FW_ECAPACITY, "emission count exceeds 2^30 particles in one frame"));
                SegHost &S = ctx->segs[dst];
                if (S.nested_fed && S.auto_capacity && S.capacity < 0x70000000u) {
                    // A type that receives Nested children too: its children are counted on the device, but its Global
                    // particles are counted right here.  Count of the latest snapshot row + every Global particle since
                    // (at most those the lifetime window still holds) + this op: past half the capacity, the segment grows
                    // NOW, however fast the burst -- the snapshot rule above only follows what the device has seen.
                    uint64_t since = S.cum_spawn - S.snap_cum;  // (includes this frame's earlier ops: note_spawned)
                    if (S.win_ok) since = std::min<uint64_t>(since, S.win_sum);
                    const uint64_t est = (uint64_t)S.snap_count + since + n;
                    if (est > S.capacity / 2) {
                        fw_status st = grow_segment(ctx, dst, (uint32_t)std::min<uint64_t>(est * 2, 0x70000000ull));
                        if (!st) st = grow_nested_children(ctx, sp, (uint32_t)es.particle_index);
                        // (the growth refreshed every bound from the device's exact counts, without this frame's appends)
                        for (auto &X : ctx->segs) X.ub += X.frame_spawn;
                        if (st) return rollback(st);
                        S.snap_count = S.ub - std::min(S.ub, S.frame_spawn), S.snap_cum = S.cum_spawn - S.frame_spawn;
                        S.dev_count = 0;
                    }
                }
                if (!S.nested_fed && (uint64_t)S.ub + n > S.capacity) {
                    fw_status st = refresh_counts_exact(ctx);
                    if (st) return rollback(st);
                    // the refresh dropped this frame's earlier appends from ub: add them back
                    for (auto &X : ctx->segs) X.ub += X.frame_spawn;
                    if ((uint64_t)S.ub + n > S.capacity) {
                        if ((uint64_t)S.ub + n > 0xF0000000ull)
                            return rollback(fail(ctx, FW_ECAPACITY, "particle type too large"));
                        // grow_segment copies `ub - frame_spawn` settled particles; appended ones are not on the
                        // device yet (spawn kernels of this frame have not been enqueued)
                        const uint32_t fs = S.frame_spawn;
                        S.ub -= fs;
                        const uint32_t settled = S.ub;
                        st = grow_segment(ctx, dst, (uint32_t)(settled + fs + n));
                        if (!st) st = grow_nested_children(ctx, sp, (uint32_t)es.particle_index);
                        for (auto &X : ctx->segs) X.ub += X.frame_spawn;
                        if (st) return rollback(st);
                    }
                }
                // does every particle of this op outlive the step?  (TypeHost::life_lo_safe; false for NaN)
                if (!S.ring() && !(dt < E.life_lo_safe)) new_static = false;
                FwOp op{};
                op.seg = dst, op.emit = E.emit_idx, op.n = (uint32_t)n;
                op.rel_base = S.frame_spawn;
                op.serial_base = E.serial;
                memcpy(op.origin_pos, sp.origin_pos, sizeof sp.origin_pos);
                memcpy(op.origin_rot, sp.origin_rot, sizeof sp.origin_rot);
                memcpy(op.parent_vel, sp.parent_vel, sizeof sp.parent_vel);
                op.speed = sp.mod_speed, op.scale = sp.mod_scale;
                if (S.fifo && S.fifo_mat && !S.virt_parent)
                    ctx->fifo_mat_ops.push_back({(uint32_t)i, op});  // routed below, once the frame's Nested ops are known
                else if (S.fifo)
                    ctx->fifo_ops.push_back(op);  // spawned inside fw_k_update_fifo, whatever else the frame holds
                else if (S.range && S.range_mat && !S.virt_parent)
                    ctx->range_mat_ops.push_back({(uint32_t)i, op});  // routed below, once the frame's Nested ops are known
                else if (S.range)
                    ctx->range_ops.push_back(op);  // spawned inside fw_k_update_range
                else
                    levels[i].g.push_back(op);
                E.serial += n;
                S.frame_spawn += (uint32_t)n;
                S.ub = (uint32_t)std::min<uint64_t>((uint64_t)S.ub + n, 0xFFFFFFFFull);
                if (S.small && S.ub > ctx->small_max) leave_small(ctx, S);  // (no longer a few hundred particles: a compacting segment from this frame on)
                note_spawned(S, n);
            } else {
                if (es.pacing_kind != FW_PACING_COUNT_OVER_DURATION) continue;  // warn_once + continue core.rs:474-485
                const SegHost &P = ctx->segs[sp.seg[es.target_particle_type]];
                // children are born with age 0 like any new particle: do they all outlive this step?
                if (!(dt < E.life_lo_safe)) new_static = false;
                FwNestOp op{};
                op.parent_seg = sp.seg[es.target_particle_type], op.child_seg = dst;
                op.emit = E.emit_idx, op.emit_slot = E.emit_slot;
                {  // parent tiles of the launch: from the host's bound of the parent count (exact counts live on the device)
                    const uint32_t par_ub = P.nested_fed ? P.capacity : std::min(P.ub, P.capacity);
                    op.n_tiles = (par_ub + FW_NEST_TILE - 1) / FW_NEST_TILE + 1;
                }
                op.speed = sp.mod_speed, op.scale = sp.mod_scale;
                // (parent_buf / parent_cap are filled in when the launch is built: a later entry of this frame may
                // still grow the parent segment)
                for (uint32_t k = 0; k < P.n_lplanes; k++)
                    if (P.lplane_emission[k] == (int32_t)i) op.parent_lplane = k;
                op.n_count = es.count, op.n_start = es.offset_start, op.n_end = es.offset_end;
                levels[i].n.push_back(op);
            }
        }
    }

    // Rings of spawners with Nested entries: a frame that runs a Nested pass (anywhere in the context: the launches are
    // per emission level) materialises their new particles with fw_k_spawn, at their level, so that the per-parent pass
    // finds them in memory (core.rs:488); any other frame spawns them inside fw_k_update_fifo like every other ring's.
    bool nested_frame = false;
    {
        for (auto &L : levels) nested_frame |= !L.n.empty();
        for (auto &io : ctx->fifo_mat_ops) {
            if (nested_frame) {
                io.second.head = ctx->segs[io.second.seg].head;
                levels[io.first].g.push_back(io.second);
            } else {
                ctx->fifo_ops.push_back(io.second);
            }
        }
        // (range rings other particles' entries emit from: the same choice; materialised at the tail of the ring -- the slot of
        // the list's first particle follows from the old part's size, which the device keeps: FwOp::range_ring)
        for (auto &io : ctx->range_mat_ops) {
            if (nested_frame) {
                io.second.head = ctx->segs[io.second.seg].young_lo, io.second.range_ring = 1u;
                levels[io.first].g.push_back(io.second);
            } else {
                ctx->range_ops.push_back(io.second);
            }
        }
        // every routed op now sits in exactly one list `rollback` walks (levels[].g, fifo_ops or range_ops): forgetting this one
        // too would take its particles out of cum_spawn twice
        ctx->fifo_mat_ops.clear(), ctx->range_mat_ops.clear();
    }
    // A ring that had to grow past its mode's slot limit inside the loop above (realloc_segment) continues as a compacting
    // segment from this very frame: the ops already queued for it as a ring's go where a compacting segment's ops go -- its
    // emission level -- or they would never be spawned while cum_spawn, ub and the lifetime window count them.
    if (ctx->seg_kind_changed) {
        auto reroute = [&](std::vector<FwOp> &list, bool was_fifo) {
            size_t w = 0;
            for (size_t r = 0; r < list.size(); r++) {
                const FwOp &op = list[r];
                const SegHost &S = ctx->segs[op.seg];
                if (was_fifo ? S.fifo : S.range) {
                    list[w++] = op;
                    continue;
                }
                size_t lvl = 0;
                const SpawnerHost &osp = ctx->spawners[S.spawner];
                for (size_t i = 0; i < osp.em.size(); i++)
                    if (osp.em[i].emit_idx == op.emit) lvl = i;
                levels[lvl].g.push_back(op);
                if (!(dt < osp.em[lvl].life_lo_safe)) new_static = false;  // (the test a compacting segment's op gets above)
            }
            list.resize(w);
        };
        reroute(ctx->fifo_ops, true);
        reroute(ctx->range_ops, false);
        for (auto &L : levels)
            for (FwOp &op : L.g)
                if (op.range_ring && !ctx->segs[op.seg].range) op.range_ring = 0u, op.head = 0u;
    }
    // ---- Nested entries run INSIDE the FIFO launch (fw_kernels.h: FwFifoNest; core.rs:471-546).  Every Nested entry of the frame
    // must qualify -- parents in a FIFO ring the host knows the count of (Global-fed), spawned inside the update kernel
    // (virt_parent), no other entry emitting from them; children received by a FIFO ring nothing else feeds and nothing emits from;
    // both in the one FIFO launch of the context, parents first; no instance buffers, no colliders -- and no ring may wait for
    // fw_k_spawn (the ops left at the levels then belong to compacting segments, which no Nested entry of the frame touches: the
    // general launch spawns them itself).  Otherwise the frame runs the separate passes (fw_k_spawn / fw_k_nest), as before.
    FwNestOp fuse_plan[FW_FIFO_NEST_MAX];
    uint32_t n_fuse = 0;
    bool fuse = false;
    if (nested_frame && ctx->nest_fuse && !any_coll && !ctx->seg_kind_changed && ctx->update_mode == FW_MODE_FUSED && ctx->n_fifo != 0 &&
        ctx->n_fifo <= FW_FIFO_PER_LAUNCH && ctx->fifo_ops.size() <= FW_INLINE_OPS && !ring_stats.fifo_coll && !ring_stats.fifo_inst &&
        dt >= 0.0f && std::isfinite(dt)) {
        fuse = true;
        for (auto &L : levels) {
            for (const FwOp &op : L.g) fuse &= !ctx->segs[op.seg].ring();
            for (const FwNestOp &op : L.n) {
                if (!fuse || n_fuse == FW_FIFO_NEST_MAX) {
                    fuse = false;
                    break;
                }
                const SegHost &P = ctx->segs[op.parent_seg], &Cs = ctx->segs[op.child_seg];
                const uint32_t p_in = P.ub - std::min(P.ub, P.frame_spawn);  // (a Global-fed FIFO ring: `ub` is exact)
                fuse = P.fifo && !P.fifo_dev && P.virt_parent && P.n_lplanes == 1 && Cs.fifo && Cs.fifo_dev && Cs.n_lplanes == 0 &&
                       op.parent_seg < op.child_seg && Cs.capacity <= FW_RANGE_MAX_CAPACITY &&
                       // the ring must not wrap into its head tile: the tiles' ranks are then the list order
                       (uint64_t)(P.head % FW_TILE) + p_in <= P.capacity;
                for (uint32_t k = 0; k < n_fuse; k++)
                    fuse &= fuse_plan[k].parent_seg != op.parent_seg && fuse_plan[k].parent_seg != op.child_seg &&
                            fuse_plan[k].child_seg != op.parent_seg && fuse_plan[k].child_seg != op.child_seg;
                if (fuse) fuse_plan[n_fuse++] = op;
            }
            if (!fuse) break;
        }
        fuse = fuse && n_fuse != 0;
    }
    if (nested_frame) (fuse ? ctx->fused_nest_frames : ctx->nest_pass_frames)++;
    // ---- segment -> tile table (device resident, re-uploaded only when a bound moves out of its band)
    prof(1);
    const uint32_t n_seg = (uint32_t)ctx->segs.size();
    fw_status st = update_tile_table(ctx);
    prof(2);
    if (st) return rollback(st);
    const uint32_t total_tiles = ctx->total_tiles_dev;

    size_t n_g = 0, n_n = 0;
    for (auto &L : levels) n_g += L.g.size(), n_n += L.n.size();
    // last thing that can fail before the frame is enqueued: room for the op tables of either form
    if ((st = ensure_param_ring(ctx, (size_t)n_seg * 16 + n_g * sizeof(FwOp) +
                                         n_n * sizeof(FwNestOp) + 16)))
        return rollback(st);
    // ---- the frame will run
    prof(3);
    const uint32_t p = ctx->parity;
    // Particle types with collision settings (core.rs:607-624) run the count / scan / update-with-collisions launches
    // (everything materialised first, like FW_UPDATE_MODE=split); the streaming kernels never see a collider.
    if (ctx->seg_kind_changed)  // (a colliding ring that left its mode inside the spawner loop is a compacting segment from this frame on)
        for (const SegHost &S : ctx->segs) any_coll |= S.in_use && S.collides && !S.ring();
    const int frame_mode = any_coll ? FW_MODE_SPLIT_COLL : ctx->update_mode;
    const bool legacy = (n_n != 0 && !fuse) || frame_mode != FW_MODE_FUSED;  // (fuse: the Nested entries run inside the FIFO launch)

    FwUpdateArgs a{};
    a.seg_tile_first = ctx->d_tile_first;
    a.tile_desc = ctx->d_tile_desc;
    a.n_seg = n_seg;
    a.total_tiles = total_tiles;
    a.parity = p;
    a.epoch = (uint32_t)((ctx->frame + 1) & 0x3FFFFFFFu);
    if (!a.epoch) a.epoch = 1;
    a.dt = dt;
    a.spin_limit = ctx->spin_limit;
    a.dbg = ctx->dbg;
    a.vt_rounds = ctx->vt_rounds;
    a.resident_slots = (uint32_t)kResidentSlots;
    a.seg0_type = n_seg ? ctx->segs[0].type_idx | (ctx->segs[0].nospin ? FW_TYPE_IDX_NOSPIN : 0u) : 0;
    a.seg0_keys_off = n_seg ? ctx->segs[0].keys_off : 0;
    a.seg0_keys_len = n_seg ? ctx->segs[0].keys_len : 0;
    a.tile_keys = ctx->d_tile_keys;
    if (n_seg == 1 && ctx->segs[0].in_use && !ctx->segs[0].ring()) {
        const SegHost &S0 = ctx->segs[0];
        a.seg0_ib = S0.buf[p], a.seg0_ob = S0.buf[p ^ 1u];
        a.seg0_destroyed = S0.destroyed, a.seg0_inst = S0.inst;
        a.seg0_capacity = S0.capacity, a.seg0_n_lplanes = S0.n_lplanes, a.seg0_inst_cap = S0.inst_cap;
    }
    if (ctx->live_ring) {
        a.live_out = ctx->live_ring + (ctx->live_ring_frames % ctx->live_ring_n);
        a.live_next = ctx->live_ring + ((ctx->live_ring_frames + 1) % ctx->live_ring_n);
        ctx->live_ring_frames++;
    }
    a.new_static = (new_static && ctx->use_static_new) ? 1u : 0u;
    a.boxes = (ctx->track_aabb && frame_mode == FW_MODE_FUSED) ? 1u : 0u;
    a.force_colors = ctx->colors_dirty ? 1u : 0u;
    // (a ring that had to grow past the ring limit during this frame's spawner loop continues as a compacting segment:
    // look again -- never in a steady-state frame)
    if (ctx->seg_kind_changed)
        for (const SegHost &S : ctx->segs) any_inst_general |= S.in_use && !S.ring() && S.inst != nullptr;
    a.any_inst = any_inst_general ? 1u : 0u;
    a.use_stream = ctx->use_stream ? 1u : 0u;
    uint32_t dt_bits;
    memcpy(&dt_bits, &dt, 4);
    // Frames with Nested entries materialise their new particles before the update; the streaming kernel takes them as
    // loaded new-particle tiles with static slots, which needs every one of them to survive the step (new_static).
    const bool split = frame_mode != FW_MODE_FUSED;
    const bool fc_frame = !split && ctx->use_forecast && ctx->d_fc != nullptr;
    if (fc_frame) {
        for (uint32_t i = 0; i < n_seg; i++) a.fc_sums |= ctx->tiles_dev[i] > FW_FC_DIRECT ? 1u : 0u;
        const bool usable = ctx->fc_ok && ctx->fc_dt_bits == dt_bits && ctx->fc_tab_seq == ctx->tab_seq && a.epoch != 1u &&
                            ctx->fc_sums_prev == a.fc_sums &&  // the previous frame left the other format otherwise
                            (!legacy || (a.use_stream && a.new_static));
        if (ctx->fc_dirty) {
            // the tile indexing changed: sums left at indices of the old table must not leak into the new one
            FW_HIP(ctx, hipMemsetAsync(ctx->d_fc, 0, 3 * ctx->fc_len * sizeof(unsigned long long), ctx->stream));
            ctx->fc_dirty = false;
        }
        a.fc_s2 = (uint32_t)ctx->tiles_cap;
        a.fc_tag = (uint32_t)(ctx->fc_len - 1);
        a.fc_out = ctx->d_fc + (size_t)(ctx->fc_seq % 3u) * ctx->fc_len;
        a.fc_zero = ctx->d_fc + (size_t)((ctx->fc_seq + 1u) % 3u) * ctx->fc_len;
        a.fce_out = ctx->d_fce + (size_t)(ctx->fc_seq & 1u) * ctx->tiles_cap;
        a.fce_in = ctx->d_fce + (size_t)((ctx->fc_seq + 1u) & 1u) * ctx->tiles_cap;
        if (usable) a.fc_in = ctx->d_fc + (size_t)((ctx->fc_seq + 2u) % 3u) * ctx->fc_len;
        ctx->fc_seq++;
        ctx->fc_sums_prev = a.fc_sums;
    }
    // a snapshot row stays armed until its stores have been seen (a free-running host can be hundreds of frames
    // ahead of the device; re-arming by frame number would never catch one)
    int snap = -1;
    if ((ctx->frame % ctx->snap_every) == 0)
        for (int k = 0; k < kSnapRing && snap < 0; k++)
            if (!ctx->snap_pending[k]) snap = k;
    // (small types are bounded by their lifetime windows and have no grid to size: a context of nothing else takes no snapshots --
    // a row is a pass over every segment record on the host and a store over the bus per segment on the device)
    if (ctx->n_in_use == ctx->n_small) snap = -1;
    const bool take_snap = snap >= 0;
    a.host_counts = take_snap ? ctx->h_snap + (size_t)snap * ctx->max_seg : nullptr;

    FwInlineOps inl;
    int spawn_form = FW_SPAWN_NONE;
    int slot = -1;
    prof(4);

    if (legacy) {
        // Frames with Nested entries: parents spawned earlier in the frame must exist in memory before the
        // per-parent pass reads them (core.rs:488), so Global ops are materialised by fw_k_spawn, level by level.
        const size_t off_nops = n_g * sizeof(FwOp);
        const size_t bytes = off_nops + n_n * sizeof(FwNestOp) + 16;
        if ((st = ensure_param_ring(ctx, bytes))) return st;
        slot = (int)(ctx->ring_seq++ % kParamRing);
        if (ctx->consumed_pending[slot]) {
            FW_HIP(ctx, hipEventSynchronize(ctx->ev_consumed[slot]));
            ctx->consumed_pending[slot] = false;
        }
        if (ctx->slot_frame[slot]) {  // zero-copy use: wait until a launch AFTER that frame has started
            const volatile unsigned long long *tag = ctx->h_done;
            for (int spin = 0; *tag < ctx->slot_frame[slot] && spin < 200000; spin++) __builtin_ia32_pause();
            if (*tag < ctx->slot_frame[slot]) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->slot_frame[slot] = 0;
        }
        char *hp = ctx->h_param[slot];
        char *dp = ctx->d_param[slot];  // staged copy here: the few workgroups of the small spawn / nest kernels would
                                        // wait for the bus on their critical path (measured: no gain from reading in place)
        struct Launch {
            bool nested;
            size_t first, count;
            uint32_t blocks;
        };
        std::vector<Launch> launches;
        FwOp *h_ops = (FwOp *)hp;
        FwNestOp *h_nops = (FwNestOp *)(hp + off_nops);
        size_t gi = 0, ni = 0, pend_first = 0;
        uint32_t pend_blocks = 0;
        auto flush_global = [&]() {
            if (gi > pend_first) launches.push_back(Launch{false, pend_first, gi - pend_first, pend_blocks});
            pend_first = gi;
            pend_blocks = 0;
        };
        for (auto &L : levels) {
            for (FwOp op : L.g) {
                op.first_block = pend_blocks;
                pend_blocks += (op.n + FW_BLOCK - 1) / FW_BLOCK;
                h_ops[gi++] = op;
            }
            if (!L.n.empty()) {
                flush_global();
                const size_t first = ni;
                uint32_t tiles = 0;
                for (FwNestOp op : L.n) {
                    op.first_tile = tiles;
                    tiles += op.n_tiles;
                    op.parent_buf = ctx->segs[op.parent_seg].buf[p];
                    op.parent_cap = ctx->segs[op.parent_seg].capacity;
                    // (a range ring: the slot of its first YOUNG particle as of the last update -- this frame's cohorts join the
                    // old part further down, after these launches have been enqueued -- and the device subtracts the old part)
                    const SegHost &PS = ctx->segs[op.parent_seg], &CS = ctx->segs[op.child_seg];
                    op.parent_head = PS.fifo ? PS.head : (PS.range ? PS.young_lo : 0u);
                    op.child_head = CS.fifo ? CS.head : (CS.range ? CS.young_lo : 0u);
                    op.parent_range = PS.range ? 1u : 0u, op.child_range = CS.range ? 1u : 0u;
                    op.parent_nospin = ctx->segs[op.parent_seg].nospin ? 1u : 0u;
                    memcpy(op.parent_rot, ctx->segs[op.parent_seg].const_rot, sizeof op.parent_rot);
                    // (its lifetimes: the lifetime plane of a compacting segment, one value for a ring)
                    op.parent_life_plane = ctx->segs[op.parent_seg].fifo ? 0xFFFFFFFFu : ctx->segs[op.parent_seg].n_lplanes;
                    op.parent_life_const = ctx->segs[op.parent_seg].fifo_life;
                    // (START tickets, fw_kernels.h: every workgroup of the op takes one)
                    op.ticket_base = ctx->nest_ticket_base[op.emit_slot], ctx->nest_ticket_base[op.emit_slot] += op.n_tiles;
                    h_nops[ni++] = op;
                }
                launches.push_back(Launch{true, first, ni - first, tiles});
            }
        }
        flush_global();
        // launches with at most FW_INLINE_OPS ops carry them in their kernel arguments; only longer lists (many
        // spawners with Nested entries) are staged through the copy stream
        bool staged = false;
        for (const Launch &L : launches) staged |= L.count > FW_INLINE_OPS;
        if (staged) {
            FW_HIP(ctx, hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
        }
        for (const Launch &L : launches) {
            const bool inl_ops = L.count <= FW_INLINE_OPS;
            if (!L.nested) {
                FW_HIP(ctx, fw_launch_spawn(ctx->stream, ctx->g, inl_ops ? nullptr : (const FwOp *)dp + L.first,
                                            h_ops + L.first, (uint32_t)L.count, L.blocks, p));
            } else {
                ctx->nest_seq = (ctx->nest_seq + 1u) & 0x3FFFFFFFu;
                if (!ctx->nest_seq) ctx->nest_seq = 1u;
                FW_HIP(ctx, fw_launch_nested(ctx->stream, ctx->g,
                                             inl_ops ? nullptr : (const FwNestOp *)(dp + off_nops) + L.first,
                                             h_nops + L.first, (uint32_t)L.count, L.blocks, p, ctx->nest_seq,
                                             ctx->spin_limit, ctx->dbg));
            }
        }
        if (!staged) slot = -1;  // nothing in the ring slot is read by the device: no consumed-event needed
    } else {
        // Global-only frame: spawn is fused into the update kernel (virtual particles).  Ops sorted by segment;
        // the order inside a segment stays the emission order (rel_base was assigned in that order).
        // (one emission level holds all of them most of the time: its list is used as it is)
        std::vector<FwOp> *one = nullptr;
        size_t n_lists = 0;
        for (auto &L : levels)
            if (!L.g.empty()) one = &L.g, n_lists++;
        if (n_lists != 1) {
            one = &ctx->ops_scratch;
            one->clear();
            one->reserve(n_g);
            for (auto &L : levels) one->insert(one->end(), L.g.begin(), L.g.end());
        }
        std::vector<FwOp> &ops = *one;
        // sorted by segment, emission order kept inside a segment; a single emission level is already in spawner
        // (= segment creation) order most of the time: skip the sort then
        if (!std::is_sorted(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; }))
            std::stable_sort(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; });
        a.n_ops = (uint32_t)ops.size();
        if (ops.size() <= FW_INLINE_OPS && (ops.empty() || !ctx->n_small)) {  // (small types read their ops from the table: fw_k_update_small)
            spawn_form = ops.empty() ? FW_SPAWN_NONE : FW_SPAWN_INLINE;
            for (size_t i = 0; i < ops.size(); i++) inl.ops[i] = ops[i];
        } else {
            spawn_form = FW_SPAWN_TABLE;
            struct OpHdr {  // FwUpdateArgs::seg_op_first: per segment {first op, one past its last, particles they spawn in all, 0}
                uint32_t o0, o1, n, pad;
            };
            const size_t off_ops = (size_t)n_seg * sizeof(OpHdr);
            const size_t bytes = off_ops + ops.size() * sizeof(FwOp);
            if ((st = ensure_param_ring(ctx, bytes))) return st;
            slot = (int)(ctx->ring_seq++ % kParamRing);
            if (ctx->consumed_pending[slot]) {
                FW_HIP(ctx, hipEventSynchronize(ctx->ev_consumed[slot]));
                ctx->consumed_pending[slot] = false;
            }
            if (ctx->slot_frame[slot]) {  // zero-copy use: wait until a launch AFTER that frame has started
                const volatile unsigned long long *tag = ctx->h_done;
                for (int spin = 0; *tag < ctx->slot_frame[slot] && spin < 200000; spin++) __builtin_ia32_pause();
                if (*tag < ctx->slot_frame[slot]) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
                ctx->slot_frame[slot] = 0;
            }
            char *hp = ctx->h_param[slot];
            char *dp = ctx->d_param[slot];
            OpHdr *hdr = (OpHdr *)hp;
            size_t oi = 0;
            for (uint32_t sgi = 0; sgi < n_seg; sgi++) {  // (ops are sorted by segment)
                const size_t b = oi;
                uint64_t n = 0;
                while (oi < ops.size() && ops[oi].seg == sgi) n += ops[oi].n, oi++;
                hdr[sgi] = OpHdr{(uint32_t)b, (uint32_t)oi, (uint32_t)std::min<uint64_t>(n, 0xFFFFFFFFull), 0u};
            }
            memcpy(hp + off_ops, ops.data(), ops.size() * sizeof(FwOp));
            if (ctx->ops_zerocopy) {
                // pinned host memory is device-visible: the tiles read their few ops over the bus (tens of bytes each)
                a.seg_op_first = (const uint4 *)hp;
                a.ops = (const FwOp *)(hp + off_ops);
                ctx->slot_frame[slot] = ctx->frame + 1;  // free once done_tag >= frame + 1
                slot = -1;                                // no consumed-event for this slot
            } else {
                FW_HIP(ctx, hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
                FW_HIP(ctx, hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
                FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
                a.seg_op_first = (const uint4 *)dp;
                a.ops = (const FwOp *)(dp + off_ops);
            }
        }
    }
    if (ctx->h_done) a.done_tag = ctx->h_done, a.done_value = ctx->frame;
    prof(5);

    // update_particles + compaction (core.rs:577-670)
    // timing: the events ride on the dispatch packet (its begin / end timestamps), no marker packets in the stream
    bool timed_frame = false;
    auto next_timing_pair = [&](hipEvent_t *e0, hipEvent_t *e1) {
        *e0 = *e1 = nullptr;
        if (!ctx->timing || ctx->tev_used + 2 > ctx->tev.size()) return;
        *e0 = ctx->tev[ctx->tev_used], *e1 = ctx->tev[ctx->tev_used + 1];
        ctx->tev_used += 2;
        timed_frame = true;
    };
    // ---- FIFO segments: in place, everything they need in the kernel arguments (fw_kernels.h: FwFifoSeg)
    bool fifo_launched = false;
    if (ctx->n_fifo) {
        // the ring launch(es) of this frame: on the side stream when a general launch runs next to them (fw_ctx: fifo_stream)
        // (never on a caller-supplied stream: work the caller orders behind fw_step on ITS stream must cover the whole
        // frame, as it did before the side stream existed)
        bool side = ctx->use_fifo_stream && ctx->own_stream && total_tiles != 0 && ctx->live_ring == nullptr;
        // (... nor with a colliding ring: a new collider set travels in the MAIN stream, fw_ctx_set_colliders)
        for (const SegHost &S : ctx->segs) side &= !(S.in_use && S.fifo && (S.fifo_mat || S.inst != nullptr || S.collides));
        if (side && (!ctx->fifo_last_side || ctx->main_reads_ring)) {
            // the previous ring launch, or a reader of ring data, sits on the main stream: this launch comes after it
            FW_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->fifo_stream, ctx->ev_main, 0));
            ctx->main_reads_ring = false;
        } else if (!side && ctx->fifo_last_side && ctx->side_dirty) {
            FW_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->fifo_stream));
            FW_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
            ctx->side_dirty = false;
        }
        ctx->fifo_last_side = side;
        const hipStream_t fstream = side ? ctx->fifo_stream : ctx->stream;
        FwFifoArgs fa{};
        FwInlineOps fio;
        // (with a colliding ring in the context every FIFO launch of it runs the COLL instantiation, whose workgroups cover one
        // round -- FW_FIFO_COLL_TILE slots -- each: the tile grid of all of them follows)
        bool fifo_coll = false;
        // ... and so do the launches of a context whose rings hold too few particles to fill the chip with four-round workgroups
        // (the reference's own stress_test: 157k particles = 154 of them on 256 CUs, each lane working through four particles one
        // after the other): below FW_FIFO_SMALL four-round tiles in all, one round per workgroup.  Not with a ring whose count
        // only the device knows (its grid covers its capacity: four times the idle workgroups).
        // (counted in the first pass over the segments of this frame; a ring that changed its kind since then: next frame)
        const bool fifo_small = !ring_stats.fifo_dev && ring_stats.fifo_parts < (uint64_t)ctx->fifo_small_tiles * FW_TILE;
        const bool fifo_coll_real = ring_stats.fifo_coll;
        fifo_coll = fifo_coll_real || fifo_small;  // (the same tile grid; which instantiation runs: FwFifoArgs::any_coll / small_tiles)
        uint32_t f_ops = 0, f_tiles = 0;
        uint64_t f_bytes = 0;  // what the launch streams, roughly: its tiles x the bytes a particle of the type moves
        uint32_t nest_status_next = 0;  // look-back words handed to the Nested entries of this launch so far (FwFifoNest::status_first)
        if (fuse) {
            ctx->nest_seq = (ctx->nest_seq + 1u) & 0x3FFFFFFFu;
            if (!ctx->nest_seq) ctx->nest_seq = 1u;
        }
        auto flush = [&]() -> hipError_t {
            if (!fa.n_segs) return hipSuccess;
            fa.parity = p, fa.epoch = a.epoch, fa.dbg = ctx->dbg, fa.dt = dt;
            // (the pinned "frame F has started" word recycles host buffers the GENERAL launch reads: when the two launches
            // run on different streams only that one reports)
            fa.done_tag = side ? nullptr : a.done_tag, fa.done_value = a.done_value;
            fa.host_counts = a.host_counts;
            fa.live_out = a.live_out, fa.live_next = a.live_next;
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            const hipError_t e = fw_launch_update_fifo(fstream, ctx->g, fa, fio, f_tiles, f_bytes > ctx->nt_bytes ? 2 : f_bytes > ctx->nt_wo_bytes ? 1 : 0, e0, e1);
            if (side) ctx->side_dirty = true;
            fa = FwFifoArgs{};
            f_ops = f_tiles = 0, f_bytes = 0;
            fifo_launched = true;
            return e;
        };
        for (uint32_t si = 0; si < n_seg; si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.fifo) continue;
            uint32_t k_ops = 0;
            for (const FwOp &op : ctx->fifo_ops) k_ops += op.seg == si ? 1u : 0u;
            if (k_ops > FW_INLINE_OPS)  // (build_spawner never makes such a type a ring)
                return poison_segment(ctx, si, "a FIFO ring with more spawn ops than its launch can carry");
            if (fa.n_segs == FW_FIFO_PER_LAUNCH || f_ops + k_ops > FW_INLINE_OPS) FW_HIP(ctx, flush());
            const int32_t wm = S.derived ? 0 : S.fifo_wm;  // (FW_TYPE_DERIVED: none of the optional planes is stored)
            if (!fa.n_segs) fa.write_mask = wm;
            else if (fa.write_mask != wm) fa.write_mask = -1;
            // frames that materialise (Nested pass): the segment's Global particles of this frame already sit in the ring
            const bool mat_frame = S.fifo_mat && nested_frame && !S.virt_parent;
            const bool mat = S.fifo_dev || mat_frame;
            const uint32_t n_spawn = mat_frame ? 0u : S.frame_spawn;  // spawned by fw_k_update_fifo itself
            // live particles before fw_k_update_fifo's own spawns (a type that receives children: only the device knows)
            const uint32_t n_in = S.fifo_dev ? 0xFFFFFFFFu : S.ub - n_spawn;
            // the cohorts age by this dt exactly as their particles do (fp32 additions, fw_survives); the oldest die first
            if (S.fifo_dev) {
                S.coh.push_back(SegHost::Cohort{0u, 0.0f, ctx->frame, false});  // size: whatever the device appends
            } else if (S.frame_spawn) {
                if (!S.coh.empty() && S.coh.back().age == 0.0f && !std::signbit(S.coh.back().age))
                    S.coh.back().n += S.frame_spawn;
                else
                    S.coh.push_back(SegHost::Cohort{S.frame_spawn, 0.0f, ctx->frame, true});
            }
            for (auto &c : S.coh) c.age = c.age + dt;
            uint32_t dead = 0;
            while (!S.coh.empty() && S.coh.front().age >= S.fifo_life) {
                SegHost::Cohort &c = S.coh.front();
                if (!c.known && c.frame != ctx->frame) {
                    // children added `lifetime` ago: the update of that frame left their number in the pinned ring
                    const uint32_t ep = (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) ? (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) : 1u;
                    const volatile unsigned long long *row = S.h_report + (c.frame % kReportRing);
                    for (int spin = 0; (uint32_t)(*row >> 32) != ep && spin < 100000; spin++) __builtin_ia32_pause();
                    if ((uint32_t)(*row >> 32) != ep) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if ((uint32_t)(*row >> 32) != ep) return poison_segment(ctx, si, "cohort report missing");
                    c.n = (uint32_t)*row, c.known = true;
                }
                dead += c.n;
                S.coh.pop_front();
            }
            FwFifoSeg &F = fa.s[fa.n_segs++];
            F.buf = S.buf[0], F.destroyed = S.destroyed, F.inst = S.inst;
            F.inst_cap = S.inst_cap, F.capacity = S.capacity, F.seg = si;
            F.type_idx = S.type_idx | (S.nospin ? FW_TYPE_IDX_NOSPIN : 0u), F.life = S.fifo_life;
            F.keys_off = S.keys_off, F.keys_len = S.keys_len;
            F.head = S.head, F.n_in = n_in, F.n_spawn = n_spawn, F.dead = dead;
            F.mat = mat ? 1u : 0u;
            F.n_lplanes = S.n_lplanes;
            F.report = S.fifo_dev ? S.h_report + (ctx->frame % kReportRing) : nullptr;
            F.op0 = f_ops;
            if (!mat_frame)
                for (const FwOp &op : ctx->fifo_ops)
                    if (op.seg == si) fio.ops[f_ops++] = op;
            F.op1 = f_ops;
            // workgroups: the new particles first, FW_BLOCK each, in two groups of consecutive slots (up to the end of the
            // buffer / from slot 0); then the ring tiles from the first slot the update touches (the first destroyed particle
            // when their records are wanted, the first survivor otherwise) to the last old particle (a type whose count
            // only the device knows: the whole ring, empty tiles leave at once); at least one in all (it publishes the counts)
            const uint32_t n_old = S.fifo_dev ? S.capacity : n_in;
            // (... or a ring whose particles a Nested entry of this launch emits from: the ones about to die still emit, and the
            // tiles' ranks count from the ring's head)
            int nest_parent = -1, nest_child = -1;
            for (uint32_t k = 0; fuse && k < n_fuse; k++) {
                if (fuse_plan[k].parent_seg == si) nest_parent = (int)k;
                if (fuse_plan[k].child_seg == si) nest_child = (int)k;
            }
            const uint32_t lo = std::min((S.destroyed || nest_parent >= 0) ? 0u : dead, n_old), cnt = n_old - lo;
            const uint32_t ftile = fifo_coll ? FW_FIFO_COLL_TILE : FW_TILE;
            const uint32_t ps = (uint32_t)(((uint64_t)S.head + lo) % S.capacity), ring_tiles = S.capacity / ftile;
            const uint32_t ns0 = (uint32_t)(((uint64_t)S.head + (S.fifo_dev ? 0u : n_in)) % S.capacity);  // slot of the first new particle
            F.spawn_a = std::min(n_spawn, S.capacity - ns0);
            F.n_vt_a = (F.spawn_a + FW_BLOCK - 1) / FW_BLOCK, F.n_vt_b = (n_spawn - F.spawn_a + FW_BLOCK - 1) / FW_BLOCK;
            F.tile0 = ps / ftile;
            const uint32_t live_tiles = cnt ? std::min<uint32_t>(ring_tiles, (ps % ftile + cnt + ftile - 1) / ftile) : 0u;
            F.n_tiles = std::max(1u, F.n_vt_a + F.n_vt_b + live_tiles);
            F.tile_first = f_tiles;
            f_tiles += F.n_tiles;
            F.nest = 0u;
            if (nest_parent >= 0) {
                const FwNestOp &op = fuse_plan[nest_parent];
                FwFifoNest &N = fa.nest[nest_parent];
                F.nest = (uint32_t)nest_parent + 1u;
                N.parent = fa.n_segs - 1u;
                N.emit = op.emit, N.emit_slot = op.emit_slot, N.parent_lplane = op.parent_lplane;
                N.n_count = op.n_count, N.n_start = op.n_start, N.n_end = op.n_end, N.speed = op.speed, N.scale = op.scale;
                N.status_first = nest_status_next, N.n_ptiles = F.n_tiles - (F.n_vt_a + F.n_vt_b);
                nest_status_next += N.n_ptiles;
                N.ticket_base = ctx->nest_ticket_base[op.emit_slot], ctx->nest_ticket_base[op.emit_slot] += N.n_ptiles;
                N.tag = ctx->nest_seq, N.spin_limit = ctx->spin_limit;
                fa.n_nest = n_fuse;
            }
            if (nest_child >= 0) {
                F.nest = ((uint32_t)nest_child + 1u) | FW_FIFO_NEST_CHILD;
                fa.nest[nest_child].child = fa.n_segs - 1u;
            }
            f_bytes += (uint64_t)live_tiles * ftile * (S.nospin ? 104u : 164u);
            fa.any_inst |= S.inst != nullptr ? 1u : 0u;
            fa.any_coll |= fifo_coll_real ? 1u : 0u;
            fa.small_tiles = (fifo_coll && !fifo_coll_real) ? 1u : 0u;
            S.head = (uint32_t)(((uint64_t)S.head + dead) % S.capacity);
            if (!S.fifo_dev) S.ub = n_in + n_spawn - std::min(dead, n_in + n_spawn);  // exact
        }
        FW_HIP(ctx, flush());
    }
    // ---- range rings: in place, one launch for all of them (fw_kernels.h: FwRangeRec)
    bool range_launched = false;
    if (ctx->n_range) {
        const int rslot = (int)(ctx->rring_seq++ % kParamRing);
        if (ctx->rslot_frame[rslot]) {  // the kernel reads the slot in place: free once a launch AFTER that frame has started
            const volatile unsigned long long *tag = ctx->h_done;
            for (int spin = 0; *tag < ctx->rslot_frame[rslot] && spin < 200000; spin++) __builtin_ia32_pause();
            if (*tag < ctx->rslot_frame[rslot]) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->rslot_frame[rslot] = 0;
        }
        FwRangeRec *recs = (FwRangeRec *)ctx->h_rparam[rslot];
        FwOp *rops = (FwOp *)(ctx->h_rparam[rslot] + round_up((uint32_t)(ctx->max_seg * sizeof(FwRangeRec)), 64));
        std::vector<FwOp> &ops = ctx->range_ops;
        if (!std::is_sorted(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; }))
            std::stable_sort(ops.begin(), ops.end(), [](const FwOp &x, const FwOp &y) { return x.seg < y.seg; });
        if (!ops.empty()) memcpy(rops, ops.data(), ops.size() * sizeof(FwOp));
        // age of a particle born in frame f before this frame's update (fw_ctx::birth_age: one entry per frame, contiguous)
        auto age_before = [&](uint64_t f) -> float {
            if (ctx->birth_age.empty() || f < ctx->birth_age.front().frame) return INFINITY;  // (long graduated)
            const size_t i = (size_t)(f - ctx->birth_age.front().frame);
            return i < ctx->birth_age.size() ? ctx->birth_age[i].age : 0.0f;  // this frame's own cohort: born with age 0
        };
        bool dirty = ctx->r_force, all_nospin = true, range_inst = false, range_coll = false;
        {  // the tile size of this launch (fw_ctx::range_small)
            const uint64_t parts = ring_stats.range_parts;
            const bool any_dev = ring_stats.range_dev, any_coll_r = ring_stats.range_coll;
            const uint64_t lim = (uint64_t)ctx->range_small_tiles * FW_TILE;
            const bool small = any_coll_r || (!any_dev && parts < (ctx->range_small ? lim + lim / 4 : lim));
            if (small != ctx->range_small) ctx->range_small = small, dirty = true;
            // ... and of its YOUNG workgroups (fw_ctx::range_young_rounds)
            const uint64_t mean = parts / std::max<uint32_t>(1u, ctx->n_range), big = ctx->range_young_big;
            uint32_t yr = ctx->range_young_rounds;
            if (small || any_dev || ring_stats.range_inst || big == 0) yr = 4;
            else if (yr == 4 && mean >= big) yr = 2;
            else if (yr == 2 && mean < big - big / 4) yr = 4;
            if (yr != ctx->range_young_rounds) ctx->range_young_rounds = yr, dirty = true;
        }
        const uint32_t OT = ctx->range_small ? (uint32_t)FW_BLOCK : (uint32_t)FW_TILE;  // slots an OLD workgroup covers
        uint64_t r_bytes = 0;  // what the launch streams, roughly (the non-temporal form of the kernel: fw_ctx::nt_bytes)
        size_t oi = 0;
        for (uint32_t si = 0; si < n_seg; si++) {
            SegHost &S = ctx->segs[si];
            if (!S.in_use || !S.range) continue;
            all_nospin &= S.nospin;
            range_inst |= S.inst != nullptr;
            range_coll |= S.collides;
            r_bytes += (uint64_t)(S.range_dev ? S.capacity / 2 : S.ub) * (S.nospin ? 104u : 164u);
            S.dead_at_end = true;
            // cohorts that are no longer provably too young to die join the old part: the boundary moves, nothing is copied
            const bool mat_frame = S.range_mat && nested_frame && !S.virt_parent;  // its Global particles of this frame already sit in the ring
            uint32_t grad = 0;
            if (!S.range_dev) {
                while (!S.ycoh.empty()) {
                    const float an = age_before(S.ycoh.front().frame) + dt;  // the device's own addition (core.rs:594)
                    if (an < S.range_life_lo) break;
                    grad += S.ycoh.front().n;
                    S.ycoh.pop_front();
                }
                if (S.frame_spawn) S.ycoh.push_back(SegHost::YCohort{ctx->frame, S.frame_spawn});
            } else {
                // a type that receives Nested children: a cohort's size is whatever the device appended in its frame; the update
                // of that frame left it in the pinned ring, and it is only needed now, a lifetime.min later
                while (!S.dcoh.empty()) {
                    SegHost::DCohort &c = S.dcoh.front();
                    const float an = age_before(c.frame) + dt;
                    if (an < S.range_life_lo) break;
                    if (!c.known) {
                        const uint32_t ep = (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) ? (uint32_t)((c.frame + 1) & 0x3FFFFFFFu) : 1u;
                        const volatile unsigned long long *row = S.h_report + (c.frame % kReportRing);
                        for (int spin = 0; (uint32_t)(*row >> 32) != ep && spin < 100000; spin++) __builtin_ia32_pause();
                        if ((uint32_t)(*row >> 32) != ep) FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        if ((uint32_t)(*row >> 32) != ep) return poison_segment(ctx, si, "cohort report missing (range ring)");
                        c.n = (uint32_t)*row, c.known = true;
                    }
                    grad += c.n;
                    if (c.n) S.gcoh.push_back(SegHost::YCohort{c.frame, c.n}), S.gcoh_sum += c.n;
                    S.dcoh.pop_front();
                }
                S.dcoh.push_back(SegHost::DCohort{ctx->frame, 0u, false});
                // graduated cohorts whose every particle an EARLIER update has destroyed (age >= lifetime.max before this frame)
                while (!S.gcoh.empty() && !(age_before(S.gcoh.front().frame) < (float)S.life_bound)) {
                    S.gcoh_sum -= S.gcoh.front().n;
                    S.gcoh.pop_front();
                }
            }
            S.young_lo = (uint32_t)(((uint64_t)S.young_lo + grad) % S.capacity);
            const uint32_t y_exist = S.range_dev ? 0u : S.young_n - std::min(S.young_n, grad);
            FwRangeRec &Rc = recs[si];
            Rc.b = S.young_lo, Rc.y_exist = y_exist, Rc.n_spawn = (mat_frame || S.range_dev) ? 0u : S.frame_spawn;
            Rc.grad = grad, Rc.flags = (mat_frame ? FW_RREC_MAT : 0u) | (S.range_dev ? (FW_RREC_MAT | FW_RREC_DEV) : 0u);
            Rc.report = S.range_dev ? S.h_report + (ctx->frame % kReportRing) : nullptr, Rc.pad2 = 0;
            while (oi < ops.size() && ops[oi].seg < si) oi++;
            Rc.op0 = (uint32_t)oi, Rc.op_n = 0;
            while (oi < ops.size() && ops[oi].seg == si) oi++, Rc.op_n++;
            S.young_n = S.range_dev ? 0u : y_exist + S.frame_spawn;
            // workgroups of each role (bands: the table is re-sent only when a need leaves its band)
            const uint32_t YT = ctx->range_small ? (uint32_t)FW_BLOCK : ctx->range_young_rounds * (uint32_t)FW_BLOCK;
            uint32_t need_old, need_new, need_young;
            if (S.range_dev) {
                // the old part: at most the cohorts that have joined it and may still hold survivors (all sizes known); the young
                // part: somewhere behind b -- the grid covers the ring, a tile without young particles leaves at once
                need_old = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>((S.gcoh_sum + OT - 1) / OT, S.capacity / OT + 1));
                need_new = 0u;
                need_young = S.capacity / YT;
            } else {
                const uint32_t live_before_ub = std::min(S.ub - std::min(S.ub, S.frame_spawn), S.capacity);
                const uint32_t old_ub = live_before_ub - std::min(live_before_ub, y_exist);
                need_old = std::max(1u, (old_ub + OT - 1) / OT);
                need_new = mat_frame ? 0u : (S.frame_spawn + FW_BLOCK - 1) / FW_BLOCK;
                need_young = std::min(S.capacity / YT, (S.young_lo % YT + y_exist + (mat_frame ? S.frame_spawn : 0u) + YT - 1) / YT);
            }
            // every provisioned workgroup is dispatched every frame, active or not (~3 us of a slot each): small needs get
            // one spare, large ones an eighth -- a re-sent table is a copy in the stream, an idle workgroup a cost in every frame
            // (the bound of the old part follows the snapshots in a sawtooth: a role grows at once, and shrinks only after its
            // need has stayed far below what is provided for 64 frames in a row -- otherwise the table would be re-sent on
            // every tooth)
            auto fit = [&](uint32_t &have, uint32_t need, uint32_t spare, uint32_t &low) {
                if (need > have) {
                    have = need + spare, low = 0, dirty = true;
                } else if (have > need + need / 4 + spare + 2) {
                    if (++low > 64) have = need + spare, low = 0, dirty = true;
                } else {
                    low = 0;
                }
            };
            S.r_need[0] = need_old, S.r_need[1] = need_new, S.r_need[2] = need_young;
            if (S.range_dev) {
                // the grid covers the ring, but how far behind b the young part reaches is roughly known: the count of the latest
                // snapshot row (+ a fifth, + what a few frames add).  Tiles beyond that are "probably idle" and go to the end of
                // the table, where they run while the launch drains (a tile that does hold particles simply updates them
                // there); the split follows the count in steps of an eighth
                const uint64_t est = (uint64_t)((double)S.dev_count * 1.2 + 8.0 * (double)S.dev_rate) + 2 * YT;
                const uint32_t likely = (uint32_t)std::min<uint64_t>(need_young, (est + YT - 1) / YT);
                if (likely > S.r_young_main || likely + likely / 4 + 8 < S.r_young_main) S.r_young_main = std::min(need_young, likely + likely / 8 + 2), dirty = true;
                S.r_need[2] = S.r_young_main;
            }
            fit(S.r_old, need_old, need_old >= 8 ? need_old / 4 : 1u, S.r_low[0]);
            fit(S.r_new, need_new, need_new ? (need_new >= 8 ? need_new / 8 : 1u) : 0u, S.r_low[1]);
            fit(S.r_young, need_young, need_young >= 16 ? need_young / 8 : 1u, S.r_low[2]);
            S.r_young = std::min(S.r_young, S.capacity / YT);
            // (START tickets, fw_kernels.h: every OLD workgroup the table provides for the segment takes one per launch -- r_old of
            // them, whether the table is re-sent this frame or not: fit() changes the number only together with `dirty`)
            Rc.ticket_base = S.ticket_base, S.ticket_base += S.r_old;
        }
        if (dirty) {
            if (ctx->rtab_pending) {  // (one staging buffer: the previous upload must have left it)
                FW_HIP(ctx, hipEventSynchronize(ctx->ev_rtab));
                ctx->rtab_pending = false;
            }
            // Workgroup order of a segment: its OLD workgroups (k ascending: whoever an old tile waits for has a lower workgroup
            // index), then its NEW ones, then the YOUNG ones; segment after segment, so the latency-bound old tiles of one
            // segment overlap the streaming of its neighbours.  A segment with many NEW workgroups (one large emitter: a
            // thousand of them, each ~5x the arithmetic of a YOUNG one and no memory traffic to speak of) gets them spread
            // over the first three quarters of its YOUNG ones instead of as a block: 374-384 -> 364-375 us at 1 x 16M,
            // nothing elsewhere (profiles/r03/range_spread_new.txt).  (Old and young workgroups interleaved within a segment --
            // so that a context with ONE large segment would not start with a front of old tiles -- was measured: 381 -> 384 us
            // at 1 x 16M, 96 -> 100 us at 512 x 8192: no.)  Look-back words are indexed per segment (old_first + k).
            size_t t = 0;
            bool ok = true;
            auto put = [&](uint32_t si, uint32_t role, uint32_t k) {
                if (t >= ctx->rdesc_cap) {
                    ok = false;
                    return;
                }
                const SegHost &S = ctx->segs[si];
                FwRangeDesc &D = ctx->h_rdesc[t++];
                D.seg = si, D.role_k = (role << 30) | k, D.old_first = S.r_status_base;
                D.type_idx = S.type_idx | (S.nospin ? FW_TYPE_IDX_NOSPIN : 0u);
                D.keys_off = S.keys_off, D.keys_len = S.keys_len, D.n_old = S.r_old, D.pad = 0;
            };
            // Workgroups that are provisioned but probably idle -- the spares of every role, and the upper part of the OLD range
            // (its bound counts everybody older than lifetime.min as alive) -- go to the END of the table, behind every
            // segment's probably-active ones: an idle workgroup still holds a slot for ~2.5 us (descriptor, record, count),
            // and there it does so while the launch drains and slots are free anyway.  An OLD tile stays behind the lower
            // tiles of its segment, so the look-back order holds; a "probably idle" workgroup that does have work simply
            // does it there.
            auto main_old = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_old, std::max(1u, (S.r_need[0] * 5u + 7u) / 8u)) : S.r_old; };
            auto main_new = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_new, S.r_need[1]) : S.r_new; };
            auto main_young = [&](const SegHost &S) { return ctx->range_idle_last ? std::min(S.r_young, S.r_need[2]) : S.r_young; };
            auto put_old = [&](uint32_t si) {
                const uint32_t n = main_old(ctx->segs[si]);
                for (uint32_t k = 0; k < n && ok; k++) put(si, FW_RANGE_OLD, k);
            };
            auto put_rest = [&](uint32_t si) {
                const SegHost &S = ctx->segs[si];
                const uint32_t n_new = main_new(S), n_young = main_young(S);
                if (!ctx->range_spread_new || n_new <= 8) {
                    for (uint32_t k = 0; k < n_new && ok; k++) put(si, FW_RANGE_NEW, k);
                    for (uint32_t k = 0; k < n_young && ok; k++) put(si, FW_RANGE_YOUNG, k);
                } else {  // many NEW workgroups (one large segment): spread over the first three quarters of the YOUNG ones
                    const uint64_t span = (uint64_t)n_new + (uint64_t)n_young * 3 / 4;
                    uint32_t kn = 0, ky = 0;
                    for (uint64_t i = 0; i < span && ok; i++) {
                        if (kn < n_new && (uint64_t)kn * span / n_new <= i) put(si, FW_RANGE_NEW, kn++);
                        else if (ky < n_young) put(si, FW_RANGE_YOUNG, ky++);
                    }
                    while (kn < n_new && ok) put(si, FW_RANGE_NEW, kn++);
                    while (ky < n_young && ok) put(si, FW_RANGE_YOUNG, ky++);
                }
            };
            auto put_tail = [&](uint32_t si) {
                const SegHost &S = ctx->segs[si];
                for (uint32_t k = main_old(S); k < S.r_old && ok; k++) put(si, FW_RANGE_OLD, k);
                for (uint32_t k = main_new(S); k < S.r_new && ok; k++) put(si, FW_RANGE_NEW, k);
                for (uint32_t k = main_young(S); k < S.r_young && ok; k++) put(si, FW_RANGE_YOUNG, k);
            };
            auto &rs = ctx->range_scratch;  // the range segments, in segment order
            rs.clear();
            uint32_t status_base = 0;
            for (uint32_t si = 0; si < n_seg; si++) {
                SegHost &S = ctx->segs[si];
                if (!S.in_use || !S.range) continue;
                S.r_status_base = status_base, status_base += S.r_old;
                rs.push_back(si);
            }
            // (the OLD workgroups of a segment dispatched n segments ahead of its other ones: measured, no gain --
            // profiles/r03/range_old_ahead.txt)
            const size_t nr = rs.size();
            // (XCD-aware order -- the runs of eight consecutive segments interleaved entry by entry, so that the workgroups of one
            // segment share an XCD and its L2 -- was built and measured in round 4: nothing at one GPU's share of configs[4]
            // (86.4 against 86.5 us), 2.5 % slower at configs[2]: profiles/r04/range_xcd_order_ab.txt)
            for (size_t i = 0; i < nr && ok; i++) put_old(rs[i]), put_rest(rs[i]);
            for (size_t i = 0; i < nr && ok; i++) put_tail(rs[i]);
            if (!ok) return poison_segment(ctx, kNoSeg, "range table overflow");
            ctx->r_total = (uint32_t)t;
            if (t) FW_HIP(ctx, hipMemcpyAsync(ctx->d_rdesc, ctx->h_rdesc, t * sizeof(FwRangeDesc), hipMemcpyHostToDevice, ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_rtab, ctx->stream));
            ctx->rtab_pending = true;
            ctx->r_force = false;
            ctx->r_uploads++;
        }
        // the ages every later frame starts from
        for (auto &e : ctx->birth_age) e.age = e.age + dt;
        ctx->birth_age.push_back(fw_ctx::BirthAge{ctx->frame, 0.0f + dt});
        // (kept while some ring may still ask for the age: until lifetime.min for every ring, until lifetime.max for those that
        // bound their old part by the cohorts in it -- range_age_keep)
        while (!ctx->birth_age.empty() && !(ctx->birth_age.front().age < std::max(ctx->range_life_max, ctx->range_age_keep))) ctx->birth_age.pop_front();
        if (ctx->r_total) {
            FwRangeArgs ra{};
            ra.desc = ctx->d_rdesc, ra.recs = recs, ra.ops = rops, ra.status = ctx->d_rstatus;
            ra.total_tiles = ctx->r_total, ra.parity = p, ra.epoch = a.epoch, ra.spin_limit = ctx->spin_limit, ra.dbg = ctx->dbg;
            ra.dt = dt;
            ra.any_inst = range_inst ? 1u : 0u;
            ra.any_coll = range_coll ? 1u : 0u;
            ra.small_tiles = ctx->range_small ? 1u : 0u;
            ra.young_rounds = ctx->range_young_rounds;
            ra.done_tag = a.done_tag, ra.done_value = a.done_value;
            ra.host_counts = a.host_counts, ra.live_out = a.live_out, ra.live_next = a.live_next;
            ra.ts = ctx->d_rts;
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            FW_HIP(ctx, fw_launch_update_range(ctx->stream, ctx->g, ra, all_nospin, r_bytes > ctx->nt_bytes ? 2 : r_bytes > ctx->nt_wo_bytes_range ? 1 : 0, e0, e1));
            ctx->rslot_frame[rslot] = ctx->frame + 1;
            range_launched = true;
        }
    }
    // ---- small types: one wave each (fw_k_small.hip)
    bool small_launched = false;
    if (ctx->n_small) {
        if (ctx->small_dirty) {  // the list changed (a spawner built or destroyed, a type that outgrew the mode): re-sent through the stream
            ctx->small_list.clear();
            for (uint32_t si = 0; si < n_seg; si++)
                if (ctx->segs[si].in_use && ctx->segs[si].small) ctx->small_list.push_back(si);
            if (ctx->small_list.size() > ctx->small_cap) return poison_segment(ctx, kNoSeg, "small-type list overflow");
            if (ctx->small_pending) {
                FW_HIP(ctx, hipEventSynchronize(ctx->ev_small));
                ctx->small_pending = false;
            }
            memcpy(ctx->h_small, ctx->small_list.data(), ctx->small_list.size() * sizeof(uint32_t));
            FW_HIP(ctx, hipMemcpyAsync(ctx->d_small, ctx->h_small, ctx->small_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->ev_small, ctx->stream));
            ctx->small_pending = true, ctx->small_dirty = false;
        }
        FwSmallArgs sa{};
        sa.list = ctx->d_small, sa.n = (uint32_t)ctx->small_list.size(), sa.parity = p, sa.epoch = a.epoch, sa.dt = dt;
        sa.seg_op_first = spawn_form == FW_SPAWN_TABLE ? a.seg_op_first : nullptr, sa.ops = a.ops;
        sa.force_colors = a.force_colors, sa.dbg = ctx->dbg;
        sa.done_tag = a.done_tag, sa.done_value = a.done_value;
        sa.host_counts = a.host_counts, sa.live_out = a.live_out, sa.live_next = a.live_next;
        if (sa.n) {
            hipEvent_t e0, e1;
            next_timing_pair(&e0, &e1);
            FW_HIP(ctx, fw_launch_update_small(ctx->stream, ctx->g, sa, e0, e1));
            small_launched = true;
        }
    }
    if (total_tiles || !(fifo_launched || range_launched || small_launched)) {
        hipEvent_t e0, e1;
        next_timing_pair(&e0, &e1);
        FW_HIP(ctx, fw_launch_update(ctx->stream, ctx->g, a, spawn_form == FW_SPAWN_INLINE ? &inl : nullptr, spawn_form,
                                     frame_mode, e0, e1));
    }
    if (timed_frame) ctx->tev_frames++;
    if (ctx->colors_dirty) {
        // that update wrote every colour of its output; the buffer it read (next frame's output) may still hold the
        // caller's colours past the survivors: back to the fill value, after which constant planes are skipped again
        for (uint32_t i = 0; i < n_seg; i++) {
            SegHost &S = ctx->segs[i];
            if (!S.in_use || !S.colors_dirty) continue;
            FW_HIP(ctx, fw_launch_fill_colors(ctx->stream, S.buf[ctx->parity], nullptr, S.capacity, S.fill_bc, S.fill_em));
            S.colors_dirty = false;
        }
        ctx->colors_dirty = false;
        // every particle has been through an update since the caller's write: scale and colours are functions of the age
        // again, a type with an attached instance buffer can stop storing them (FW_TYPE_DERIVED; waits for this frame: rare)
        // (the flag flips at the START of the next fw_step, before anything of that frame is enqueued: flipping it waits for
        // the stream -- the frame just enqueued still stores the planes -- and may fail; fw_step itself only enqueues, and
        // by now this frame must finish its bookkeeping whatever happens)
        for (uint32_t i = 0; i < n_seg; i++)
            if (ctx->segs[i].in_use && ctx->segs[i].derive_pending) {
                ctx->segs[i].derive_pending = false;
                if (ctx->segs[i].inst != nullptr) ctx->segs[i].derive_ready = true, ctx->derive_ready_any = true;
            }
    }
    prof(6);
    if (slot >= 0) {
        FW_HIP(ctx, hipEventRecord(ctx->ev_consumed[slot], ctx->stream));
        ctx->consumed_pending[slot] = true;
    }
    if (take_snap) {
        ctx->snap_pending[snap] = true;
        ctx->snap_seen[snap] = false;
        ctx->snap_epoch[snap] = a.epoch;
        ctx->snap_cum[snap].resize(n_seg);
        for (uint32_t i = 0; i < n_seg; i++) ctx->snap_cum[snap][i] = ctx->segs[i].cum_spawn;
    }

    ctx->fc_ok = fc_frame;
    ctx->boxes_epoch = a.boxes ? a.epoch : 0u;
    ctx->fc_dt_bits = dt_bits;
    ctx->fc_tab_seq = ctx->tab_seq;
    ctx->parity ^= 1u;
    ctx->frame++;
    ctx->sim_time += (double)dt;
    prof(7);
    if (ctx->frame > ctx->host_prof_skip) ctx->prof_frames++;
    return FW_OK;
}

// ---- outputs -----------------------------------------------------------------------------
fw_status fw_spawner_counts(fw_ctx *ctx, fw_spawner h, uint32_t *per_type, uint32_t n_types) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !per_type) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    for (uint32_t t = 0; t < n_types && t < sp->seg.size(); t++) per_type[t] = c[sp->seg[t]];
    return st;
}

fw_status fw_spawner_active(fw_ctx *ctx, fw_spawner h, int32_t *out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    *out = spawner_active(ctx, *sp, c) ? 1 : 0;
    return st;
}

fw_status fw_spawner_poll_finished(fw_ctx *ctx, fw_spawner h, int32_t *out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    bool all_empty = true;
    for (uint32_t si : sp->seg) all_empty &= c[si] == 0;
    *out = 0;
    if (all_empty && !spawner_active(ctx, *sp, c) && sp->initialized && !sp->finished_notified) {  // core.rs:679-686
        sp->finished_notified = true;
        *out = 1;
    }
    return st;
}

static fw_status stage_buffer(fw_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->stage_bytes) {
        fw_status st = sync(ctx);
        if (st) return st;
        if (ctx->d_stage) FW_HIP(ctx, hipFree(ctx->d_stage));
        ctx->d_stage = nullptr, ctx->stage_bytes = 0;
        const size_t nb = (std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20) + 65535u) & ~(size_t)65535u;
        FW_HIP(ctx, hipMalloc(&ctx->d_stage, nb));
        ctx->stage_bytes = nb;
    }
    *out = ctx->d_stage;
    return FW_OK;
}

static fw_status read_records(fw_ctx *ctx, const char *buf, uint32_t cap_seg, uint32_t n, int32_t pbr, bool aos,
                              fw_particle *out, uint64_t cap, uint32_t head = 0, const float *const_rot = nullptr,
                              uint32_t life_plane = 0xFFFFFFFFu, float life_const = 0.f, const FwType *derived = nullptr) {
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return FW_OK;
    if (aos) {
        FW_HIP(ctx, hipMemcpy(out, buf, m * sizeof(fw_particle), hipMemcpyDeviceToHost));
        return FW_OK;
    }
    void *tmp = nullptr;
    fw_status st = stage_buffer(ctx, m * sizeof(fw_particle), &tmp);
    if (st) return st;
    hipError_t e = fw_launch_gather(ctx->stream, buf, cap_seg, head, (uint32_t)m, pbr, tmp, const_rot, life_plane, life_const,
                                    derived, ctx->d_keys.d);
    // (the copy goes through the stream the kernel ran on, then one wait for both)
    if (e == hipSuccess) e = hipMemcpyAsync(out, tmp, m * sizeof(fw_particle), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    FW_HIP(ctx, e);
    return FW_OK;
}

fw_status fw_spawner_read_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    fw_status st2 = read_records(ctx, S.buf[ctx->parity], S.capacity, n, sp->types[type].ps.pbr, false, out, cap,
                                 ring_head_exact(S, n), S.nospin ? S.const_rot : nullptr,
                                 (S.nospin && !S.fifo) ? S.n_lplanes : 0xFFFFFFFFu, S.fifo_life,
                                 S.derived ? ctx->d_types.d + S.type_idx : nullptr);
    return st2 ? st2 : st;
}

fw_status fw_spawner_read_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index, float *out,
                                       uint64_t cap, uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || emission_index >= sp->em.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return st;
    int plane = -1;
    for (uint32_t k = 0; k < S.n_lplanes; k++)
        if (S.lplane_emission[k] == (int32_t)emission_index) plane = (int)k;
    if (plane < 0) {
        for (uint64_t i = 0; i < m; i++) out[i] = FW_F32_MIN;  // never touched: still vec![f32::MIN; n] (core.rs:467)
        return st;
    }
    const char *pl = S.buf[ctx->parity] + FW_OFF_L((size_t)S.capacity, plane);
    const uint32_t h0 = ring_head_exact(S, n);  // a ring: from the head to the end of the buffer, then from slot 0
    const uint64_t m1 = std::min<uint64_t>(m, S.capacity - h0);
    FW_HIP(ctx, hipMemcpy(out, pl + (size_t)h0 * sizeof(float), m1 * sizeof(float), hipMemcpyDeviceToHost));
    if (m > m1) FW_HIP(ctx, hipMemcpy(out + m1, pl, (m - m1) * sizeof(float), hipMemcpyDeviceToHost));
    return st;
}

fw_status fw_spawner_write_particles(fw_ctx *ctx, fw_spawner h, uint32_t type, const fw_particle *in, uint64_t n) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || (n && !in) || n > 0xF0000000ull) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    const uint32_t si = sp->seg[type];
    ctx->fc_ok = false, ctx->boxes_epoch = 0;
    if ((st = fifo_to_general(ctx, si))) return st;  // ages and lifetimes will be whatever the caller writes
    leave_small(ctx, ctx->segs[si]);                  // (any number of particles, any colours: the compacting kernels take it from here)
    if ((st = leave_nospin(ctx, si))) return st;      // ... and so will rotations and angular velocities
    // ... and scales and colours: the planes are stored and read again until every particle has been through an update
    // (an attached buffer keeps receiving records; the mode comes back after the next step)
    if ((st = set_derived(ctx, si, false, false))) return st;
    ctx->segs[si].derive_pending = ctx->segs[si].inst != nullptr && ctx->use_derived && !ctx->segs[si].collides;
    if (n > ctx->segs[si].capacity) {
        ctx->segs[si].ub = 0;
        const uint32_t zero = 0;
        FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, &zero, 4, hipMemcpyHostToDevice));
        if ((st = grow_segment(ctx, si, (uint32_t)n))) return st;
    }
    SegHost &S = ctx->segs[si];
    if (n) {
        void *tmp = nullptr;
        if ((st = stage_buffer(ctx, n * sizeof(fw_particle), &tmp))) return st;
        hipError_t e = hipMemcpyAsync(tmp, in, n * sizeof(fw_particle), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = fw_launch_scatter(ctx->stream, S.buf[ctx->parity], S.capacity, (uint32_t)n, S.n_lplanes, tmp);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        FW_HIP(ctx, e);
    }
    const uint32_t n32 = (uint32_t)n;
    FW_HIP(ctx, hipMemcpy(ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, &n32, 4, hipMemcpyHostToDevice));
    S.ub = n32;
    S.win_ok = false;  // ages and lifetimes are now whatever the caller wrote
    S.colors_dirty = true, ctx->colors_dirty = true;  // and so are the colours: the next update writes them all
    for (int i = 0; i < kSnapRing; i++) ctx->snap_pending[i] = false;
    return FW_OK;
}

fw_status fw_spawner_write_last_emitted(fw_ctx *ctx, fw_spawner h, uint32_t type, uint32_t emission_index,
                                        const float *in, uint64_t n) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || emission_index >= sp->em.size() || (n && !in)) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if ((st = fifo_to_general(ctx, sp->seg[type]))) return st;  // (caller-written state: the general path takes over)
    const SegHost &S = ctx->segs[sp->seg[type]];
    int plane = -1;
    for (uint32_t k = 0; k < S.n_lplanes; k++)
        if (S.lplane_emission[k] == (int32_t)emission_index) plane = (int)k;
    if (plane < 0) return FW_OK;  // entry never reads it
    const uint64_t m = std::min<uint64_t>(n, S.capacity);
    if (m)
        FW_HIP(ctx, hipMemcpy(S.buf[ctx->parity] + FW_OFF_L((size_t)S.capacity, plane), in, m * sizeof(float),
                              hipMemcpyHostToDevice));
    return FW_OK;
}

fw_status fw_spawner_read_destroyed(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    const SegHost &S = ctx->segs[sp->seg[type]];
    uint32_t n = 0;
    if (S.destroyed) FW_HIP(ctx, hipMemcpy(&n, ctx->g.ndestroyed + sp->seg[type], 4, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    // (a range ring fills its records from the END of the buffer, the youngest dead first: the last n are in list order)
    const char *first = S.destroyed + (S.dead_at_end && n <= S.capacity ? (size_t)(S.capacity - n) * sizeof(fw_particle) : (size_t)0);
    return read_records(ctx, first, S.capacity, n, 0, true, out, cap);
}

fw_status fw_spawner_pack_instances_device(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap,
                                           uint64_t *n_upper_bound) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || !d_out) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    const uint32_t si = sp->seg[type];
    const SegHost &S = ctx->segs[si];
    const uint32_t ub = (uint32_t)std::min<uint64_t>(S.nested_fed ? S.capacity : std::min(S.ub, S.capacity), cap);
    if (n_upper_bound) *n_upper_bound = ub;
    if (S.fifo) {
        fw_status jst = join_side(ctx);
        if (jst) return jst;
    }
    // (a range ring: particle 0 sits `count - young_n` slots before the first young particle -- the kernel reads the count)
    FW_HIP(ctx, fw_launch_pack_instances(ctx->stream, S.buf[ctx->parity], S.capacity, S.range ? S.young_lo : (S.fifo ? S.head : 0u),
                                         ctx->g.count + (size_t)ctx->parity * ctx->max_seg + si, ub, d_out,
                                         S.nospin ? S.const_rot : nullptr, S.range ? ctx->g.rold + (size_t)ctx->parity * ctx->max_seg + si : nullptr,
                                         S.derived ? ctx->d_types.d + S.type_idx : nullptr, ctx->d_keys.d, S.life_plane(), S.fifo_life));
    return FW_OK;
}

static fw_status attach_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap, bool window) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || (d_out && !cap)) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);  // kernels in flight hold the old record
    // ... and whatever the caller enqueued on ITS streams to initialise the buffer has happened before a frame writes to it
    // (the context's streams are non-blocking ones: nothing else orders them against, say, a fill on the null stream)
    if (!st && d_out) FW_HIP(ctx, hipDeviceSynchronize());
    if (st) return st;
    if (d_out && !window && ctx->segs[sp->seg[type]].range) {
        // records at index `list position` counted from 0: a tile of a range ring only knows that once the whole old part
        // has been counted -- such a type continues on the compacting path (the windowed attach keeps it a ring)
        if ((st = fifo_to_general(ctx, sp->seg[type]))) return st;
    }
    SegHost &S = ctx->segs[sp->seg[type]];
    if (d_out) leave_small(ctx, S);  // (instance records are written by the compacting kernels)
    S.inst = (char *)d_out;
    S.inst_cap = d_out ? (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFull) : 0u;
    S.inst_window = d_out != nullptr && window;
    // the records carry scale and colours from now on: the update stops storing the three planes that would duplicate them
    // (every reader of those planes evaluates them instead, so a buffer smaller than the live count loses nothing either);
    // colliding types stay as they are (the feature path)
    const bool derive = d_out != nullptr && ctx->use_derived && !S.collides;
    S.derive_pending = derive && S.colors_dirty;  // (particles written by the caller, not updated yet: one frame later)
    if ((st = set_derived(ctx, sp->seg[type], derive && !S.colors_dirty))) return st;
    return upload_seg(ctx, sp->seg[type]);
}

fw_status fw_spawner_attach_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap) {
    return attach_instances(ctx, h, type, d_out, cap, false);
}

fw_status fw_spawner_attach_instances_window(fw_ctx *ctx, fw_spawner h, uint32_t type, void *d_out, uint64_t cap) {
    return attach_instances(ctx, h, type, d_out, cap, true);
}

fw_status fw_spawner_instance_window(fw_ctx *ctx, fw_spawner h, uint32_t type, uint64_t *first, uint64_t *count) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size() || !first || !count) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const uint32_t si = sp->seg[type];
    const SegHost &S = ctx->segs[si];
    uint32_t dead = 0;
    // a range ring's update numbers its records from the particles it destroys (fw_k_update_range): they sit behind them
    if (S.dead_at_end && S.inst_window) FW_HIP(ctx, hipMemcpy(&dead, ctx->g.ndestroyed + si, 4, hipMemcpyDeviceToHost));
    *first = dead, *count = c[si];
    return st;
}

fw_status fw_spawner_pack_instances(fw_ctx *ctx, fw_spawner h, uint32_t type, fw_particle_instance *out, uint64_t cap,
                                    uint64_t *n_out) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    const uint32_t n = c[sp->seg[type]];
    if (n_out) *n_out = n;
    const uint64_t m = std::min<uint64_t>(n, cap);
    if (!m || !out) return st;
    void *tmp = nullptr;
    fw_status sst = stage_buffer(ctx, m * sizeof(fw_particle_instance), &tmp);
    if (sst) return sst;
    uint64_t ub = 0;
    fw_status st2 = fw_spawner_pack_instances_device(ctx, h, type, tmp, m, &ub);
    hipError_t e = hipMemcpyAsync(out, tmp, m * sizeof(fw_particle_instance), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    FW_HIP(ctx, e);
    return st2 ? st2 : st;
}

fw_status fw_spawner_aabb(fw_ctx *ctx, fw_spawner h, float out_min[3], float out_max[3], int32_t *any) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || !out_min || !out_max) return FW_EINVAL;
    if (poll_device_error(ctx), sp->poisoned) return poisoned_status(ctx);
    hipSetDevice(ctx->device);
    if (!ctx->h_aabb) FW_HIP(ctx, hipHostMalloc((void **)&ctx->h_aabb, 8 * sizeof(float), hipHostMallocDefault));
    {
        fw_status jst = join_side(ctx);  // (the query kernels run on the main stream and may read rings)
        if (jst) return jst;
    }
    // the query kernels take up to eight particle types at a time (their segment list rides in the kernel arguments): a
    // spawner with more is answered in chunks, folded here -- min / max are exact and order-independent
    const size_t nt = sp->seg.size();
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    bool got = false;
    fw_status keep = FW_OK;
    for (size_t t0 = 0; t0 < nt || t0 == 0; t0 += 8) {
        const uint32_t n = (uint32_t)std::min<size_t>(8, nt - std::min(nt, t0));
        if (!n) break;
        bool any_ring = false;  // rings leave no per-tile boxes: the two-pass query reads them
        uint32_t heads[8] = {}, range_y[8], life_plane[8];
        float life_const[8];
        for (uint32_t t = 0; t < n; t++) {
            const SegHost &S = ctx->segs[sp->seg[t0 + t]];
            any_ring |= S.ring();
            heads[t] = S.range ? S.young_lo : (S.fifo ? S.head : 0u);
            range_y[t] = S.range ? 1u : 0xFFFFFFFFu;  // (a range ring: the kernel takes the old part's size from FwGlobals::rold)
            life_plane[t] = S.life_plane(), life_const[t] = S.fifo_life;
        }
        if (ctx->boxes_epoch && ctx->d_tile_first && !any_ring) {
            // the last update left the box of every tile's survivors (fw_ctx_track_aabbs): fold those -- one small launch
            FW_HIP(ctx, fw_launch_aabb_from_tiles(ctx->stream, ctx->g, sp->seg.data() + t0, n, ctx->parity, ctx->boxes_epoch,
                                                  ctx->d_tile_first, ctx->h_aabb));
        } else {
            // two launches over the particles, the result lands in pinned memory: one synchronisation, no copies
            FW_HIP(ctx, fw_launch_aabb(ctx->stream, ctx->g, sp->seg.data() + t0, heads, n, ctx->parity, ctx->d_aabb, ctx->h_aabb,
                                       range_y, life_plane, life_const));
        }
        fw_status st = sync(ctx);
        if (!st) st = check_device_errors(ctx);
        if (st && st != FW_ECAPACITY) return st;
        if (st) keep = st;
        const volatile float *r = ctx->h_aabb;
        if (r[3] != 0.0f) {
            for (int c = 0; c < 3; c++) {
                mn[c] = got ? std::min(mn[c], (float)r[c]) : (float)r[c];
                mx[c] = got ? std::max(mx[c], (float)r[4 + c]) : (float)r[4 + c];
            }
            got = true;
        } else if (!got && t0 + 8 >= nt) {  // nothing anywhere: report the last chunk's (empty) box as before
            for (int c = 0; c < 3; c++) mn[c] = r[c], mx[c] = r[4 + c];
        }
    }
    if (any) *any = got ? 1 : 0;
    for (int c = 0; c < 3; c++) out_min[c] = mn[c], out_max[c] = mx[c];
    return keep;
}

fw_status fw_ctx_track_aabbs(fw_ctx *ctx, int32_t enable) {
    if (!ctx) return FW_EINVAL;
    ctx->track_aabb = enable != 0;
    if (!enable) ctx->boxes_epoch = 0;
    if (enable)  // (per-tile boxes are left by the compacting kernels' tiles)
        for (SegHost &S : ctx->segs)
            if (S.in_use) leave_small(ctx, S);
    return FW_OK;
}

fw_status fw_ctx_live_count(fw_ctx *ctx, uint64_t *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    std::vector<uint32_t> c;
    fw_status st = read_counts(ctx, c);
    if (st && st != FW_ECAPACITY) return st;
    uint64_t t = 0;
    for (size_t i = 0; i < c.size(); i++)
        if (ctx->segs[i].in_use) t += c[i];
    *out = t;
    return st;
}

fw_status fw_ctx_live_count_device(fw_ctx *ctx, void *d_out_u64) {
    if (!ctx || !d_out_u64) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status jst = join_side(ctx);  // the counts of ring segments are written by the side stream's launches
    if (jst) return jst;
    FW_HIP(ctx, fw_launch_total(ctx->stream, ctx->g.count + (size_t)ctx->parity * ctx->max_seg,
                                (uint32_t)ctx->segs.size(), (unsigned long long *)d_out_u64));
    return FW_OK;
}

fw_status fw_ctx_live_count_ring(fw_ctx *ctx, void *d_ring_u64, uint32_t n_slots) {
    if (!ctx || (d_ring_u64 && n_slots < 2)) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (!st && d_ring_u64) FW_HIP(ctx, hipDeviceSynchronize());  // (the caller's zero-fill of the ring, on whatever stream)
    if (st) return st;
    ctx->live_ring = (unsigned long long *)d_ring_u64;
    ctx->live_ring_n = d_ring_u64 ? n_slots : 0;
    ctx->live_ring_frames = 0;
    if (d_ring_u64) FW_HIP(ctx, fw_memset_done(d_ring_u64, 0, (size_t)n_slots * sizeof(unsigned long long)));
    return FW_OK;
}

fw_status fw_ctx_last_step_updated(fw_ctx *ctx, uint64_t *out) {
    if (!ctx || !out) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    unsigned long long now = 0;
    FW_HIP(ctx, hipMemcpy(&now, ctx->g.stats, sizeof now, hipMemcpyDeviceToHost));
    *out = now;  // running total of particles that entered update_particles
    return FW_OK;
}

fw_status fw_ctx_kernel_timing(fw_ctx *ctx, int32_t enable) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    if (enable && ctx->tev.empty()) {
        ctx->tev.resize(kTimingEvents);
        for (auto &ev : ctx->tev) FW_HIP(ctx, hipEventCreate(&ev));
    }
    if (enable) {
        // for reference only: what an empty hipEventRecord pair on the stream costs (the timed launches do not use
        // marker packets: their events are attached to the dispatch, see fw_launch_update)
        const int n = 64;
        for (int i = 0; i < n; i++) {
            FW_HIP(ctx, hipEventRecord(ctx->tev[2 * i], ctx->stream));
            FW_HIP(ctx, hipEventRecord(ctx->tev[2 * i + 1], ctx->stream));
        }
        FW_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double tot = 0;
        for (int i = 8; i < n; i++) {
            float t = 0;
            FW_HIP(ctx, hipEventElapsedTime(&t, ctx->tev[2 * i], ctx->tev[2 * i + 1]));
            tot += t;
        }
        ctx->tev_overhead_ms = tot / (n - 8);
    }
    ctx->timing = enable != 0;
    ctx->tev_used = 0, ctx->tev_frames = 0;
    unsigned long long now = 0;
    FW_HIP(ctx, hipMemcpy(&now, ctx->g.stats, sizeof now, hipMemcpyDeviceToHost));
    ctx->timing_particles_start = now;
    return FW_OK;
}

fw_status fw_ctx_kernel_timing_read(fw_ctx *ctx, double *ms_total, uint64_t *launches, uint64_t *particles) {
    if (!ctx) return FW_EINVAL;
    hipSetDevice(ctx->device);
    fw_status st = sync(ctx);
    if (st) return st;
    double ms = 0;
    for (size_t i = 0; i + 1 < ctx->tev_used; i += 2) {
        float t = 0;
        FW_HIP(ctx, hipEventElapsedTime(&t, ctx->tev[i], ctx->tev[i + 1]));
        ms += t;
    }
    const uint64_t nl = ctx->tev_frames;  // frames: a frame's update may be several launches (FIFO + general), all summed
    if (ms_total) *ms_total = ms;
    if (launches) *launches = nl;
    unsigned long long now = 0;
    FW_HIP(ctx, hipMemcpy(&now, ctx->g.stats, sizeof now, hipMemcpyDeviceToHost));
    if (particles) *particles = now - ctx->timing_particles_start;
    return FW_OK;
}

// profiling hook (not in the public header): per-tile timestamps of the last update (and, with `prev`, of the one
// before it: the two are kept apart by launch parity) when FW_DEBUG & 8
fw_status fw_debug_read_timestamps2(fw_ctx *ctx, unsigned long long *out, unsigned long long *prev, uint64_t max_tiles,
                                    uint64_t *n_tiles) {
    if (!ctx || !ctx->g.dbg_ts) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    const uint64_t n = std::min<uint64_t>(max_tiles, ctx->total_tiles_dev);
    if (n_tiles) *n_tiles = n;
    const uint32_t last = (uint32_t)(ctx->frame & 1u);  // epoch of the last launch = frame (after the increment)
    const size_t stride = (size_t)ctx->total_tiles_dev * 8;
    if (n && out)
        FW_HIP(ctx, hipMemcpy(out, ctx->g.dbg_ts + 32768 + last * stride, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (n && prev)
        FW_HIP(ctx, hipMemcpy(prev, ctx->g.dbg_ts + 32768 + (last ^ 1u) * stride, n * 8 * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost));
    return FW_OK;
}
// ... and of the last range-ring launch: 8 words per workgroup {start, 0, 0, end of wave 0, role_k, segment, 0, 0}
fw_status fw_debug_read_range_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles) {
    if (!ctx || !ctx->d_rts) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    const uint64_t n = std::min<uint64_t>(max_tiles, ctx->r_total);
    if (n_tiles) *n_tiles = n;
    if (n && out) FW_HIP(ctx, hipMemcpy(out, ctx->d_rts, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FW_OK;
}
// {~earliest workgroup start [64], latest workgroup end [64]} of the last 256 update launches (slot = epoch & 255);
// *epoch = the last launch's.  out512: room for 32768 words.
fw_status fw_debug_read_launches(fw_ctx *ctx, unsigned long long *out512, uint32_t *epoch) {
    if (!ctx || !ctx->g.dbg_ts || !out512) return FW_EINVAL;
    fw_status st = sync(ctx);
    if (st) return st;
    FW_HIP(ctx, hipMemcpy(out512, ctx->g.dbg_ts, 32768 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (epoch) *epoch = (uint32_t)(ctx->frame & 0x3FFFFFFFu);
    return FW_OK;
}
// which update path a particle type is on: *mode = 1 FIFO ring (in place), 0 general (compacting); *moved_bytes = the
// bytes one update of a live particle reads and writes on that path; *algorithmic_bytes = the part of them that carries
// state the update needs or changes (SURVEY.md 8(d)'s convention: a float4 plane rewritten for three changed components
// moves 4 bytes -- initial_scale, lifetime -- that are not algorithmic)
fw_status fw_debug_update_path(fw_ctx *ctx, fw_spawner h, uint32_t type, int32_t *mode, uint32_t *moved_bytes,
                               uint32_t *algorithmic_bytes) {
    SpawnerHost *sp = get_spawner(ctx, h);
    if (!sp || type >= sp->seg.size()) return FW_EINVAL;
    const SegHost &S = ctx->segs[sp->seg[type]];
    const TypeHost &T = sp->types[type];
    if (mode) *mode = S.fifo ? 1 : (S.range ? 2 : (S.small ? 3 : 0));
    const uint32_t colours = (T.base.kind != 0 ? 16u : 0u) + (T.emis.kind != 0 ? 16u : 0u);  // one-key gradients: never rewritten
    uint32_t moved, algo;
    if (S.ring()) {
        // in place: position+age and velocity always; rotation only where some emitter makes the particles spin (or the
        // type accelerates them), angular velocity only if it then changes; scale unless its curve is constant
        bool spins = false;
        for (const EmissionHost &E : sp->em)
            if ((uint32_t)E.es.particle_index == type)
                spins |= !(E.es.initial_angular_velocity.magnitude.min == 0.f && E.es.initial_angular_velocity.magnitude.max == 0.f);
        const float *aa = T.ps.angular_acceleration;
        const bool acc = aa[0] != 0.f || aa[1] != 0.f || aa[2] != 0.f;
        const uint32_t q2 = (spins || acc) ? 16u : 0u, q3 = ((spins && T.ps.angular_drag != 0.f) || acc) ? 16u : 0u;
        // (a type that cannot turn: neither the rotation nor the angular-velocity / lifetime plane is read)
        moved = (S.nospin ? 32u : 64u) + 32u + q2 + q3 + (T.scale.kind != 0 ? 4u : 0u) + colours;
        algo = moved - 4u - (q3 ? 4u : 0u);
        // (a range ring: lifetimes differ from particle to particle -- 4 B read for a type that cannot turn, Q3 otherwise --
        // and the scale depends on initial_scale; the part of the list that may lose particles, a fifth of configs[2], is
        // compacted in place and rewrites every plane it keeps: the figure is the young part's)
        if (S.range && S.nospin) moved += 4u, algo += 4u;
    } else {
        // compacting: every state plane lands at a new slot (+ the last_emitted planes of a Nested parent, read and written);
        // a type that cannot turn keeps no rotation plane: -16 B read, -16 B written,
        // ... and its lifetimes in a 4-byte plane instead of Q3: -32 B again, +4 B read, +4 B written
        moved = 64u + 64u + 4u + colours + 8u * S.n_lplanes - (S.nospin ? 32u + 32u - 8u : 0u);
        algo = moved - 8u;
    }
    if (S.derived) {  // scale and colour planes are not stored (the instance record carries them: +64 B written per particle)
        const uint32_t skipped = colours + ((S.ring() && T.scale.kind == 0) ? 0u : 4u);
        moved -= std::min(moved, skipped), algo -= std::min(algo, skipped);
    }
    // an attached instance buffer: the update also writes the 64-byte ParticleInstance record of every survivor (render.rs:95-103)
    if (S.inst != nullptr) moved += 64u, algo += 64u;
    if (moved_bytes) *moved_bytes = moved;
    if (algorithmic_bytes) *algorithmic_bytes = algo;
    return FW_OK;
}
// frames with Nested entries so far: those whose entries ran inside the FIFO launch (FwFifoNest) / those that ran the separate
// fw_k_spawn / fw_k_nest passes
fw_status fw_debug_nest_frames(fw_ctx *ctx, uint64_t *fused, uint64_t *separate) {
    if (!ctx) return FW_EINVAL;
    if (fused) *fused = ctx->fused_nest_frames;
    if (separate) *separate = ctx->nest_pass_frames;
    return FW_OK;
}
fw_status fw_debug_read_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles) {
    return fw_debug_read_timestamps2(ctx, out, nullptr, max_tiles, n_tiles);
}

fw_status fw_ctx_kernel_timing_overhead(fw_ctx *ctx, double *ms_per_pair) {
    if (!ctx || !ms_per_pair) return FW_EINVAL;
    *ms_per_pair = ctx->tev_overhead_ms;
    return FW_OK;
}

fw_status fw_ctx_measure_copy_bandwidth(fw_ctx *ctx, uint64_t bytes, int32_t iters, double *bytes_per_s) {
    if (!ctx || !bytes_per_s || bytes < 4096 || iters < 1) return FW_EINVAL;
    hipSetDevice(ctx->device);
    bytes &= ~(uint64_t)0xFFF;
    void *a = nullptr, *b = nullptr;
    FW_HIP(ctx, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) {
        hipFree(a);
        return fail(ctx, FW_ENOMEM, "copy probe allocation");
    }
    fw_memset_done(a, 1, bytes);
    fw_memset_done(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) fw_launch_copy_probe(ctx->stream, a, b, bytes);
    hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < iters; i++) fw_launch_copy_probe(ctx->stream, (i & 1) ? b : a, (i & 1) ? a : b, bytes);
    hipEventRecord(e1, ctx->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0), hipEventDestroy(e1);
    hipFree(a), hipFree(b);
    FW_HIP(ctx, e);
    *bytes_per_s = 2.0 * (double)bytes * iters / (ms * 1e-3);
    return FW_OK;
}

}  // extern "C"

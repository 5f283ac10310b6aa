// sharded.cpp -- native multi-GPU host of the particle path (BASELINE configs[4]) over the C++ mirror
// include/firework.hpp: what a Rust / C++ engine would do, with no Python in sight.
//
//   * spawners are independent units (nothing in core.rs:367-670 reads another spawner's state; the reference itself
//     relies on that for par_iter_mut, core.rs:583-585): emitter e lives on GPU e mod N, one context per GPU, no
//     particle ever crosses GPUs;
//   * the only exchange is the sum of live-particle counts: every update kernel leaves its frame's total in a DEVICE
//     ring registered with fw_ctx_live_count_ring (no extra launch, no host synchronisation), and every --reduce-every
//     frames ONE ncclAllReduce (RCCL over xGMI) carries the whole bucket of per-frame totals, enqueued on the context's
//     stream behind the frames that produced it -- a few bytes, latency-bound, never between two update kernels;
//   * RNG streams are keyed by the GLOBAL emitter index (uid), so results do not depend on the number of GPUs.
//
// Two ways to run it:
//   ./examples/sharded --gpus N [...]                               ONE process drives N devices (ncclCommInitAll)
//   ./examples/sharded --rank r --world N --id-file /path [...]     one process per GPU (rank 0 writes the ncclUniqueId
//                                                                   to the file, the others read it: ncclCommInitRank)
//   ./examples/sharded --ranks-on-one-device N [...]                REHEARSAL on one GPU (round 6: no multi-GPU node has been
//                                                                   available to any round): N contexts on device 0, one host
//                                                                   thread each, emitter e in context e mod N -- the share,
//                                                                   the streams, the live-count rings and the bucketing of the
//                                                                   N-rank run, with a host-side sum of the buckets standing in
//                                                                   for ncclAllReduce (RCCL refuses two ranks on one device).
//                                                                   Prints each context's steady-state us per frame.
//   common: --emitters E (4096) --live L (8192 per emitter) --frames F (96) --reduce-every K (16)
// Output (rank 0): one line per reduced frame "frame f global_live X", then a digest of the per-emitter counts of this
// process and the rate.  tests/test_cpp_host.py compares these lines with bevy_firework_amd.sharding on the same workload.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "firework.hpp"

using namespace firework;

#define HIPCHECK(x)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
            std::exit(1);                                                                        \
        }                                                                                        \
    } while (0)
#define NCCLCHECK(x)                                                                             \
    do {                                                                                         \
        ncclResult_t r_ = (x);                                                                   \
        if (r_ != ncclSuccess) {                                                                 \
            std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                        \
            std::exit(1);                                                                        \
        }                                                                                        \
    } while (0)

// configs[2] / configs[4] of BASELINE.json: bevy_firework_amd/workloads.py many_emitters(), value for value (the Python
// side computes in double and stores floats: so does this)
static ParticleSpawner emitter_settings(int e, double live_per_emitter) {
    const int k = e % 7;
    ParticleSpawner sp;
    ParticleSettings &ps = sp.particle_settings[0];
    ps.lifetime = {0.8f, 1.2f};
    ps.initial_scale = {0.02f, 0.06f};
    ps.scale_curve = FireworkCurve::even_samples({1.0f, (float)(1.5 + 0.1 * k), 0.2f});
    ps.acceleration = {(float)(0.1 * k), (float)(-9.81 + 0.5 * k), (float)(-0.05 * k)};
    ps.linear_drag = (float)(0.1 + 0.02 * k);
    ps.base_color = FireworkGradient::uneven_samples({{0.0f, {(float)(4.0 + k), 2.0f, (float)(0.5 * k), 1.0f}},
                                                      {(float)(0.5 + 0.05 * k), {1.0f, (float)(0.5 + 0.1 * k), 0.2f, 1.0f}},
                                                      {1.0f, {0.1f, 0.1f, 0.1f, 0.0f}}});
    ps.emissive_color = FireworkGradient::even_samples({{2.0f, (float)(1.0 + 0.1 * k), 0.0f, 1.0f}, {0.0f, 0.0f, 0.0f, 1.0f}});
    EmissionSettings &es = sp.emission_settings[0];
    es.emission_pacing = EmissionPacing::rate((float)live_per_emitter);  // mean lifetime 1.0 s
    es.emission_shape = EmissionShape::Sphere(1.0f);
    es.initial_velocity = RandVec3::constant({0, 0, 0});
    es.initial_velocity_radial = {1.0f, 4.0f};
    return sp;
}

struct Shard {  // one GPU: its context, its share of the emitters, its feed of the exchange
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t cstream = nullptr;  // the collective's stream: behind the bucket copy, never in front of the next frame (round 6)
    hipEvent_t ev_bucket = nullptr;
    std::unique_ptr<ParticleSystemPlugin> app;
    std::vector<ParticleSpawnerData *> emitters;
    std::vector<int> global_index;
    unsigned long long *ring = nullptr;     // device: per-frame live totals, written by the update kernels
    unsigned long long *buckets = nullptr;  // device: reduced buckets, one after the other
    ncclComm_t comm = nullptr;
};

int main(int argc, char **argv) {
    int gpus = 1, rank = -1, world = 0, emitters = 4096, frames = 96, every = 16, one_device = 0;
    double live = 8192.0;
    std::string id_file;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char *name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("--gpus")) gpus = atoi(argv[++i]);
        else if (arg("--rank")) rank = atoi(argv[++i]);
        else if (arg("--ranks-on-one-device")) one_device = atoi(argv[++i]);
        else if (arg("--world")) world = atoi(argv[++i]);
        else if (arg("--id-file")) id_file = argv[++i];
        else if (arg("--emitters")) emitters = atoi(argv[++i]);
        else if (arg("--live")) live = atof(argv[++i]);
        else if (arg("--frames")) frames = atoi(argv[++i]);
        else if (arg("--reduce-every")) every = atoi(argv[++i]);
        else {
            std::fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    const bool multi_process = rank >= 0;
    if (multi_process && (world < 1 || rank >= world || id_file.empty())) {
        std::fprintf(stderr, "--rank needs --world N and --id-file PATH\n");
        return 2;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        std::fprintf(stderr, "no HIP device available; this backend has no CPU fallback\n");
        return 1;
    }
    if (one_device > 0 && (multi_process || gpus != 1)) {
        std::fprintf(stderr, "--ranks-on-one-device excludes --gpus / --rank\n");
        return 2;
    }
    if (one_device > 0) gpus = one_device;
    const int n_ranks = multi_process ? world : gpus;       // GPUs the emitters are spread over
    const int n_local = multi_process ? 1 : gpus;           // ... of which this process drives
    if (!multi_process && one_device == 0 && gpus > visible) {
        std::fprintf(stderr, "--gpus %d but only %d devices are visible: refusing to run fewer ranks\n", gpus, visible);
        return 1;
    }
    every = std::max(1, every);
    const int n_buckets = (frames + every - 1) / every;
    try {
        std::vector<Shard> shards(n_local);
        // ---- RCCL communicators: one per GPU
        if (multi_process) {
            ncclUniqueId id;
            if (rank == 0) {
                NCCLCHECK(ncclGetUniqueId(&id));
                FILE *f = fopen((id_file + ".tmp").c_str(), "wb");
                if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 1;
                fclose(f);
                rename((id_file + ".tmp").c_str(), id_file.c_str());
            } else {
                FILE *f = nullptr;
                for (int tries = 0; tries < 600 && !(f = fopen(id_file.c_str(), "rb")); tries++)
                    std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (!f || fread(&id, sizeof id, 1, f) != 1) return 1;
                fclose(f);
            }
            shards[0].device = rank % visible;
            HIPCHECK(hipSetDevice(shards[0].device));
            NCCLCHECK(ncclCommInitRank(&shards[0].comm, world, id, rank));
        } else if (one_device > 0) {
            for (int d = 0; d < n_local; d++) shards[d].device = 0;  // no communicator: the buckets are summed on the host
        } else {
            std::vector<int> devs(n_local);
            std::vector<ncclComm_t> comms(n_local);
            for (int d = 0; d < n_local; d++) devs[d] = d;
            NCCLCHECK(ncclCommInitAll(comms.data(), n_local, devs.data()));
            for (int d = 0; d < n_local; d++) shards[d].device = d, shards[d].comm = comms[d];
        }
        // ---- contexts and emitters: emitter e on GPU e mod N (bevy_firework_amd/sharding.py: owner_rank)
        const int side = std::max(1, (int)std::ceil(std::sqrt((double)emitters)));
        for (int l = 0; l < n_local; l++) {
            Shard &S = shards[l];
            const int r = multi_process ? rank : l;
            HIPCHECK(hipSetDevice(S.device));
            HIPCHECK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
            HIPCHECK(hipStreamCreateWithFlags(&S.cstream, hipStreamNonBlocking));
            HIPCHECK(hipEventCreateWithFlags(&S.ev_bucket, hipEventDisableTiming));
            S.app = std::make_unique<ParticleSystemPlugin>(S.device, 0x00C0FFEE, S.stream);
            for (int e = r; e < emitters; e += n_ranks) {
                const Transform tf{{(float)(3.0 * (e % side)), 0.0f, (float)(3.0 * (e / side))}, {}};
                S.emitters.push_back(S.app->spawn(emitter_settings(e, live), tf, (uint32_t)e));
                S.global_index.push_back(e);
            }
            HIPCHECK(hipMalloc((void **)&S.ring, 2 * every * sizeof(unsigned long long)));
            HIPCHECK(hipMalloc((void **)&S.buckets, (size_t)n_buckets * every * sizeof(unsigned long long)));
            HIPCHECK(hipMemset(S.buckets, 0, (size_t)n_buckets * every * sizeof(unsigned long long)));
            HIPCHECK(hipDeviceSynchronize());
            S.app->check(fw_ctx_live_count_ring(S.app->raw(), S.ring, 2 * (uint32_t)every));
        }
        // ---- frames.  Nothing here waits for a GPU: fw_step enqueues, the bucket copy and the collective are enqueued on
        // the same stream behind the frames that wrote the bucket.
        const float dt = 1.0f / 60.0f;
        if (one_device > 0) {
            // ---- the rehearsal: every context is driven by its own host thread, as N ranks would be by N processes; a bucket of
            // per-frame totals leaves each context as ONE copy into pinned memory behind the frames that wrote it
            std::vector<unsigned long long *> h_b(n_local, nullptr);
            std::vector<double> steady_us(n_local, 0.0), all_us(n_local, 0.0);
            for (int l = 0; l < n_local; l++) {
                HIPCHECK(hipHostMalloc((void **)&h_b[l], (size_t)n_buckets * every * sizeof(unsigned long long), hipHostMallocDefault));
                memset(h_b[l], 0, (size_t)n_buckets * every * sizeof(unsigned long long));
            }
            const int half = frames / 2;
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            std::vector<int> failed(n_local, 0);
            std::atomic<int> arrived{0};
            for (int l = 0; l < n_local; l++)
                th.emplace_back([&, l]() {
                    try {
                        Shard &S = shards[l];
                        HIPCHECK(hipSetDevice(0));
                        const auto a0 = std::chrono::steady_clock::now();
                        auto a1 = a0;
                        int sent = 0;
                        auto flush = [&](int first, int n) {
                            HIPCHECK(hipMemcpyAsync(h_b[l] + first, S.ring + first % (2 * every), (size_t)n * sizeof(unsigned long long),
                                                    hipMemcpyDeviceToHost, S.stream));
                        };
                        for (int f = 0; f < frames; f++) {
                            if (f == half) {  // the steady half is timed on its own: wait for the fill, start the clock
                                HIPCHECK(hipStreamSynchronize(S.stream));
                                S.app->synchronize();
                                arrived.fetch_add(1);  // (every context starts its steady half together: the halves overlap in full)
                                while (arrived.load() < n_local) std::this_thread::yield();
                                a1 = std::chrono::steady_clock::now();
                            }
                            if (f == 0) S.app->update(dt);
                            else S.app->step(dt);
                            if (f + 1 - sent == every) flush(sent, every), sent = f + 1;
                        }
                        if (frames > sent) flush(sent, frames - sent);
                        S.app->synchronize();
                        HIPCHECK(hipStreamSynchronize(S.stream));
                        const auto a2 = std::chrono::steady_clock::now();
                        steady_us[l] = std::chrono::duration<double, std::micro>(a2 - a1).count() / std::max(1, frames - half);
                        all_us[l] = std::chrono::duration<double, std::micro>(a2 - a0).count() / frames;
                    } catch (const Error &e) {
                        std::fprintf(stderr, "context %d: firework error %d: %s\n", l, (int)e.status, e.what());
                        failed[l] = 1;
                        arrived.fetch_add(n_local);  // (nobody waits for a context that has given up)
                    }
                });
            for (auto &t : th) t.join();
            for (int l = 0; l < n_local; l++)
                if (failed[l]) return 1;
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            for (int f = 0; f < frames; f++) {  // the stand-in for ncclAllReduce(sum): live counts only, a few bytes per frame
                unsigned long long g = 0;
                for (int l = 0; l < n_local; l++) g += h_b[l][f];
                std::printf("frame %d global_live %llu\n", f, g);
            }
            uint64_t updated = 0, live_total = 0;
            double worst = 0.0;
            for (int l = 0; l < n_local; l++) {
                Shard &S = shards[l];
                uint64_t live_here = 0;
                for (auto *em : S.emitters) live_here += em->counts()[0];
                live_total += live_here;
                updated += S.app->updated_total();
                worst = std::max(worst, steady_us[l]);
                std::printf("context %d of %d on device 0: %zu emitters, %llu live, steady %.1f us per frame (frames %d..%d), %.1f us per frame over all\n",
                            l, n_local, S.emitters.size(), (unsigned long long)live_here, steady_us[l], half, frames - 1, all_us[l]);
            }
            std::printf("ranks_on_one_device %d emitters %d live_total %llu slowest_context_steady_us %.1f particles_per_s_all_contexts_steady %.3e\n",
                        n_local, emitters, (unsigned long long)live_total, worst, worst > 0 ? live_total / (worst * 1e-6) : 0.0);
            std::printf("particles updated/s (this process, incl. the fill): %.3e over %d frames, %.1f us per frame\n", updated / sec,
                        frames, sec / frames * 1e6);
            for (int l = 0; l < n_local; l++) {
                Shard &S = shards[l];
                S.app->check(fw_ctx_live_count_ring(S.app->raw(), nullptr, 0));
                S.app.reset();
                (void)hipFree(S.ring), (void)hipFree(S.buckets), (void)hipHostFree(h_b[l]);
                (void)hipStreamDestroy(S.stream), (void)hipStreamDestroy(S.cstream), (void)hipEventDestroy(S.ev_bucket);
            }
            return 0;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int sent = 0;
        auto reduce = [&](int first, int n) {
            const int lo = first % (2 * every), b = first / every;
            // the bucket leaves the ring IN ORDER with the frames (the ring's slots are rewritten 2 * every frames later); the collective
            // itself -- a few bytes, latency-bound -- runs on a stream of its own behind that copy: the next frames do not wait for it
            for (Shard &S : shards) {
                HIPCHECK(hipSetDevice(S.device));
                unsigned long long *dst = S.buckets + (size_t)b * every;
                HIPCHECK(hipMemcpyAsync(dst, S.ring + lo, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, S.stream));
                HIPCHECK(hipEventRecord(S.ev_bucket, S.stream));
                HIPCHECK(hipStreamWaitEvent(S.cstream, S.ev_bucket, 0));
            }
            NCCLCHECK(ncclGroupStart());
            for (Shard &S : shards) {
                HIPCHECK(hipSetDevice(S.device));
                unsigned long long *dst = S.buckets + (size_t)b * every;
                NCCLCHECK(ncclAllReduce(dst, dst, (size_t)n, ncclUint64, ncclSum, S.comm, S.cstream));  // live counts only
            }
            NCCLCHECK(ncclGroupEnd());
        };
        for (int f = 0; f < frames; f++) {
            for (Shard &S : shards) {
                HIPCHECK(hipSetDevice(S.device));
                if (f == 0) S.app->update(dt);  // the first frame also pushes the spawner transforms
                else S.app->step(dt);
            }
            if (f + 1 - sent == every) reduce(sent, every), sent = f + 1;
        }
        if (frames > sent) reduce(sent, frames - sent);  // a partial bucket
        uint64_t updated = 0;
        for (Shard &S : shards) {
            HIPCHECK(hipSetDevice(S.device));
            S.app->synchronize();
            HIPCHECK(hipStreamSynchronize(S.stream));
            HIPCHECK(hipStreamSynchronize(S.cstream));
            updated += S.app->updated_total();
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // ---- results (this is the only place anything is brought to the host)
        if (!multi_process || rank == 0) {
            std::vector<unsigned long long> h((size_t)n_buckets * every);
            HIPCHECK(hipSetDevice(shards[0].device));
            HIPCHECK(hipMemcpy(h.data(), shards[0].buckets, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int f = 0; f < frames; f++) std::printf("frame %d global_live %llu\n", f, h[f]);
        }
        unsigned long long digest = 1469598103934665603ull;  // FNV-1a over (emitter, count) of this process, emitter order
        uint64_t live_total = 0;
        std::vector<std::pair<int, uint32_t>> counts;
        for (Shard &S : shards) {
            HIPCHECK(hipSetDevice(S.device));
            for (size_t i = 0; i < S.emitters.size(); i++) counts.push_back({S.global_index[i], S.emitters[i]->counts()[0]});
        }
        std::sort(counts.begin(), counts.end());
        for (auto &c : counts) {
            live_total += c.second;
            const uint32_t w[2] = {(uint32_t)c.first, c.second};
            for (int j = 0; j < 8; j++) digest = (digest ^ ((const unsigned char *)w)[j]) * 1099511628211ull;
        }
        std::printf("ranks %d local_devices %d emitters_here %zu live_here %llu counts_digest %016llx\n", n_ranks, n_local,
                    counts.size(), (unsigned long long)live_total, digest);
        std::printf("particles updated/s (this process, incl. the fill): %.3e over %d frames, %.1f us per frame\n", updated / sec,
                    frames, sec / frames * 1e6);
        for (Shard &S : shards) {
            HIPCHECK(hipSetDevice(S.device));
            S.app->check(fw_ctx_live_count_ring(S.app->raw(), nullptr, 0));
            S.app.reset();
            (void)ncclCommDestroy(S.comm);
            (void)hipFree(S.ring), (void)hipFree(S.buckets);
            (void)hipStreamDestroy(S.stream), (void)hipStreamDestroy(S.cstream), (void)hipEventDestroy(S.ev_bucket);
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "firework error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
    return 0;
}

/*
 * firework_hip_debug.h -- measurement and debugging hooks of libfirework_hip.so.
 *
 * NOT part of the drop-in boundary (include/firework_hip.h is what a host binds): these entry points exist for bench.py,
 * the tools under tools/ and the tests -- kernel timing by events attached to the dispatches, the copy-bandwidth probe of
 * the measured roofline, in-kernel timestamps, and which update path a particle type is on.  They may change between
 * builds without an ABI version bump.
 *
 * Environment knobs.  The library reads A/B and debugging switches from the environment (FW_FIFO, FW_RANGE, FW_NOSPIN,
 * FW_FORECAST, FW_UPDATE_MODE, FW_DEBUG, ...; DESIGN.md section 7 lists them) ONLY when FW_ENABLE_KNOBS=1 is set: a
 * product process never changes behaviour because of a stray variable.  The tests and the tools set it.
 */
#ifndef FIREWORK_HIP_DEBUG_H
#define FIREWORK_HIP_DEBUG_H

#include "firework_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing of the dominant kernel on the context's stream: enable, run
 * steps, then read (sum of kernel durations in ms, number of launches).  The start / stop
 * events are attached to the update dispatch itself (hipExtLaunchKernel), so each pair
 * spans exactly the kernel's begin / end timestamps -- the duration rocprofv3 reports. */
fw_status fw_ctx_kernel_timing(fw_ctx *ctx, int32_t enable);
fw_status fw_ctx_kernel_timing_read(fw_ctx *ctx, double *ms_total, uint64_t *launches, uint64_t *particles);
/* cost of an empty hipEventRecord pair on the stream (informational; nothing is subtracted from the figure above) */
fw_status fw_ctx_kernel_timing_overhead(fw_ctx *ctx, double *ms_per_pair);
/* device-to-device copy bandwidth probe (bytes moved R+W per second) for the measured-roofline line */
fw_status fw_ctx_measure_copy_bandwidth(fw_ctx *ctx, uint64_t bytes, int32_t iters, double *bytes_per_s);

/* in-kernel timestamps of the update kernel when the context was created under FW_DEBUG=8 (tools/tile_timeline.py,
 * tools/launch_gaps.py): per tile {entry, after the count barrier, after the prefix, end, 4 more phase marks} of the
 * last launch (and of the one before it: `prev`); {~earliest start [64], latest end [64]} of the last 256 launches */
fw_status fw_debug_read_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles);
fw_status fw_debug_read_timestamps2(fw_ctx *ctx, unsigned long long *out, unsigned long long *prev, uint64_t max_tiles,
                                    uint64_t *n_tiles);
fw_status fw_debug_read_launches(fw_ctx *ctx, unsigned long long *out32768, uint32_t *epoch);
/* ... and of the last range-ring launch (tools/range_timeline.py): 8 words per workgroup {start, 0, 0, end of its wave 0,
 * role << 30 | k, segment, 0, 0} */
fw_status fw_debug_read_range_timestamps(fw_ctx *ctx, unsigned long long *out, uint64_t max_tiles, uint64_t *n_tiles);
/* which update path a particle type is on (1 = FIFO ring updated in place, 2 = range ring: young part in place, old part
 * compacted in place, 0 = general compacting path), the bytes one
 * update of a live particle moves on it, and how many of those are algorithmic (bench.py's roofline accounting) */
fw_status fw_debug_update_path(fw_ctx *ctx, fw_spawner spawner, uint32_t type, int32_t *mode, uint32_t *moved_bytes,
                               uint32_t *algorithmic_bytes);

/* frames with Nested entries stepped so far: those whose entries ran INSIDE the FIFO ring launch (one launch per frame, DESIGN.md
 * 4.0) and those that ran the separate fw_k_spawn / fw_k_nest passes first */
fw_status fw_debug_nest_frames(fw_ctx *ctx, uint64_t *fused, uint64_t *separate);

/* tiles of the compacting launch's tile table as of the last fw_step, and how many entries each per-tile array of the context (status
 * words, forecast entries, tile boxes) holds: the first never exceeds the second (fw_step refuses the frame otherwise) */
fw_status fw_debug_tile_scratch(fw_ctx *ctx, uint64_t *table_tiles, uint64_t *scratch_tiles);

/* rings that fw_step moved to the compacting path -- particles and order kept -- because the device's report of a cohort size was
 * still missing when the cohort was due (a failed check of the host's bookkeeping that is found BEFORE the frame is committed:
 * recoverable) */
fw_status fw_debug_recovered_rings(fw_ctx *ctx, uint64_t *n);

/* frames of the compacting launch that ran under a dt different from the previous frame's on the STREAMING schedule (threshold
 * forecast: fw_k_fc_resolve in front of fw_k_update_stream) instead of the decoupled look-back */
fw_status fw_debug_tf_frames(fw_ctx *ctx, uint64_t *n);

/* *on = 1: the context keeps the per-frame records of its range launches and its small op tables in DEVICE memory that the host writes
 * through the large BAR; 0: in pinned host memory (the platform does not map device memory for the host, or
 * FW_PARAM_BAR=0) */
fw_status fw_debug_param_bar(fw_ctx *ctx, int32_t *on);

#ifdef __cplusplus
}
#endif
#endif /* FIREWORK_HIP_DEBUG_H */

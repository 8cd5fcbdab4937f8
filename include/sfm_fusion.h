/*
 * sfm_fusion.h -- C ABI of the MI355X dense-fusion library (libsfm_fusion.so).
 *
 * Drop-in boundary for ONE stage of CarterDiOrio/SfM: the dense fusion that the reference
 * runs as
 *     sfm::DenseReconstruction dense_reconstruction{map};      (src/reconstruct.cpp:72)
 *     dense_reconstruction.reconstruct(dense_file);            (src/reconstruct.cpp:73)
 * The reference has no plugin/FFI boundary of its own; this header is what a C++ facade
 * with the reference's two-call shape (sfm_amd/cpp/dense_reconstruction.hpp) binds.
 * Citations are relative to /root/reference/reconstruction.
 *
 * Conventions
 *   - plain C types, no exceptions cross the ABI; every call returns an int status
 *     (SFM_OK or a negative SFM_E_*), sfm_fusion_last_error() gives the text.
 *   - 4x4 matrices are 16 doubles, COLUMN-major, i.e. Eigen::Matrix4d::data().
 *   - images: depth = CV_16UC1 rows, colour = CV_8UC3 rows in B,G,R order, with a row
 *     stride in bytes (cv::Mat::step).  The library COPIES pixels into its own HBM buffers
 *     inside sfm_fusion_add_keyframe; the caller's buffers may be released on return.
 *   - a ctx is driven by one host thread at a time; distinct ctxs are independent.
 *   - all geometry is in RAW depth units, exactly like the reference.
 */
#ifndef SFM_FUSION_H
#define SFM_FUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_OK 0
#define SFM_E_INVALID (-1)     /* bad argument / bad call order */
#define SFM_E_HIP (-2)         /* HIP runtime error (text in last_error) */
#define SFM_E_TABLE_FULL (-3)  /* voxel table capacity exhausted */
#define SFM_E_KEY_RANGE (-4)   /* the cloud's voxel indices need more than 63 bits in all (see sfm_fusion_get_key_range) */
#define SFM_E_CAPACITY (-5)    /* accepted-point list capacity exhausted */
#define SFM_E_UNSUPPORTED (-6) /* e.g. more than 4095 filtered neighbours of one keyframe */
#define SFM_E_NOMEM (-7)

#define SFM_MEM_HOST 0   /* pointers are host memory   */
#define SFM_MEM_DEVICE 1 /* pointers are device memory on the ctx's GPU */

#define SFM_MAX_NEIGHBORS 32 /* filtered neighbours per keyframe that the fused kernels hold on chip; a keyframe with
                              * more (the reference has no cap, src/map.cpp:152-160) sends the run through the
                              * two-phase chain of sfm_fusion_precompute, whose neighbours are a loop: same cloud */

typedef struct sfm_fusion_ctx sfm_fusion_ctx;

/* Fusion constants.  Defaults = the reference's compile-time values:
 * include/reconstruction/dense_reconstruction.hpp:44-56, src/dense_reconstruction.cpp:23-25,
 * :85 (0.30), :229 (0.50), :309 (1000). */
typedef struct sfm_fusion_params {
    double base_line;       /* 0.50 */
    double T_cov;           /* 0.5  */
    double voxel_size;      /* 0.05 */
    double T_dist;          /* 0.50 */
    double min_dist_filter; /* 0.0  */
    double dir_dot;         /* 0.30 */
    double S[3];            /* diag(0.05, 0.05, 10.0) */
    double depth_div;       /* 1000 */
    double close_frac;      /* 0.50 */
    uint64_t table_capacity; /* voxel-table slots (rounded up to 2^n); 0 = size automatically */
    uint64_t max_points;     /* accepted-point list capacity; 0 = size automatically */
} sfm_fusion_params;

/* The four progress counts the reference logs (src/dense_reconstruction.cpp:56-58,
 * :118-120) plus traffic counters used for the roofline's algorithmic byte count. */
typedef struct sfm_fusion_stats {
    uint64_t n_frames;
    uint64_t global_before; /* "Points before voxel removal" */
    uint64_t global_after;  /* "Points after voxel removal"  */
    uint64_t n_px;          /* pixels of all keyframes */
    uint64_t n_gather;      /* neighbour depth reads (visible projections) */
    uint64_t n_acc;         /* accepted pixels = sum of per-frame "before" counts */
    uint64_t n_hit;         /* visibility-mask stores the reference performs */
    uint64_t n_vox;         /* output points */
    uint64_t table_capacity;
    uint64_t table_used;
    uint64_t max_probe;     /* longest probe sequence seen by an insert */
    double last_run_ms;     /* device time of the last sfm_fusion_run (HIP events) */
    double frame_kernels_ms;/* device time between first and last per-frame kernel */
    /* filled when sfm_fusion_set_profiling(ctx,1): per-dispatch start/stop timestamps of the
     * per-keyframe kernel (same clock rocprofv3 --kernel-trace reads) */
    uint64_t fuse_kernel_launches;   /* dispatches of the keyframe-chain kernel: fuse_chain_kernel (one per
                                      * group of 32 keyframes) or observe_frame_kernel (one per keyframe) */
    double fuse_kernel_sum_ms;       /* sum of their durations */
    double scan_kernel_sum_ms;       /* sum of the scan_frame_kernel durations (0 with fuse_chain_kernel) */
    uint64_t fused_chain;            /* 1: the persistent fuse_chain_kernel ran; 0: the scan+observe pair */
    uint64_t precomputed;            /* 1: the run used the verdicts of sfm_fusion_precompute (two-phase chain) */
    double precompute_ms;            /* device time of the last sfm_fusion_precompute (phase A) when the run used it */
    uint64_t chain_fallbacks;        /* runs of this ctx that a device-wide barrier timeout (CUs held by someone
                                      * else) sent from fuse_chain_kernel to the kernel pair; the next 8 runs stay
                                      * on the pair, the third fallback is permanent */
} sfm_fusion_stats;

void sfm_fusion_default_params(sfm_fusion_params *p);

/* Replaces `DenseReconstruction(sfm::Map&)` (dense_reconstruction.cpp:20-26).
 * device = HIP device ordinal.  `stream` = a hipStream_t to run on, or NULL for a private one. */
int sfm_fusion_create(const sfm_fusion_params *params, int device, void *stream,
                      sfm_fusion_ctx **out);
void sfm_fusion_destroy(sfm_fusion_ctx *ctx);

/* One selected keyframe (reconstruct() selection rule :32-38 stays in the facade).
 * order_idx = position in processing order (the reference iterates a std::set, :48);
 * indices must be added as 0,1,2,... .  Reads what the reference reads through
 * KeyFrame::img / get_depth_image() / camera_calibration() / transform() /
 * world_to_camera() (src/keyframe.cpp:31-54).  K = {fx, fy, cx, cy}. */
int sfm_fusion_add_keyframe(sfm_fusion_ctx *ctx, uint32_t order_idx, int32_t W, int32_t H,
                            const uint16_t *depth, size_t depth_stride_bytes,
                            const uint8_t *bgr, size_t bgr_stride_bytes,
                            const double K[4], const double T_wk[16], const double T_kw[16],
                            int mem_kind);

/* The same keyframe with its pose straight from the bundle-adjustment result: the 6-vector that
 * Map::local_bundle_adjustment optimises, written back as
 *     Eigen::Matrix4d tf = Sophus::SE3<double>::exp(se3_vec).matrix();  kf->set_world_to_camera(tf);
 * (src/map.cpp:321-324; set_world_to_camera stores T_kw = tf and T_wk = tf.inverse(), src/keyframe.cpp:56-60).
 * se3_kw = (upsilon[3], omega[3]) in Sophus' tangent order.  sfm_fusion_se3_exp is the conversion by itself
 * (column-major 4x4 out; either output may be NULL).  Parity note: exp follows Sophus' published formulas; the
 * inverse is the rigid inverse [R^T | -R^T t], whereas the reference calls Eigen's general 4x4 inverse -- the two
 * agree to rounding, not bit for bit.  A caller that needs the reference's exact bits passes both matrices to
 * sfm_fusion_add_keyframe, as the facade does. */
int sfm_fusion_add_keyframe_se3(sfm_fusion_ctx *ctx, uint32_t order_idx, int32_t W, int32_t H,
                                const uint16_t *depth, size_t depth_stride_bytes,
                                const uint8_t *bgr, size_t bgr_stride_bytes,
                                const double K[4], const double se3_kw[6], int mem_kind);
void sfm_fusion_se3_exp(const double se3[6], double T[16], double T_inv[16]);

/* Map::get_neighbors(kf, 100) (src/map.cpp:152-160) as order indices, in the order the
 * reference would iterate them.  The distance/direction filter of process_frame
 * (src/dense_reconstruction.cpp:72-89) is applied inside the library. */
int sfm_fusion_set_neighbors(sfm_fusion_ctx *ctx, uint32_t order_idx,
                             const uint32_t *nbr_order_idx, uint32_t n);

/* Alternative to one sfm_fusion_set_neighbors call per keyframe: hand over the covisibility
 * graph itself -- `map.covisibility_edge` (include/reconstruction/map.hpp:143) as n_edges triples
 * (a, b, shared count) over the order indices of ALL keyframes of the map, added with
 * sfm_fusion_add_keyframe in map order -- and let the library apply the reference's rules:
 * a keyframe is selected iff it is an endpoint of an edge with count >= select_min (100,
 * src/dense_reconstruction.cpp:32-38); j is a neighbour of i iff count(i,j) > neighbor_min (100,
 * src/dense_reconstruction.cpp:70, src/map.cpp:157), neighbours in ascending index order.
 * Unselected keyframes produce nothing, are nobody's neighbour and do not count in n_px; tags
 * keep the order index the keyframe was added with.  Replaces any earlier neighbour lists. */
int sfm_fusion_set_covisibility(sfm_fusion_ctx *ctx, const uint32_t *edge_a, const uint32_t *edge_b,
                                const uint32_t *edge_count, uint64_t n_edges, uint32_t select_min,
                                uint32_t neighbor_min);

/* Replaces the body of `reconstruct(std::ostream&)` up to the writer loop (:40-58):
 * zeroes the visibility masks, fuses every keyframe in order, applies the per-frame and the
 * global first-wins voxel filter.  May be called repeatedly (each call starts from clean
 * masks); the result stays in HBM until the next run. */
int sfm_fusion_run(sfm_fusion_ctx *ctx);

int sfm_fusion_num_points(sfm_fusion_ctx *ctx, uint64_t *n);

/* Copies the cloud to host arrays, sorted by voxel key (k0,k1,k2 unsigned lexicographic).
 * xyz: N*3 doubles; bgr: N*3 int32 in B,G,R order (PointInfo::color); key: N*3 uint64
 * (the reference's std::tuple<size_t,size_t,size_t>, :270-274); tag: N uint64 =
 * (order_idx << 32) | (v*W + u) of the winning pixel.  Any pointer may be NULL. */
int sfm_fusion_read_points(sfm_fusion_ctx *ctx, double *xyz, int32_t *bgr, uint64_t *key,
                           uint64_t *tag);

int sfm_fusion_get_stats(sfm_fusion_ctx *ctx, sfm_fusion_stats *out);
/* Measurement aid (no reference counterpart): record start/stop events around every
 * per-keyframe kernel dispatch of subsequent runs. */
int sfm_fusion_set_profiling(sfm_fusion_ctx *ctx, int enable);
/* per-frame "Points before/after voxel filter" (:118-120); arrays of n_frames. */
int sfm_fusion_get_frame_counts(sfm_fusion_ctx *ctx, uint64_t *before, uint64_t *after);

/* ---- multi-GPU: one ctx per GPU, each fusing whole covisibility components ----------------
 * No reference counterpart (the reference is single-process).  Keyframes are sharded by
 * covisibility component (the visibility-mask chain never crosses components); the global
 * voxel filter (src/dense_reconstruction.cpp:57) becomes: bucket every GPU's winners by
 * owner = hash(voxel key) mod n_parts, all-to-all the buckets (RCCL, done by the caller),
 * min-reduce per owner.  The union of the owners' clouds is the reference's cloud. */
#define SFM_RECORD_BYTES 48 /* {u64 key, u64 tag, f64 x,y,z, u32 bgr, u32 pad} */

/* Global processing order of each of this ctx's keyframes (strictly increasing, one entry per
 * keyframe).  Exported records and sfm_fusion_read_points tags then carry global orders. */
int sfm_fusion_set_global_order(sfm_fusion_ctx *ctx, const uint32_t *order, uint32_t n);
/* The voxel key -- three size_t in the reference (:270-274) -- travels inside the library as ONE exact 64-bit word:
 * per axis the code of an index within a range [lo, hi] (or of x86's "indefinite" index), +-(2^20 - 1) and 21 bits
 * each by default.  A run with voxel indices outside the ranges (a cloud far from the origin, a very large one)
 * notices, scans the index range of its fused points in a second pass and runs a third time with exactly those
 * ranges; later runs of the ctx start with them.  Exact while the three code widths sum to <= 63 bits
 * (SFM_E_KEY_RANGE beyond: a cloud spanning more than 2^21 voxels in every one of the three axes at once).
 * sfm_fusion_read_points always returns the full three size_t.  Ranks that exchange records must pack alike:
 * the exchange takes the hull of all ranks' get_key_range and sets it everywhere (set_key_range re-encodes the
 * cloud at hand and fails, changing nothing, if that cloud has an index outside the new ranges). */
int sfm_fusion_get_key_range(sfm_fusion_ctx *ctx, int64_t lo[3], int64_t hi[3]);
int sfm_fusion_set_key_range(sfm_fusion_ctx *ctx, const int64_t lo[3], const int64_t hi[3]);
/* After sfm_fusion_run: per axis the smallest and largest voxel index that the cloud at hand really holds (the
 * indefinite index aside; lo > hi on an axis, or for a cloud, without any).  A packing's range is wider than the cloud
 * (the default holds +-(2^20 - 1) whatever the cloud): ranks whose packings differ build the common one from the hull
 * of these, not of their packings -- the hull of a default packing and a scanned one may need more than 63 bits where
 * the clouds' own hull does not.  No reference counterpart. */
int sfm_fusion_get_cloud_index_range(sfm_fusion_ctx *ctx, int64_t lo[3], int64_t hi[3]);
/* After sfm_fusion_run: how many local winners each of n_parts owners gets (the owner is a hash of the three voxel
 * indices, independent of the packing). */
int sfm_fusion_partition_counts(sfm_fusion_ctx *ctx, uint32_t n_parts, uint64_t *counts);
/* sfm_fusion_partition_counts for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc, a host thread beside a graph capture).  It still synchronises its stream and reads a pinned buffer on the
 * host, so the call itself cannot be captured into a HIP graph.  No counterpart in the reference.  Same result; the kernel's per-owner
 * counters live in a static __device__ array of the library (8 KiB on every GPU the library runs on, held until the
 * process ends) instead of a buffer the library allocates, and the call performs no hipMalloc / hipFree.  Limit:
 * n_parts <= 1024 (SFM_E_CAPACITY beyond).  Calls on one GPU share that array: calls that overlap on one GPU (two
 * ctxs, two threads) race.  Unlike sfm_fusion_partition_counts it does not arm sfm_fusion_export_records. */
int sfm_fusion_partition_counts_noalloc(sfm_fusion_ctx *ctx, uint32_t n_parts, uint64_t *counts);
/* Writes the local winners as records, grouped by owner in owner order, into device memory. */
int sfm_fusion_export_records(sfm_fusion_ctx *ctx, void *dst_dev, uint64_t dst_bytes);
/* Owner side: first-wins reduce over n received records (device memory); the result replaces
 * this ctx's cloud (sfm_fusion_num_points / sfm_fusion_read_points). */
int sfm_fusion_merge(sfm_fusion_ctx *ctx, const void *records_dev, uint64_t n);

/* ---- multi-GPU: ONE covisibility component (one sequence) over several GPUs, by keyframe blocks ----------
 * No reference counterpart.  The visibility masks (src/dense_reconstruction.cpp:100, :151-154) are the only
 * state that one keyframe hands to a later one, so the frame loop (:48-52) may be cut anywhere: a ctx fuses the
 * keyframes [first, end) only, starting from the masks that the blocks before it left, and hands the masks of
 * later keyframes on.  Every keyframe that a fused keyframe names as a neighbour must be added too (its depth
 * image and pose are read); keyframes outside the range produce nothing and do not count in n_px.
 * Call order: add keyframes, neighbours -> set_active_range -> import_masks (0 or more) -> run -> export_masks.
 * The union of the blocks' clouds after the voxel exchange above is bit-identical to one ctx fusing everything. */
int sfm_fusion_set_active_range(sfm_fusion_ctx *ctx, uint32_t first, uint32_t end);
/* Optional, for a rank that waits for its predecessors' masks: everything process_pixel computes except its first
 * test, the visibility mask (:100), is independent of the masks -- the covariance gate (:136-140), geometric_check
 * (:177-234), the pixels it would mark (:152-154), the fused point (:157-165).  This call computes that for every
 * pixel of the active keyframes (asynchronously, on the ctx's stream; 36 + 4 * later-neighbours bytes per pixel of
 * device memory); the sfm_fusion_run calls that follow only read masks, accept, mark and emit, keyframe after
 * keyframe.  Same cloud, same counts.  It re-observes every pixel, not only those the masks leave (about four
 * times the arithmetic): worth it only where it overlaps another rank's chain.  The verdicts stay valid until
 * keyframes, neighbours or the active range change.  Where the device-wide barrier is unavailable the runs that follow
 * launch the accept step keyframe by keyframe instead. */
int sfm_fusion_precompute(sfm_fusion_ctx *ctx);
/* ORs masks into the state the next runs start from: keyframes [first, first+count), W*H bytes each, packed back
 * to back, device memory.  Dropped by set_active_range, add_keyframe, set_neighbors and clear_imported_masks. */
int sfm_fusion_import_masks(sfm_fusion_ctx *ctx, uint32_t first, uint32_t count, const void *src_dev, uint64_t bytes);
/* Drops the imports.  It also zeroes the mask arena right away (asynchronously, where the layout stands): the next run
 * then starts from that arena as it is, and the imports made in between go straight into it -- a rank that calls this
 * before it starts waiting has nothing but the imports themselves between its predecessors' masks and its chain.
 * Imports still last until the next clear: a second run without one starts from the same masks. */
int sfm_fusion_clear_imported_masks(sfm_fusion_ctx *ctx);
/* After sfm_fusion_run (or sfm_fusion_run_chain): the masks of keyframes [first, first+count) as the run left them,
 * same packing. */
int sfm_fusion_export_masks(sfm_fusion_ctx *ctx, uint32_t first, uint32_t count, void *dst_dev, uint64_t bytes);
/* sfm_fusion_run in two halves, for a rank that has successors waiting.  The frame loop (:48-52) and the voxel filter
 * behind it (:54-58) run on two streams; the masks are final when the frame loop ends, while the filter of the last
 * keyframe groups still runs (on the 2000-keyframe workload the filter is the longer of the two: 38 ms against 22).
 * sfm_fusion_run_chain launches the whole run and returns when the frame loop has finished: export_masks is allowed
 * from then on, nothing else that reads the result is.  sfm_fusion_finish waits for the rest and completes the run
 * exactly as sfm_fusion_run would have (same retries, same stats); sfm_fusion_run does the same.  Between the two
 * calls every call that changes what a run computes (add_keyframe, set_neighbors, set_covisibility, set_active_range,
 * import_masks, clear_imported_masks, precompute, set_key_range, set_global_order, merge) returns SFM_E_INVALID.
 * A retry that sfm_fusion_finish has to make (list, table or key packing too small) recomputes the same masks.
 * No reference counterpart. */
int sfm_fusion_run_chain(sfm_fusion_ctx *ctx);
int sfm_fusion_finish(sfm_fusion_ctx *ctx);

const char *sfm_fusion_last_error(sfm_fusion_ctx *ctx);
const char *sfm_fusion_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SFM_FUSION_H */

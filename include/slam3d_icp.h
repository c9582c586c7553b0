/*
 * slam3d_icp.h -- C-ABI of the MI355X-native plane-ICP registration path.
 *
 * This is the drop-in boundary for the per-frame alignment of gaoxiang12/slam3d_gx:
 * the reference's seam is the virtual method
 *     RESULT_OF_MULTIPNP GraphicEnd::multiPnP(vector<PLANE>&, vector<PLANE>&, bool, int, int)
 *         src/GraphicEnd.h:134, src/GraphicEnd.cpp:557-659   (callers :168,:187,:195,:700,:736,:813,:892)
 * and the frame producer GraphicEnd::readimage() src/GraphicEnd.cpp:266-302.  A subclass in the
 * style of GraphicEnd2 (src/GraphicEnd.h:262-275) overrides those two and forwards to the entry
 * points below (INTEGRATION.md shows the binding).  Plain C structs, fixed-width integers,
 * caller-owned host memory, no C++/torch types.
 *
 * Conventions kept from the reference:
 *   - pose direction: X_target(present) = T * X_source(keyframe)        (src/GraphicEnd.cpp:169-170,589-590)
 *   - norm = |min(angle, 2pi-angle)| + 0.9*||t||                        (src/GraphicEnd.cpp:618)
 *   - failure is signalled by T == Identity                             (src/GraphicEnd.cpp:173)
 *   - thresholds minimum_inliers / error_threshold                      (src/GraphicEnd.cpp:599,:621)
 *   - never throws; returns int status
 * T is ROW-MAJOR double[16] here (Eigen::Isometry3d of RESULT_OF_MULTIPNP, src/GraphicEnd.h:59-69,
 * is column-major: transpose when copying into .matrix().data()).
 */
#ifndef SLAM3D_ICP_H
#define SLAM3D_ICP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM3D_ICP_ABI_VERSION 8
#define SLAM3D_ICP_NSUMS 29   /* the sums of the trace: 21 upper-tri AtA + 6 Atb + count + sum r^2, derived from the Gram totals */
#define SLAM3D_ICP_NRAW  36   /* what the dense mode exchanges: the upper triangle of the 8x8 integer Gram matrix of the quantised row vectors (DESIGN.md spec S4) */

/* return codes: 0 ok; >0 algorithmic (result.T == Identity); <0 usage / runtime errors */
enum {
    SLAM3D_OK = 0,
    SLAM3D_TOO_FEW_INLIERS = 1,     /* inliers < min_inliers        src/GraphicEnd.cpp:599 */
    SLAM3D_NORM_EXCEEDED = 2,       /* norm > error_threshold       src/GraphicEnd.cpp:621 */
    SLAM3D_DEGENERATE = 3,          /* normal equations needed damping (planes do not span R^3) */
    SLAM3D_E_INVALID = -1,          /* bad argument */
    SLAM3D_E_HIP = -2,              /* HIP runtime error (see slam3d_last_error) */
    SLAM3D_E_NOMEM = -3,
    SLAM3D_E_NODEVICE = -4,         /* no gfx950 device visible */
    SLAM3D_E_STATE = -5,            /* call order violated (e.g. fetch before run) */
    SLAM3D_E_COMM = -6              /* RCCL error / librccl not loadable (see slam3d_comm_last_error) */
};

/* Estimators.  POINT2PLANE: target normals from the 7x7 organized window (src/planarFeatures.cpp:88-136); SVD: point-to-point
 * (Kabsch).  PLANE (ABI 6, DESIGN.md spec S2p / S4p): plane-ICP proper -- the target frame's PLANES give the normals.  The frame is
 * segmented on the device exactly as slam3d_segment_planes does (the pcl::SACSegmentation loop of src/GraphicEnd.cpp:353-430, with
 * the parameters of slam3d_icp_set_seg_params), every pixel labelled with plane r takes the plane's least-squares normal
 * (a, b, c) (the fit of src/GraphicEnd.cpp:360-375, d >= 0 :383-387: toward the camera) -- SURVEY.md App. C2 "points take their
 * plane's normal" --, a pixel on no plane keeps its 7x7-window normal (or is no target at all with SLAM3D_PLANE_ONLY); rows,
 * search, solve and result are those of POINT2PLANE.  Role in the reference: planes are extracted per frame and the pose is
 * derived from plane-wise correspondences, src/GraphicEnd.cpp:158,168,557-659. */
enum { SLAM3D_EST_POINT2PLANE = 0, SLAM3D_EST_SVD = 1, SLAM3D_EST_PLANE = 2 };
/* slam3d_icp_params.plane_flags (SLAM3D_EST_PLANE only):
 *   PAIR_GATE   SURVEY.md 8(a) row a9, src/GraphicEnd.cpp:459-484,:572 (PnP only inside matched plane pairs): the source frame is
 *               segmented too, its planes are carried into the target frame by the run's initial pose (Identity without T_init)
 *               and associated with the target's planes as GraphicEnd::match does (nearest (a, b, c, d), exact); a correspondence
 *               is kept iff the target pixel's plane is the one associated with the source pixel's plane (pixels on no plane:
 *               clutter only matches clutter).  A rejected source has no correspondence in that iteration.
 *   ONLY        pixels on no plane are not targets (the literal per-plane variant of App. C2).  Off by default: two or three planes
 *               rarely constrain all six degrees of freedom (two of them are often parallel), the clutter between them does. */
enum { SLAM3D_PLANE_PAIR_GATE = 1, SLAM3D_PLANE_ONLY = 2 };
/* NN search variants: all return bit-identical correspondences.
 *   BRUTE_VALU  every source x every target, LDS-tiled, fp32 VALU
 *   BRUTE_MFMA  same scan, the distance step as a dense contraction on the matrix cores used as a conservative
 *               filter (bf16 MFMA on exact three-way bf16 splits of every float; the f32 MFMA form with the
 *               environment variable SLAM3D_MFMA_BF16=0), flagged pairs re-evaluated in canonical fp32
 *   TILES       exact search restricted to the 8x8-pixel target tiles whose bounding box can hold
 *               a winner (AABB culling against a per-query upper bound); AUTO selects this */
enum { SLAM3D_NN_AUTO = 0, SLAM3D_NN_BRUTE_VALU = 1, SLAM3D_NN_BRUTE_MFMA = 2, SLAM3D_NN_TILES = 3 };

typedef struct slam3d_icp_params {
    int32_t width, height;          /* organized cloud size (640x480 Kinect)                         */
    double  fx, fy, cx, cy;         /* pinhole intrinsics, parameters.yaml camera_* (:82-86)         */
    double  depth_factor;           /* parameters.yaml camera_factor                                 */
    double  z_filter;               /* validity 0 < z <= z_filter, parameters.yaml:65                */
    int32_t iterations;             /* icp_iterations (20); fixed count, no early exit               */
    double  max_corr_dist;          /* icp_max_corr_dist (0.10 m)                                    */
    int32_t estimator;              /* SLAM3D_EST_*                                                  */
    int32_t normal_window;          /* 7   (src/planarFeatures.cpp:92)                               */
    int32_t normal_min_inliers;     /* 41  (src/planarFeatures.cpp:128, "> 40")                      */
    double  normal_inlier_dist;     /* 0.01 (src/planarFeatures.cpp:123)                             */
    int32_t min_inliers;            /* 12  (multiPnP default, src/GraphicEnd.h:134)                  */
    double  error_threshold;        /* 1.0 (parameters.yaml:39)                                      */
    int32_t max_batch;              /* frame pairs resident per handle                               */
    int32_t device;                 /* HIP device ordinal                                            */
    int32_t nn_mode;                /* SLAM3D_NN_*                                                   */
    int32_t extra_frames;           /* resident frames beyond the 2*max_batch implicit ones (keyframes) */
    /* optional correspondence gates of the point-to-plane estimator, 0 = off:
     *   max_plane_residual2  drop a correspondence whose squared point-to-plane residual e*e exceeds it -- the
     *                        per-pixel test of src/GraphicEnd.cpp~:484-489 with parameters.yaml:45 min_error_plane (0.02);
     *   min_normal_cos       drop it when the source pixel has no normal or (R n_src).n_tgt is below it -- the
     *                        outlier-rejection role of solvePnPRansac's inlier subset, src/GraphicEnd.cpp:522-554.    */
    float   max_plane_residual2;
    float   min_normal_cos;
    /* icp_coarse_iterations (3): the first coarse_iterations iterations of a run -- never its last one -- take only the sources
     * of every fourth 8x8-pixel tile, (tile_x + 2 tile_y) mod 4 == 0 (DESIGN.md spec S4c).  The reference's per-frame call hands
     * multiPnP no initial guess (src/GraphicEnd.cpp:168): the first iterations run on a pose that is centimetres off, where a
     * quarter of the rows yields the same update and the searches are at their widest.  0 = every iteration uses every source. */
    int32_t coarse_iterations;
    int32_t plane_flags;            /* SLAM3D_PLANE_* (ABI 6; the padding word of ABI 5: 0 = neither)                    */
} slam3d_icp_params;

/* UNORGANIZED clouds (round 5): a handle created with height == 1 aligns point lists of up to `width` points -- the cloud readimage
 * produces (PCD -> PassThrough -> VoxelGrid: 16,034 / 14,758 points for the reference's data/exp1 frames, src/GraphicEnd.cpp:283-295)
 * and hands to the plane extraction (:158).  Views may then carry any width <= params.width (height 1); records beyond a view are
 * invalid.  No camera model is assumed (fx .. cy unused; |x|, |y| <= z for the range check), there are no image windows or tiles:
 * the estimator is SLAM3D_EST_SVD or SLAM3D_EST_PLANE with SLAM3D_PLANE_ONLY (SLAM3D_E_INVALID otherwise) and SLAM3D_NN_AUTO selects
 * one persistent launch per run (round 6: Morton-cell tiles, exact tile-pruned search; SLAM3D_NN_MFMA16 / _MFMA / _VALU keep the full scans).
 * Spec S4c's coarse iterations take the points whose index i satisfies (i / 8) mod 4 == 0.  The persistent launch needs all its blocks
 * resident (at most four such runs in flight per device and process): should a grid barrier not open within
 * about two seconds -- another process's persistent work on the device -- the run is given up, fetch / align return SLAM3D_E_HIP and the
 * pairs carry the failure convention (identity, SLAM3D_DEGENERATE); nothing hangs, the handle stays usable. */
/* a borrowed view of an organized cloud: `data` points at width*height records of
 * `stride_bytes` each whose first 12 bytes are float x,y,z (pcl::PointXYZRGBA is 32 B,
 * src/GraphicEnd.h:71-72; a packed float4 cloud is 16 B).  Invalid = NaN or z <= 0. */
typedef struct slam3d_cloud_view {
    const void *data;
    int32_t stride_bytes;
    int32_t width, height;
} slam3d_cloud_view;

/* mirrors RESULT_OF_MULTIPNP {T, norm, inliers} (src/GraphicEnd.h:59-69) + diagnostics */
typedef struct slam3d_icp_result {
    double  T[16];                  /* row-major; Identity unless status == SLAM3D_OK               */
    double  norm;
    int32_t inliers;
    int32_t status;                 /* SLAM3D_OK / TOO_FEW_INLIERS / NORM_EXCEEDED / DEGENERATE     */
    int32_t iterations;
    int32_t n_src, n_tgt;           /* valid source points / valid target points searched           */
    int32_t _pad;
    double  rmse;                   /* sqrt(sum r^2 / inliers) at the last iteration                */
    double  T_raw[16];              /* the converged estimate even when status != OK                */
} slam3d_icp_result;

/* plane of PLANE::coff (src/GraphicEnd.h:43): a,b,c,d with unit normal, d >= 0 (src/GraphicEnd.cpp:383-387).
 * (SURVEY.md row a3 sketched a `cov[6]` member as well.  It is deliberately absent: the reference's PLANE carries no covariance
 * (src/GraphicEnd.h:41-49), nothing on the path consumes one, and the fit's moments are integer fixed point inside the kernels --
 * exporting them as six floats would freeze an internal representation into the ABI for no reader.) */
typedef struct slam3d_plane {
    float   coeff[4];
    int32_t count;
    float   centroid[3];
} slam3d_plane;

typedef struct slam3d_icp_handle slam3d_icp_handle;

/* ---- threading contract ---------------------------------------------------------------------
 * A handle is used by ONE thread at a time (it is not internally locked).  DIFFERENT handles may be used from different threads
 * concurrently, on different devices or on the same one (GraphicEndICP::multiPnPBatch runs one thread per handle; the reference
 * itself is single-threaded and not re-entrant: process globals g_pParaReader / camera_*, src/ParameterReader.cpp:8-9).  Handles
 * hold no process-global mutable state; the handles of one DEVICE share exactly one word of library-owned device memory -- the
 * count of slam3d_icp_run runs in flight on that GPU, updated by device-side atomics only --, which shapes scheduling inside the
 * search kernel (poll another block's pose or solve locally) and never a result: any interleaving gives the same bits. */
/* ---- lifecycle --------------------------------------------------------------------------- */
void        slam3d_icp_default_params(slam3d_icp_params *p);
/* SLAM3D_E_INVALID also when width*height*(z_filter*sqrt(1+tan^2))^2 >= 2^28: the normal-equation totals (the integer Gram
 * matrix of DESIGN.md spec S4) and the plane moments are int64 and such a configuration could overflow them (640x480 at
 * z_filter 7 m is 11 times below); likewise when the farthest valid point is 90 m or more away, or the image has 2^22 pixels
 * or more (a wavefront's Gram sums must stay exact on the fp64 matrix cores).
 * A process may hold 256 handles at a time (each owns an entry of the search kernel's constant table); SLAM3D_E_NOMEM beyond. */
int         slam3d_icp_create(const slam3d_icp_params *p, slam3d_icp_handle **out);
void        slam3d_icp_destroy(slam3d_icp_handle *h);
const char *slam3d_strerror(int code);
const char *slam3d_last_error(const slam3d_icp_handle *h);   /* text of the last HIP failure */
int         slam3d_icp_abi_version(void);

/* ---- one-call pose API (replaces multiPnP, src/GraphicEnd.cpp:557-659) ------------------- */
/* source = keyframe cloud, target = present cloud; T_init nullable (Identity). */
int slam3d_icp_align(slam3d_icp_handle *h, const slam3d_cloud_view *src, const slam3d_cloud_view *tgt,
                     const double *T_init, slam3d_icp_result *out);
/* B independent pairs (loop-closure candidates, src/GraphicEnd.cpp:685-762, are such a batch). */
int slam3d_icp_align_batch(slam3d_icp_handle *h, int32_t B, const slam3d_cloud_view *src,
                           const slam3d_cloud_view *tgt, const double *T_init /* B*16 or NULL */,
                           slam3d_icp_result *out /* B */);
/* same, from raw 16-bit depth images (readimage + back-projection, src/GraphicEnd.cpp:266-302,
 * src/convert2PCD.cpp:54-72) */
int slam3d_icp_align_depth_batch(slam3d_icp_handle *h, int32_t B, const uint16_t *const *src_depth,
                                 const uint16_t *const *tgt_depth, const double *T_init,
                                 slam3d_icp_result *out);

/* ---- staged API (inputs resident in HBM; no host sync between set/run/fetch) ------------- */
int slam3d_icp_set_clouds_host(slam3d_icp_handle *h, int32_t slot, const slam3d_cloud_view *src,
                               const slam3d_cloud_view *tgt);
int slam3d_icp_set_depth_host(slam3d_icp_handle *h, int32_t slot, const uint16_t *src_depth,
                              const uint16_t *tgt_depth);
/* device pointers to organized float4 {x,y,z,-} clouds of width*height points; borrowed until
 * the next set_* on that slot. */
int slam3d_icp_set_clouds_device(slam3d_icp_handle *h, int32_t slot, const void *d_src_xyz4,
                                 const void *d_tgt_xyz4);
/* device pointers to u16 depth images; back-projected on the device into the handle's clouds. */
int slam3d_icp_set_depth_device(slam3d_icp_handle *h, int32_t slot, const void *d_src_depth,
                                const void *d_tgt_depth);
/* ---- resident frames ---------------------------------------------------------------------------------------
 * A frame is the resident unit: its organized cloud plus what the two ICP roles need of it (target: normals, tile
 * records, boxes; source: tile-major slots).  Those are built ONCE per frame and role, at the first run that uses the
 * frame that way after it was (re)set, and reused by every later pair: the keyframe of GraphicEnd::run
 * (src/GraphicEnd.cpp:168) stays the source of many consecutive pairs, and the loop-closure candidates of
 * src/GraphicEnd.cpp:685-762 all align against the same new keyframe.  Frame ids: slot b's implicit frames are 2b
 * (source) and 2b+1 (target) -- what set_clouds_* / set_depth_* above fill --, ids 2*max_batch .. 2*max_batch +
 * extra_frames - 1 are free for the caller (keyframe store).  All frame uploads and runs of one handle are ordered on
 * the handle's stream (or the stream given to run).
 * A frame set from a DEPTH image is back-projected by the library with the handle's intrinsics; as a target it also
 * serves the projective window search (same results, fewer candidates).  A frame set from a CLOUD is taken as it is:
 * no camera model is assumed for it, the tile search alone runs. */
int slam3d_icp_frame_count(const slam3d_icp_handle *h);
int slam3d_icp_frame_set_depth_host(slam3d_icp_handle *h, int32_t frame, const uint16_t *depth);      /* H2D + back-projection */
int slam3d_icp_frame_set_depth_device(slam3d_icp_handle *h, int32_t frame, const void *d_depth);
int slam3d_icp_frame_set_cloud_host(slam3d_icp_handle *h, int32_t frame, const slam3d_cloud_view *cloud);
int slam3d_icp_frame_set_cloud_device(slam3d_icp_handle *h, int32_t frame, const void *d_xyz4);       /* borrowed */
/* A frame's normals / tiles are built once per SET and cached (the same keyframe serves many pairs).  After rewriting a
 * BORROWED device buffer in place, call this (or set the frame again) before the next run; frames set from host memory or
 * depth images are copies and need nothing. */
int slam3d_icp_frame_invalidate(slam3d_icp_handle *h, int32_t frame);
/* pair `slot` = (source frame, target frame); the frames must have been set before the run */
int slam3d_icp_set_pair(slam3d_icp_handle *h, int32_t slot, int32_t src_frame, int32_t tgt_frame);

/* enqueue preprocessing of the frames that need it (normals, tiles) + `iterations` ICP iterations for slots
 * [0,B) on `stream` (hipStream_t, NULL = the handle's own stream).  Asynchronous. */
int slam3d_icp_run(slam3d_icp_handle *h, int32_t B, const double *T_init, void *stream);
/* wait for the run and produce results (norm / thresholds evaluated on the host). */
int slam3d_icp_fetch_results(slam3d_icp_handle *h, int32_t B, slam3d_icp_result *out);

/* ---- introspection for parity tests ------------------------------------------------------ */
/* correspondences of the LAST iteration: idx[N] original linear target index or -1, d2[N] (inf) */
int slam3d_icp_get_correspondences(slam3d_icp_handle *h, int32_t slot, int32_t *idx, float *d2);
/* T_trace[(iterations+1)*16], sums_trace[iterations*29] (either nullable) */
int slam3d_icp_get_trace(slam3d_icp_handle *h, int32_t slot, double *T_trace, double *sums_trace);
/* Correspondences of EVERY iteration ("index parity = memcmp of idx[N] per iteration"): opt-in, costs
 * one device copy of the slot-order indices per iteration and disables the captured graph.  Takes effect from the
 * next slam3d_icp_run; get_correspondences_at then returns iteration `it` of the last run (idx[N], pixel order). */
int slam3d_icp_set_corr_trace(slam3d_icp_handle *h, int32_t on);
int slam3d_icp_get_correspondences_at(slam3d_icp_handle *h, int32_t slot, int32_t it, int32_t *idx);
/* organized float4 clouds / target normals as the device holds them (each N*4 floats, nullable) */
int slam3d_icp_get_clouds(slam3d_icp_handle *h, int32_t slot, float *src_xyz4, float *tgt_xyz4,
                          float *tgt_nrm4);
/* Per-launch HIP events around every iteration's NN kernel (off by default: every event record serialises the
 * stream for ~6 us, ~15 % of a single-pair run).  Takes effect from the next slam3d_icp_run. */
int slam3d_icp_set_profiling(slam3d_icp_handle *h, int32_t on);
/* kernel time of the last run, by bucket (ms): [0] preprocess [1] nn [2] accumulate+solve [3] total;
 * without profiling [0] and [1] are 0 and [2] holds everything */
int slam3d_icp_get_timings(slam3d_icp_handle *h, float ms[4]);
/* duration of each iteration's NN launch of the last run (ms), nn_ms[iterations]; SLAM3D_E_STATE unless the
 * run was profiled */
int slam3d_icp_get_iteration_timings(slam3d_icp_handle *h, float *nn_ms);
/* Launch stamps (measurement aid, off by default; no reference counterpart): with ring_runs > 0 every NN and solve
 * launch of the following runs records when its first block started and its last wave ended on the device's
 * constant-rate 100 MHz real-time counter (one clock for all handles and streams of a GPU), with fire-and-forget
 * atomics into a device-resident ring of the last ring_runs runs -- unlike HIP events or a tracer this neither
 * serialises the streams nor copies anything while runs are in flight, so it shows how many launches of concurrent
 * handles are really resident at once.  ring_runs = 0 turns it off.
 * get_stamps: (start, end) ticks (10 ns) of the launches of the last runs, oldest first, out[run][2*iterations][2]; rows
 * [0, iterations) = NN launches, [iterations, 2*iterations) = solve launches (unused rows read (~0, 0)); *n_runs = runs
 * written (<= max_runs, <= ring_runs).  Waits for the handle's stream. */
int slam3d_icp_set_stamping(slam3d_icp_handle *h, int32_t ring_runs);
int slam3d_icp_get_stamps(slam3d_icp_handle *h, uint64_t *out, int32_t max_runs, int32_t *n_runs);
/* developer statistics of the LAST NN launch (slot 0), 20 int64 per source tile: clock at start / after
 * prologue / after the own scan / after the wide scan / at the end, tiles scanned, candidates, batches, where the
 * wave ran (HW_ID | XCC_ID << 32), its launch slot, real-time counter at start / end (100 MHz), clocks around the two block barriers, items drained.
 * Only available when the handle was created with SLAM3D_NN_DEBUG=1 in the environment. */
int slam3d_icp_get_nn_debug(slam3d_icp_handle *h, int64_t *out, int32_t n);

/* ---- building blocks (rows a5, a6 of the scope table) ------------------------------------ */
/* u16 depth (host) -> organized float4 cloud (host); src/convert2PCD.cpp:54-72 */
int slam3d_backproject_u16(slam3d_icp_handle *h, const uint16_t *depth, float *xyz4);
/* per-plane LS fit from labels (PCL's optimizeCoefficients inside SACSegmentation::segment,
 * src/GraphicEnd.cpp:360-375; sign rule :383-387).  labels[N] in {-1, 0..nplanes-1}. */
int slam3d_fit_planes(slam3d_icp_handle *h, const slam3d_cloud_view *cloud, const int32_t *labels,
                      int32_t nplanes, slam3d_plane *planes);

/* Plane association of GraphicEnd::match(vector<PLANE>&, vector<PLANE>&) (src/GraphicEnd.cpp:459-484): for every plane of
 * p1 the nearest plane of p2 by L2 distance on (a, b, c, d) -- what FlannBasedMatcher::match returns as
 * DMatch{queryIdx = i, trainIdx, distance}, computed exactly (<= 8 x 8 planes; ties -> lowest index).  Host code,
 * needs no device.  train_idx[i] = -1 when n2 == 0. */
int slam3d_match_planes(const slam3d_plane *p1, int32_t n1, const slam3d_plane *p2, int32_t n2, int32_t *train_idx,
                        float *distance);

/* Plane-association GATE built on the match above ("plane pairing gates correspondences
 * (optional)", the outlier-rejection role of src/GraphicEnd.cpp:522-554): the planes of frame 1 are carried into frame 2
 * by the estimated T (X_2 = T X_1: n' = R n, d' = d - n'.t, sign rule d' >= 0 of src/GraphicEnd.cpp:383-387), matched
 * to the planes of frame 2 exactly as GraphicEnd::match does, and a pair counts when its (a, b, c, d) distance is at
 * most max_dist.  *n_matched = number of such pairs.  A pose that maps no plane of frame 1 onto a plane of frame 2 is
 * geometrically unsupported however many point inliers it has.  Host code, needs no device. */
int slam3d_plane_gate(const slam3d_plane *p1, int32_t n1, const slam3d_plane *p2, int32_t n2, const double *T /* 16, row-major */,
                      float max_dist, int32_t *n_matched);
/* number of HIP devices visible (0 when none): lets a host front end shard a loop-closure batch, one handle per GPU */
int slam3d_device_count(void);

/* ---- frame ingestion filters of GraphicEnd::readimage (src/GraphicEnd.cpp:283-295): pcl::PassThrough on
 * z in [0, z_filter] followed by pcl::VoxelGrid with a cubic leaf (grid_leaf, 0.03), on n 16-byte records
 * {float x, y, z; uint32 rgba} -- the layout of the reference's binary PCD files (data/exp1/pcd/1.pcd header).
 * One output record per occupied voxel (centroid of all fields), ordered by voxel index (iz, iy, ix) like PCL.
 * n <= width*height of the handle; out has room for n records.
 * _device with a caller's stream: the call returns as soon as *n_out is known; the records in d_out16 are ready IN
 * STREAM ORDER (work queued on that stream afterwards sees them; synchronise the stream before touching them from
 * anywhere else).  With stream == NULL the handle's own stream is used and drained before the call returns. */
int slam3d_voxel_grid(slam3d_icp_handle *h, const void *points16, int32_t n, float leaf, void *out16, int32_t *n_out);
int slam3d_voxel_grid_device(slam3d_icp_handle *h, const void *d_points16, int32_t n, float leaf, void *d_out16,
                             int32_t *n_out, void *stream);
/* The same filters on B clouds in ONE launch sequence (grid.y = cloud): the keyframes saveOutput merges (src/saveOutput.cpp:58-96), the
 * loop-closure candidates of src/GraphicEnd.cpp:685-762.  d_points16 / d_out16: host arrays of B device pointers, n[b] <= width*height
 * records in cloud b, room for n[b] records in d_out16[b]; n_out[b] = voxels of cloud b.  Stream semantics as slam3d_voxel_grid_device.
 * The handle's voxel tables grow to B clouds at the first call that needs them (about 93 MB per 640x480 cloud). */
int slam3d_voxel_grid_batch_device(slam3d_icp_handle *h, int32_t B, const void *const *d_points16, const int32_t *n, float leaf,
                                   void *const *d_out16, int32_t *n_out, void *stream);
/* Measurement aid (no reference counterpart; ABI 8): how the voxel-grid calls of this handle were ordered so far.  counts[0] = calls whose
 * every cloud stayed inside the dense key range (three launches: insert, scan, finalize -- every cloud in a camera frame at the reference's
 * 3 cm leaf), counts[1] = calls in which a cloud left it (|ix| > 256 cells, iy or iz outside the row table) and took the general ordering
 * path as well.  The tables grow by 17 MB per cloud for the dense path's bitmaps. */
int slam3d_voxel_grid_path_counts(slam3d_icp_handle *h, int64_t counts[2]);
/* ---- keyframe cloud merge of saveOutput (src/saveOutput.cpp:47-103): per keyframe VoxelGrid alone (:80-83), then
 * PassThrough z in [0, pass_z] and pcl::transformPointCloud by the keyframe's pose (:84-92); the merged cloud goes
 * through VoxelGrid alone once more (:97-100).  Same 16-byte records.  pass_transform writes NaN for dropped
 * records (VoxelGrid ignores them), so no compaction is needed between the steps. */
int slam3d_voxel_grid_only(slam3d_icp_handle *h, const void *points16, int32_t n, float leaf, void *out16, int32_t *n_out);
int slam3d_pass_transform(slam3d_icp_handle *h, const void *points16, int32_t n, float z_max, const double *T /* 16, row-major */,
                          void *out16, int32_t *n_kept);

/* ---- plane segmentation (replaces the pcl::SACSegmentation loop of extractPlanesAndGenerateImage,
 * src/GraphicEnd.cpp:353-430): up to max_planes rounds of seeded RANSAC + least-squares refinement while more
 * than plane_percent of the valid points are unassigned (:372); planes in extraction order, unit normal, d >= 0
 * (:383-387).  labels[N]: -2 invalid pixel, -1 valid but on no plane, r = plane r. */
typedef struct slam3d_seg_params {
    float    distance_threshold;   /* parameters.yaml distance_threshold (0.08), :365 */
    float    plane_percent;        /* parameters.yaml plane_percent (0.2), :372 */
    int32_t  max_planes;           /* parameters.yaml max_planes (3), :424; <= 8 */
    int32_t  hypotheses;           /* RANSAC hypotheses per round, 1..64 (PCL's default budget is 50) */
    uint64_t seed;                 /* counter-based PRNG seed: same seed, same planes */
} slam3d_seg_params;
void slam3d_seg_default_params(slam3d_seg_params *sp);
/* one organized host cloud; planes[max_planes], labels nullable */
int slam3d_segment_planes(slam3d_icp_handle *h, const slam3d_cloud_view *cloud, const slam3d_seg_params *sp,
                          slam3d_plane *planes, int32_t *nplanes, int32_t *labels);
/* B organized float4 clouds resident on the device (d_clouds: host array of B device pointers);
 * planes[B*max_planes], nplanes[B] on the host, d_labels (device, B*N) nullable */
int slam3d_segment_planes_device(slam3d_icp_handle *h, int32_t B, const void *const *d_clouds,
                                 const slam3d_seg_params *sp, slam3d_plane *planes, int32_t *nplanes,
                                 int32_t *d_labels, void *stream);

/* ---- SLAM3D_EST_PLANE: the segmentation behind the normals ------------------------------------------------
 * Parameters of the per-frame segmentation (defaults = slam3d_seg_default_params: parameters.yaml distance_threshold 0.08,
 * plane_percent 0.2, max_planes 3; 64 hypotheses, seed 1).  Frames already built are rebuilt at the next run.
 * SLAM3D_E_STATE unless the handle's estimator is SLAM3D_EST_PLANE. */
int slam3d_icp_set_seg_params(slam3d_icp_handle *h, const slam3d_seg_params *sp);
/* the planes the library extracted for a resident frame at its last build (extraction order; planes[8]); *nplanes = 0 when the
 * frame was never built under SLAM3D_EST_PLANE.  Waits for the handle's streams.  GraphicEnd keeps them as _currKF.planes /
 * _present.planes (src/GraphicEnd.cpp:158,168): the front end reads them here instead of segmenting the frame a second time. */
int slam3d_icp_get_frame_planes(slam3d_icp_handle *h, int32_t frame, slam3d_plane *planes /* 8 */, int32_t *nplanes);
/* the association the pair gate of the LAST run used for pair `slot`: assoc[i] = target plane of source plane i, or -1; 8 entries */
int slam3d_icp_get_plane_assoc(slam3d_icp_handle *h, int32_t slot, int32_t *assoc /* 8 */);

/* ---- dense (single pair sharded over ranks) building blocks, one exchange per iteration -- */
/* restrict the source rows this handle works on to [row_begin,row_end) of slot 0 */
int slam3d_icp_dense_set_rows(slam3d_icp_handle *h, int32_t row_begin, int32_t row_end);
/* preprocess slot 0 (normals, compaction) and reset T to T_init */
int slam3d_icp_dense_begin(slam3d_icp_handle *h, const double *T_init, void *stream);
/* one NN + accumulate pass over the local rows: the 36 partial Gram totals on the host.  They are exact int64 sums of
 * products of integer row-vector components: integer addition is associative, so the all-reduced totals -- and with
 * them the pose -- do not depend on how the rows were sharded. */
int slam3d_icp_dense_partial(slam3d_icp_handle *h, int64_t sums[SLAM3D_ICP_NRAW], void *stream);
/* solve with the (all-reduced, integer SUM) sums and update T on every rank identically */
int slam3d_icp_dense_update(slam3d_icp_handle *h, const int64_t sums[SLAM3D_ICP_NRAW], void *stream);
int slam3d_icp_dense_finish(slam3d_icp_handle *h, const int64_t last_sums[SLAM3D_ICP_NRAW],
                            slam3d_icp_result *out);
/* the same three with the 36 totals in a caller-owned DEVICE buffer: partial writes it, the caller all-reduces it
 * in place ON THE SAME STREAM, update reads it.  No host synchronisation until finish.  `stream` must be the stream
 * the caller's collective is enqueued on: with more than one rank pass it explicitly -- NULL means the handle's own
 * non-blocking stream, which nothing outside the library is ordered with (slam3d_icp_dense_run does all of this
 * inside the library and is the form to use). */
int slam3d_icp_dense_partial_device(slam3d_icp_handle *h, int64_t *d_sums, void *stream);
int slam3d_icp_dense_update_device(slam3d_icp_handle *h, const int64_t *d_sums, void *stream);
int slam3d_icp_dense_finish_device(slam3d_icp_handle *h, const int64_t *d_last_sums, void *stream,
                                   slam3d_icp_result *out);


/* ---- multi-GPU: one process (or thread) per GPU, RCCL over xGMI behind the C-ABI ------------------------------
 * SURVEY.md 8(e).  The communicator wraps ncclCommInitRank; librccl is loaded at the first slam3d_comm_* call
 * (dlopen, reusing a copy the process already holds).  The 128-byte id is created on rank 0 and handed to the
 * other ranks by the host program (a file, MPI, torch.distributed's store: 128 bytes, once).  Every collective
 * below is enqueued on a HIP stream the library owns or on the handle's stream, in program order with the
 * kernels that produce / consume its buffer -- no ordering is left to the caller. */
typedef struct slam3d_comm slam3d_comm;
#define SLAM3D_COMM_ID_BYTES 128
int  slam3d_comm_get_unique_id(void *id /* SLAM3D_COMM_ID_BYTES */);
int  slam3d_comm_init(const void *id, int32_t rank, int32_t world, int32_t device, slam3d_comm **out);
void slam3d_comm_destroy(slam3d_comm *c);
int  slam3d_comm_rank(const slam3d_comm *c);
int  slam3d_comm_world(const slam3d_comm *c);
const char *slam3d_comm_last_error(const slam3d_comm *c);
/* contiguous block [begin, end) of `rank` when n items are dealt over `world` ranks (remainder to the lowest
 * ranks): pairs of a batch (configs 3/4), source rows of the dense mode (config 5) */
void slam3d_shard_range(int32_t n, int32_t world, int32_t rank, int32_t *begin, int32_t *end);

/* BASELINE config 5: ONE pair whose source rows are sharded over the ranks of `comm` (NULL = one rank).  Slot 0 of
 * every rank's handle holds the same pair.  Per iteration: NN + accumulate on the local rows -> ONE in-place
 * ncclAllReduce(SUM) of the iteration's int64 accumulator set (the 36 Gram totals in 16 replicas) on the handle's stream -> the
 * same solve on every rank at the head of the next launch; no host synchronisation until the result.  Integer sums are
 * order-free: the pose is bit-identical for any world size. */
int slam3d_icp_dense_run(slam3d_icp_handle *h, slam3d_comm *comm, const double *T_init, slam3d_icp_result *out);

/* The same loop over a CALLER's transport (MPI, gloo, a test harness): `allreduce` must SUM `count` int64 at d_buf over the ranks,
 * in place, ordered on hip_stream (or complete when it returns), and return 0.  rank / world shard the source rows.
 * Failure protocol (both forms): a rank whose iteration k cannot be enqueued still takes part in the remaining exchanges, with
 * zero totals and a failure word set, and returns its own error; every other rank completes its run and returns SLAM3D_E_COMM
 * (T = Identity) -- nobody is left waiting in a collective for a rank that is gone.  Only when even those exchanges cannot be
 * enqueued is the RCCL communicator aborted (the slam3d_comm is dead from then on).
 * A rank that fails before its first iteration (preprocessing) is drained the same way. */
typedef int (*slam3d_allreduce_fn)(void *ctx, void *d_int64_buf, int64_t count, void *hip_stream);
int slam3d_icp_dense_run_with(slam3d_icp_handle *h, int32_t rank, int32_t world, slam3d_allreduce_fn allreduce, void *ctx,
                              const double *T_init, slam3d_icp_result *out);
/* Test hook of the failure protocol above: iteration `dense_fail_at` of THIS handle's dense runs behaves as if it could not be
 * enqueued (SLAM3D_E_HIP on this rank, SLAM3D_E_COMM on its peers); -2: the failure happens before the first iteration; 1000 + k: behind
 * iteration k's exchange (the peers' iteration k is then complete); -1 (the default) = off.  2000 + k: the hook of the point-list kernel's
 * barrier watchdog instead -- in this handle's list runs one block never reaches the grid barrier of iteration k, the run is given up after
 * about two seconds (SLAM3D_E_HIP, see the note on unorganized clouds above).  An explicit call on a handle, not an environment variable:
 * nothing a production process inherits can switch it on. */
int slam3d_icp_set_fault_injection(slam3d_icp_handle *h, int32_t dense_fail_at);

/* BASELINE configs 3/4: pairs are independent, the only exchange is the gather of the SE(3) pose records. */
typedef struct slam3d_pose_record {      /* 160 bytes */
    double  T[16];
    double  norm;
    int32_t inliers, status;
    double  rmse;
    double  _pad;
} slam3d_pose_record;
/* ncclAllGather of n_local records per rank (the same n_local on every rank; pad with status = -1 records) into
 * all[world * n_local], rank order.  submit enqueues H2D + all-gather + D2H on the communicator's own stream and
 * returns (the exchange overlaps the caller's next kernels); collect waits for the OLDEST submitted gather.  At most
 * two gathers may be pending. */
int slam3d_pose_gather_submit(slam3d_comm *c, const slam3d_pose_record *local, int32_t n_local);
int slam3d_pose_gather_collect(slam3d_comm *c, slam3d_pose_record *all);
int slam3d_pose_gather(slam3d_comm *c, const slam3d_pose_record *local, int32_t n_local, slam3d_pose_record *all);
void slam3d_pose_record_from_result(const slam3d_icp_result *r, slam3d_pose_record *rec);

#ifdef __cplusplus
}
#endif
#endif /* SLAM3D_ICP_H */

// icp_capi.hip -- C-ABI (include/slam3d_icp.h) over the HIP kernels in icp_kernels.hpp.
//
// Replaces, behind the same pose semantics, GraphicEnd::multiPnP (src/GraphicEnd.cpp:557-659) and
// the cloud half of GraphicEnd::readimage (src/GraphicEnd.cpp:266-302).  No CPU fallback exists:
// every entry point that computes runs HIP kernels on a gfx950 device or returns an error.
#include "../../include/slam3d_icp.h"
#include "icp_kernels.hpp"
#include "list_icp.hpp"
#include "plane_seg.hpp"
#include "voxel.hpp"
#include "rccl_comm.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

using namespace s3d;

// A resident frame on the host side: which cloud the device holds for it and for which epoch of it the two roles
// were built (frames are built once, pairs reference them).
struct FrameHost {
    const float4 *cloud = nullptr;     // device: the frame's slice of the handle's pool, or a borrowed pointer
    uint64_t epoch = 0;                // bumped by every set_*; 0 = never set
    uint64_t src_epoch = 0; int src_row0 = -1, src_row1 = -1;    // source role: built for this epoch and row shard
    uint64_t tgt_epoch = 0; int tgt_normals = -1;                // target role: built for this epoch with/without normals
    uint64_t nrm_epoch = 0;                                      // the frame's normals were computed for this epoch
    bool nrm_full = false;             // ... with the normal VECTORS of every pixel (SLAM3D_EST_PLANE builds labels only for a frame that is a source of the pair gate)
    bool from_depth = false;           // the cloud is OUR back-projection of a depth image with the handle's intrinsics
    uint64_t ls_epoch[2] = { 0, 0 };   // point-list handles (list_icp.hpp): the sorted list of role 0 / 1 was built for this epoch ...
    int ls_normals = -1;               // ... the target list with / without the normal filter
};

struct slam3d_icp_handle {
    slam3d_icp_params p;
    Geometry g;
    int N = 0, maxB = 0, maxF = 0;
    hipStream_t stream = nullptr;
    hipStream_t run_stream = nullptr;
    // ---- frames (device pools, maxF entries each)
    float4 *f_cloud = nullptr, *f_nrm = nullptr, *f_srcT = nullptr, *f_tgtT = nullptr, *f_tbox = nullptr, *f_cbox = nullptr, *f_tq = nullptr;
    int *f_scount = nullptr, *f_counts = nullptr;        // [maxF][2][ntiles] per-tile counts of each role; [maxF][4] totals
    std::vector<FrameHost> frames;
    // ---- pairs
    std::vector<int> pair_src, pair_tgt;                 // frame ids, -1 = unset
    std::vector<PairPtrs> h_pairs, up_pairs;             // wanted / uploaded pair table
    int pairs_uploaded = 0;
    PairPtrs *d_pairs = nullptr;
    float4 *src_c = nullptr, *tgt_c = nullptr;           // brute-force modes: raster-compacted lists per pair
    int *ccounts = nullptr, *corr = nullptr, *flags = nullptr;
    int *chunk_cnt = nullptr;     // [maxB][2][ceil(N / 1024)] kept records per chunk (k_compact_count -> k_compact_scatter; brute-force modes)
    unsigned int *ticket = nullptr;
    unsigned long long *best = nullptr;
    float *cd2 = nullptr;
    long long *acc = nullptr, *sums = nullptr;     // integer accumulators (ACC_R replicas per pair) / raw sums of dense mode
    double *Tcur = nullptr, *trace_T = nullptr, *trace_S = nullptr;
    unsigned char *d_raw = nullptr; size_t raw_bytes = 0;
    uint16_t *d_depth = nullptr;
    int *d_idx = nullptr; float *d_d2 = nullptr;
    float4 *d_scratch4 = nullptr;
    int *corr_trace = nullptr; bool want_corr_trace = false, ran_corr_trace = false;   // [iters][maxB][nslots], opt-in
    // plane segmentation (f-2): allocated on first use
    SegState *seg_state = nullptr, *pin_seg = nullptr;
    SegScratch *seg_scratch = nullptr;     // the persistent segmentation launch's accumulators (two frames)
    int ls_test_stall_it = -1;                    // slam3d_icp_set_fault_injection(h, 2000 + k): block 0 of pair 0 never reaches the barrier of iteration k (test of the list kernel's watchdog, list_icp.hpp LS_ABORT)
    bool seg_persist = false;              // SLAM3D_SEG_PERSIST=1: developer knob, one persistent launch per pass (plane_seg.hpp: bit-identical, measured SLOWER -- 263 vs 139 us per frame)
    int *seg_labels = nullptr;
    const float4 **seg_ptrs = nullptr;
    FitState *fit_state = nullptr, *pin_fit = nullptr;   // slam3d_fit_planes
    // SLAM3D_EST_PLANE (spec S2p): the library's own segmentation scratch (one slot per frame a preprocessing pass may rebuild), the
    // planes it found per frame, the pair gate's association per pair
    SegState *pl_state = nullptr; int *pl_labels = nullptr; const float4 **pl_ptrs = nullptr;
    FramePlanes *f_planes = nullptr;      // [maxF]
    int *assoc = nullptr;                 // [maxB][8]
    slam3d_seg_params seg_sp;             // slam3d_icp_set_seg_params
    std::vector<const float4 *> pl_ptrs_up;   // what pl_ptrs holds on the device
    // voxel grid (f-1): allocated on first use
    unsigned char *vox_mem = nullptr;
    VoxTable vox;
    unsigned long long *vox_lkey = nullptr, *vox_gkey = nullptr;
    int *vox_lslot = nullptr, *vox_gslot = nullptr, *vox_m = nullptr, *vox_hist = nullptr;   // hist | start | cursor
    long long vox_calls_dense = 0, vox_calls_general = 0;      // slam3d_voxel_grid_path_counts
    unsigned long long *vox_bits = nullptr, *vox_rowbits = nullptr; int *vox_flags = nullptr;   // dense ordering path (voxel.hpp, k_voxel_finalize)
    float4 *vox_out = nullptr;
    int *pin_vox_m = nullptr, *pin_vox_m_dev = nullptr;     // host-mapped: k_voxel_scan writes the voxel counts there (one per frame of a batch)
    VoxFrame *vox_frames = nullptr;                         // a batch's (records, output, count) table
    int vox_B = 0, vox_nblk = 0;                            // frames the tables are allocated for; insert blocks per frame
    bool vox_dirty = true;        // table / histogram need a full clear before the next voxel call
    hipEvent_t vox_done = nullptr; bool vox_done_valid = false;   // end of the last voxel call's launches: the next call (any stream) waits for it
    // 8x8-pixel tiles (slot order of the sums; target tiles + boxes for the pruned NN)
    TileGrid tg;
    float4 *prevq = nullptr;
    int nn_slot = -1;             // this handle's entry of c_nn_static (icp_kernels.hpp)
    int *dev_runs = nullptr;      // the device's run counter (device_state_take)
    float *tile_cum = nullptr;    // [maxB][ntiles] motion totals of the certificates (icp_kernels.hpp)
    float2 *slot_rec = nullptr;   // every slot's result after the last iteration of the tile search: (match, clearance) (icp_kernels.hpp); [maxB][nslots]
    bool cert_on = true;          // SLAM3D_CERT=0: developer knob, every iteration searches
    long long *dbg = nullptr;     // per-tile NN statistics, only with SLAM3D_NN_DEBUG=1
    int *cost = nullptr;                    // cycles per tile of the last launch: input of k_balance (throughput build)
    int *perm_d = nullptr;                  // its cost-balanced tile->(block,wave) assignment
    bool proj_search = true;                // SLAM3D_PROJ_SEARCH=0: developer knob, the hierarchical search alone
    int nn_gx = 0, nn_gx_d = 0;             // k_nn_tiles_acc grid widths (multiples of 8): cooperative / throughput build
    float *tgtB = nullptr;        // BRUTE_MFMA: B-layout targets, f32 form [B][4][npad] floats / bf16 form [B][npad / 16][64] x 8 bytes
    bool valu_filter = false;     // BRUTE_VALU with the expanded-form filter in front of the canonical distances (SLAM3D_VALU_FILTER=1)
    int mfma_split = 0;           // SLAM3D_MFMA_SPLIT (developer knob): target slices of the matrix-core scans, 0 = automatic
    bool mfma_bf16 = true;        // BRUTE_MFMA runs the bf16-split contraction (k_nn_mfma16); SLAM3D_MFMA_BF16=0: the f32 one (k_nn_mfma)
    unsigned int *qmax2 = nullptr; int npad = 0;
    // host
    double *pin_res = nullptr, *d_res = nullptr;   // host-mapped result records (RES_REC doubles per pair) and their device address
    bool res_mapped = false;                      // the last run wrote pin_res
    double *pin_out = nullptr;    // maxB*(16+29)
    int *pin_int = nullptr;       // maxB*5
    std::vector<hipEvent_t> ev;   // 0 start, 1 after preprocess, 2 end, then (nn0,nn1) per iteration
    int dense_batch = 8;          // pairs per launch from which the throughput build of the NN kernel is used
    // point lists (height == 1, SLAM3D_NN_AUTO; list_icp.hpp): sorted lists + tile boxes per frame and role, per-pair previous matches,
    // the counting sort's scratch (LS_TASKS lists per launch sequence)
    bool list_on = false;
    bool ran_list = false;        // the last run went through the persistent list launch (its correspondences are scattered without tile slots)
    int ls_npad = 0, ls_ntile = 0;
    float4 *ls_pts = nullptr, *ls_box = nullptr;   // [maxF][2][ls_npad], [maxF][2][ls_ntile * 2]
    int2 *ls_tile = nullptr;                        // [maxF][2][ls_ntile] (start, count)
    int *ls_n = nullptr;                            // [maxF][2][2]: points, tiles
    int *ls_cnt = nullptr, *ls_cstart = nullptr, *ls_grp = nullptr; int2 *ls_cr = nullptr, *ls_super = nullptr;
    float4 *ls_match = nullptr;                     // [maxB][ls_npad]: previous match (point, index) by sorted source position
    long long *ls_dbg = nullptr;                    // SLAM3D_LIST_DEBUG=1: block 0's per-iteration stamps (slam3d_icp_get_nn_debug)
    int dense_fail_at = -1;       // slam3d_icp_set_fault_injection (tests): the dense loop's iteration that "cannot be enqueued" on this handle
    hipGraphExec_t graph_exec = nullptr;   // the captured iteration loop (slam3d_icp_run without profiling)
    int graph_B = 0;
    bool use_graph = true;
    int nsets = 1;                // accumulator sets per pair (= iterations: one per launch when the solve runs at the head of the next)
    int head_solve = 1;           // 1: head solve, polling while other runs are in flight; developer knob SLAM3D_HEAD_SOLVE: 0 two launches per iteration, 2 never poll, 3 always poll
    bool profiling = false;       // record the per-iteration events (each costs ~6 us of stream serialisation)
    bool stamping = false;        // launch stamps (slam3d_icp_set_stamping): a device ring of the last stamp_ring runs' rows
    unsigned long long *d_stamps = nullptr; unsigned int *d_stamp_seq = nullptr; int stamp_rows = 0, stamp_ring = 0;
    bool ran_profiled = false;
    bool ran = false; int last_B = 0;
    bool run_counted = false;     // k_pair_init of the run being enqueued counted it into the device's run counter; cleared once its last k_solve_acc is enqueued too
    int row0 = 0, row1 = 0; int dense_it = 0;
    std::string err;
};

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            char buf__[512];                                                                    \
            snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            if (h) (h)->err = buf__;                                                            \
            return SLAM3D_E_HIP;                                                                \
        }                                                                                       \
    } while (0)

static inline StampRing stamp_ring_of(const slam3d_icp_handle *h, bool on = true)
{
    StampRing sr;
    sr.rows = (h->stamping && on) ? h->d_stamps : nullptr; sr.seq = h->d_stamp_seq; sr.ring = h->stamp_ring; sr.rows_per_run = h->stamp_rows;
    return sr;
}

static inline int nn_mode_of(const slam3d_icp_handle *h)
{
    // an UNORGANIZED cloud (height == 1: the voxel-grid output of readimage, src/GraphicEnd.cpp:283-295) has no image tiles to prune
    // with: its search is the full scan on the matrix cores (16 k x 15 k points: a few microseconds)
    if (h->p.nn_mode == SLAM3D_NN_AUTO) return h->p.height == 1 ? SLAM3D_NN_BRUTE_MFMA : SLAM3D_NN_TILES;
    return h->p.nn_mode;
}

// rows / solve of the estimator: SLAM3D_EST_PLANE differs from POINT2PLANE only in where the target normals come from (spec S2p)
static inline bool is_p2p(const slam3d_icp_handle *h) { return h->p.estimator != SLAM3D_EST_SVD; }
static inline bool is_plane(const slam3d_icp_handle *h) { return h->p.estimator == SLAM3D_EST_PLANE; }
static inline int row_estimator(const slam3d_icp_handle *h) { return h->p.estimator == SLAM3D_EST_SVD ? 1 : 0; }

static inline void identity16(double *T) { for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0; }

extern "C" int slam3d_icp_abi_version(void) { return SLAM3D_ICP_ABI_VERSION; }

extern "C" void slam3d_icp_default_params(slam3d_icp_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->width = 640; p->height = 480;
    p->fx = 525.0; p->fy = 525.0; p->cx = 319.5; p->cy = 235.5; p->depth_factor = 1000.0; // src/convert2PCD.cpp:19-23
    p->z_filter = 7.0;
    p->iterations = 20;
    p->max_corr_dist = 0.10;
    p->estimator = SLAM3D_EST_POINT2PLANE;
    p->normal_window = 7; p->normal_min_inliers = 41; p->normal_inlier_dist = 0.01;
    p->min_inliers = 12; p->error_threshold = 1.0;
    p->max_batch = 1; p->device = 0; p->nn_mode = SLAM3D_NN_AUTO;
    p->extra_frames = 0;
    p->max_plane_residual2 = 0.0f; p->min_normal_cos = 0.0f;      // optional gates off
    p->coarse_iterations = 3;                                      // spec S4c
    p->plane_flags = 0;
}

extern "C" const char *slam3d_strerror(int code)
{
    switch (code) {
    case SLAM3D_OK: return "ok";
    case SLAM3D_TOO_FEW_INLIERS: return "too few inliers (T = Identity)";
    case SLAM3D_NORM_EXCEEDED: return "norm of transform above error_threshold (T = Identity)";
    case SLAM3D_DEGENERATE: return "degenerate geometry: normal equations needed damping or could not be solved (T = Identity)";
    case SLAM3D_E_INVALID: return "invalid argument";
    case SLAM3D_E_HIP: return "HIP runtime error";
    case SLAM3D_E_NOMEM: return "out of memory";
    case SLAM3D_E_NODEVICE: return "no gfx950 (MI355X) device visible; this library has no CPU fallback";
    case SLAM3D_E_STATE: return "call order violated";
    case SLAM3D_E_COMM: return "RCCL error";
    default: return "unknown status";
    }
}

extern "C" const char *slam3d_last_error(const slam3d_icp_handle *h) { return h ? h->err.c_str() : ""; }

// Per-DEVICE state owned by the library: the count of slam3d_icp_run runs in flight on the device, which the
// head solve of the search kernel reads to decide between polling block 0's pose and solving locally (icp_kernels.hpp).  One word
// of device memory per GPU, allocated with the first handle created on that GPU and freed with the last.  Multi-thread contract
// (documented in include/slam3d_icp.h): a handle is used by one thread at a time; DIFFERENT handles -- on the same device or not
// -- may be used from different threads concurrently (GraphicEndICP::multiPnPBatch runs one thread per handle).  The word is
// shared by the handles of a device, is touched only by device-side atomics, and shapes speed, never a result.
struct DeviceState { int *runs = nullptr; int refs = 0; };
static std::mutex g_dev_mu;
static std::unordered_map<int, DeviceState> g_dev;
static int *device_state_take(int device, hipStream_t s)      // the caller has made `device` current; s: the new handle's own stream
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    DeviceState &d = g_dev[device];
    if (!d.runs) {
        // (cleared through the handle's stream, never the null stream: a default-stream operation between the creation of one handle's
        // stream and the next changes how the runtime deals the streams to its hardware queues -- measured twice now: 65 k it/s
        // instead of 86 k with eight handles in flight, 2.2 launches resident instead of 3.1)
        if (hipMalloc((void **)&d.runs, sizeof(int)) != hipSuccess || hipMemsetAsync(d.runs, 0, sizeof(int), s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            (void)hipGetLastError();
            if (d.runs) { (void)hipFree(d.runs); d.runs = nullptr; }
            return nullptr;
        }
    }
    d.refs += 1;
    return d.runs;
}
static void device_state_give(int device)
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto it = g_dev.find(device);
    if (it == g_dev.end()) return;
    if (--it->second.refs <= 0) { if (it->second.runs) (void)hipFree(it->second.runs); g_dev.erase(it); }
}

// entries of c_nn_static (the per-handle constants of the NN kernel): a process-wide free list
static std::mutex g_nn_slot_mu;
static bool g_nn_slot_used[NN_STATIC_SLOTS];
static int nn_slot_take()
{
    std::lock_guard<std::mutex> lk(g_nn_slot_mu);
    for (int i = 0; i < NN_STATIC_SLOTS; ++i)
        if (!g_nn_slot_used[i]) { g_nn_slot_used[i] = true; return i; }
    return -1;
}
static void nn_slot_give(int i)
{
    if (i < 0) return;
    std::lock_guard<std::mutex> lk(g_nn_slot_mu);
    g_nn_slot_used[i] = false;
}

static void free_all(slam3d_icp_handle *h)
{
    auto F = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->f_cloud); F(h->f_nrm); F(h->f_srcT); F(h->f_tgtT); F(h->f_tbox); F(h->f_cbox); F(h->f_tq); F(h->f_scount); F(h->f_counts);
    F(h->seg_scratch); F(h->ls_dbg); F(h->ls_pts); F(h->ls_box); F(h->ls_tile); F(h->ls_super); F(h->ls_n); F(h->ls_cnt); F(h->ls_cstart); F(h->ls_grp); F(h->ls_cr); F(h->ls_match);
    F(h->src_c); F(h->tgt_c); F(h->ccounts); F(h->chunk_cnt); F(h->corr); F(h->ticket);
    F(h->flags); F(h->best); F(h->cd2); F(h->acc); F(h->sums); F(h->Tcur); F(h->trace_T); F(h->trace_S);
    F(h->d_pairs); F(h->d_raw); F(h->d_depth); F(h->d_idx); F(h->d_d2); F(h->d_scratch4); F(h->corr_trace);
    F(h->d_stamps); F(h->d_stamp_seq);
    if (h->vox_done) (void)hipEventDestroy(h->vox_done);
    nn_slot_give(h->nn_slot); h->nn_slot = -1;
    if (h->dev_runs) { device_state_give(h->p.device); h->dev_runs = nullptr; }
    F(h->dbg); F(h->prevq); F(h->slot_rec); F(h->tile_cum); F(h->perm_d); F(h->cost); F(h->tgtB); F(h->qmax2);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->pin_res) (void)hipHostFree(h->pin_res);
    if (h->pin_seg) (void)hipHostFree(h->pin_seg);
    if (h->pin_fit) (void)hipHostFree(h->pin_fit);
    F(h->fit_state);
    F(h->pl_state); F(h->pl_labels); F(h->pl_ptrs); F(h->f_planes); F(h->assoc);
    F(h->seg_state); F(h->seg_labels); F(h->seg_ptrs);
    F(h->vox_mem); F(h->vox_lkey); F(h->vox_lslot); F(h->vox_m); F(h->vox_out); F(h->vox_gkey); F(h->vox_gslot); F(h->vox_hist); F(h->vox_frames); F(h->vox_bits); F(h->vox_rowbits); F(h->vox_flags);
    if (h->pin_vox_m) (void)hipHostFree(h->pin_vox_m);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_int) (void)hipHostFree(h->pin_int);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
}

extern "C" void slam3d_icp_destroy(slam3d_icp_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->p.device);
    (void)hipDeviceSynchronize();
    free_all(h);
    delete h;
}

template <class T> static hipError_t dalloc(T *&p, size_t n) { return hipMalloc((void **)&p, n * sizeof(T)); }

extern "C" int slam3d_icp_create(const slam3d_icp_params *p, slam3d_icp_handle **out)
{
    if (!p || !out) return SLAM3D_E_INVALID;
    *out = nullptr;
    if (p->width <= 0 || p->height <= 0 || p->max_batch <= 0 || p->iterations < 0 || p->extra_frames < 0 || p->coarse_iterations < 0) return SLAM3D_E_INVALID;
    if (p->normal_window < 1 || (p->normal_window & 1) == 0 || p->normal_window / 2 > NRM_RMAX) return SLAM3D_E_INVALID;
    if (p->estimator != SLAM3D_EST_POINT2PLANE && p->estimator != SLAM3D_EST_SVD && p->estimator != SLAM3D_EST_PLANE) return SLAM3D_E_INVALID;
    if (p->plane_flags & ~(SLAM3D_PLANE_PAIR_GATE | SLAM3D_PLANE_ONLY)) return SLAM3D_E_INVALID;
    if (p->plane_flags != 0 && p->estimator != SLAM3D_EST_PLANE) return SLAM3D_E_INVALID;
    // unorganized clouds (height == 1) have no 7x7 image windows: svd, or the planes alone
    if (p->height == 1 && p->width > 1 && (p->estimator == SLAM3D_EST_POINT2PLANE || (p->estimator == SLAM3D_EST_PLANE && !(p->plane_flags & SLAM3D_PLANE_ONLY))))
        return SLAM3D_E_INVALID;
    if (p->nn_mode < SLAM3D_NN_AUTO || p->nn_mode > SLAM3D_NN_TILES) return SLAM3D_E_INVALID;
    if (!(p->max_corr_dist > 0.0) || !(p->z_filter > 0.0)) return SLAM3D_E_INVALID;
    {   // range of the int64 fixed-point sums (unit 2^-32): a term is at most |p|^2, all N slots may carry one.  With the
        // declared camera the farthest valid point is z_filter * sqrt(1 + tx^2 + ty^2); N * |p|^2 must stay below 2^28
        // (2^60 in fixed point; the plane moments about a sample point need the factor 4 of (2|p|)^2).  640x480 @ 7 m
        // is 11 times below, 1280x960 @ 7 m 2.8 times; a configuration beyond it is refused, not wrapped.
        // (an unorganized cloud, height == 1, has no camera model: |x|, |y| <= z is assumed -- a 90 degree field of view)
        const bool unorganized = p->height == 1;
        const double tx = (p->fx > 0.0 && !unorganized) ? fmax(p->cx, p->width - 1 - p->cx) / p->fx : 1.0;
        const double ty = (p->fy > 0.0 && !unorganized) ? fmax(p->cy, p->height - 1 - p->cy) / p->fy : 1.0;
        const double r2 = p->z_filter * p->z_filter * (1.0 + tx * tx + ty * ty);
        if (!((double)p->width * p->height * r2 < 268435456.0)) return SLAM3D_E_INVALID;
        // spec S4 (round 4): a wave's 64-row Gram sums run on the fp64 matrix cores and are converted with a 2^51 magic number:
        // 64 (r 2^16)^2 < 2^51  <=>  r^2 < 2^13 (r < 90 m); the global totals of the n.n and a.n entries need N < 2^22 and N r < 2^26
        if (!(r2 < 8192.0) || !((double)p->width * p->height < 4194304.0) || !((double)p->width * p->height * sqrt(r2) < 67108864.0))
            return SLAM3D_E_INVALID;
        // spec S2: the window moments C' = n S2 - S1 S1^T must be exact integers below 2^53: n^2 (r 2^16)^2 < 2^53 with n = window^2
        // (7x7 at z_filter 7 m: 11 times below; a 9x9 window beyond ~17 m is refused, not computed inexactly -- ADVICE r4)
        if (p->estimator != SLAM3D_EST_SVD) {
            const double nw = (double)p->normal_window * p->normal_window;
            if (!(nw * nw * r2 * 4294967296.0 < 9007199254740992.0)) return SLAM3D_E_INVALID;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) return SLAM3D_E_NODEVICE;
    auto *h = new slam3d_icp_handle();
    h->p = *p;
    if (hipSetDevice(p->device) != hipSuccess) { delete h; return SLAM3D_E_NODEVICE; }
    h->N = p->width * p->height;
    h->maxB = p->max_batch;
    h->maxF = 2 * p->max_batch + p->extra_frames;
    Geometry &g = h->g;
    g.W = p->width; g.H = p->height; g.N = h->N;
    g.zmax = (float)p->z_filter;
    g.win_r = p->normal_window / 2; g.min_in = p->normal_min_inliers; g.in_dist = p->normal_inlier_dist;
    g.gate2 = (float)(p->max_corr_dist * p->max_corr_dist);
    {   // spec S4: the residual component of a row vector is rint(b 2^eb), eb = 20 - min(k, 8) with max_corr_dist = m 2^k, 0.5 <= m < 1.
        // |b| <= |q - p'| < 2^8 whatever the gate (the range check above keeps every point within 90.5 m of the sensor), so a gate
        // beyond 256 m must not coarsen the residual any further (with eb = 20 - k a gate of 1e6 m quantised b to metres,
        // A^T b became 0, the pose never moved and the run still said OK; PCL's default gate is sqrt(DBL_MAX))
        int k = 0;
        (void)frexp(p->max_corr_dist, &k);
        g.eb = 20 - (k < 8 ? k : 8);
        g.b_scale = ldexp(1.0, g.eb);
    }
    {   // projective window search: the constant of its radius bound, from the image corners' rays
        const double a0 = fabs((0.0 - p->cx) / p->fx), a1 = fabs(((double)p->width - 1.0 - p->cx) / p->fx);
        const double b0 = fabs((0.0 - p->cy) / p->fy), b1 = fabs(((double)p->height - 1.0 - p->cy) / p->fy);
        const double am = a0 > a1 ? a0 : a1, bm = b0 > b1 ? b0 : b1;
        g.proj_c = (float)((p->fx > p->fy ? p->fx : p->fy) * sqrt(1.0 + am * am + bm * bm) * 1.001);
        h->proj_search = !(getenv("SLAM3D_PROJ_SEARCH") && atoi(getenv("SLAM3D_PROJ_SEARCH")) == 0);
    }
    g.cert_m = getenv("SLAM3D_CERT_M") ? (float)atof(getenv("SLAM3D_CERT_M")) : CERT_M;                       // developer knobs (exact for any value)
    g.cert_track = getenv("SLAM3D_CERT_TRACK") ? (float)atof(getenv("SLAM3D_CERT_TRACK")) : CERT_TRACK_MOTION;
    g.resid2 = p->max_plane_residual2 > 0.0f ? p->max_plane_residual2 : 0.0f;
    g.min_ncos = p->min_normal_cos > 0.0f ? p->min_normal_cos : 0.0f;
    g.estimator = p->estimator == SLAM3D_EST_SVD ? 1 : 0;          // the kernels know two row forms; PLANE is POINT2PLANE with other normals
    g.pair_gate = (p->estimator == SLAM3D_EST_PLANE && (p->plane_flags & SLAM3D_PLANE_PAIR_GATE)) ? 1 : 0;
    slam3d_seg_default_params(&h->seg_sp);
    h->seg_sp.distance_threshold = 0.04f;      // SLAM3D_EST_PLANE's own segmentation (spec S2p; round 6: swept 0.02-0.08 on the Kinect frames and both synthetic workloads,
                                               // DESIGN.md section 3 -- slam3d_seg_default_params keeps the reference's plane-extraction key, 0.08)
    g.fx = p->fx; g.fy = p->fy; g.cx = p->cx; g.cy = p->cy; g.factor = p->depth_factor; g.zf = p->z_filter;
    h->row0 = 0; h->row1 = p->height;
    TileGrid &tg = h->tg;
    tg.ntx = (p->width + TILE_PX - 1) / TILE_PX; tg.nty = (p->height + TILE_PX - 1) / TILE_PX;
    tg.ntiles = tg.ntx * tg.nty;
    tg.ncx = (tg.ntx + COARSE_TILES - 1) / COARSE_TILES; tg.ncy = (tg.nty + COARSE_TILES - 1) / COARSE_TILES;
    tg.ncoarse = tg.ncx * tg.ncy;
    tg.mag_ncx = (unsigned int)((0x100000000ull + tg.ncx - 1) / tg.ncx);
    tg.mag_W = (unsigned int)((0x100000000ull + p->width - 1) / p->width);
    tg.mag_ntx = (unsigned int)((0x100000000ull + tg.ntx - 1) / tg.ntx);
    tg.nchunks = (tg.ntiles + TILES_PER_CHUNK - 1) / TILES_PER_CHUNK;
    tg.nslots = tg.nchunks * CHUNK;
    const size_t BN = (size_t)h->maxB * h->N;
    const size_t BS = (size_t)h->maxB * tg.nslots;
    const size_t F = (size_t)h->maxF;
    const bool brute = nn_mode_of(h) != SLAM3D_NN_TILES;
    const int iters = p->iterations > 0 ? p->iterations : 1;
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess && r != hipSuccess) e = r; };
    A(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // frame pools: 16 B cloud + 16 B normals + 16 B source slots + 18 B target records per pixel and frame
    A(dalloc(h->f_cloud, F * h->N)); A(dalloc(h->f_nrm, F * h->N));
    A(dalloc(h->f_srcT, F * tg.ntiles * TILE_SLOTS)); A(dalloc(h->f_tgtT, F * tg.ntiles * TILE_REC));
    A(dalloc(h->f_tq, F * (size_t)h->N));
    A(dalloc(h->f_tbox, F * tg.ntiles * 2)); A(dalloc(h->f_cbox, F * tg.ncoarse * 2));
    A(dalloc(h->f_scount, F * 2 * tg.ntiles)); A(dalloc(h->f_counts, F * 4));
    if (brute) { A(dalloc(h->src_c, BN)); A(dalloc(h->tgt_c, BN)); A(dalloc(h->best, BS)); }
    h->npad = ((h->N + MF_TCH - 1) / MF_TCH + 2) * MF_TCH;      // two chunks of far-away padding: the fragment stream of k_nn_mfma runs up to 17 groups past the end
    if (nn_mode_of(h) == SLAM3D_NN_BRUTE_VALU && getenv("SLAM3D_VALU_FILTER") && atoi(getenv("SLAM3D_VALU_FILTER")) != 0) {
        h->valu_filter = true;                                                                        // developer knob
        A(dalloc(h->qmax2, (size_t)h->maxB));
    }
    if (nn_mode_of(h) == SLAM3D_NN_BRUTE_MFMA) {
        if (getenv("SLAM3D_MFMA_BF16")) h->mfma_bf16 = atoi(getenv("SLAM3D_MFMA_BF16")) != 0;            // developer knob
        if (getenv("SLAM3D_MFMA_SPLIT")) h->mfma_split = std::max(1, std::min(64, atoi(getenv("SLAM3D_MFMA_SPLIT"))));
        A(dalloc(h->tgtB, (size_t)h->maxB * (h->mfma_bf16 ? 8 : 4) * h->npad)); A(dalloc(h->qmax2, (size_t)h->maxB));
    }
    // point lists: one persistent launch per run (list_icp.hpp); SLAM3D_LIST_ICP=0 (developer knob) keeps round 5's three launches per iteration
    h->list_on = p->height == 1 && p->width > 1 && p->nn_mode == SLAM3D_NN_AUTO && !(getenv("SLAM3D_LIST_ICP") && atoi(getenv("SLAM3D_LIST_ICP")) == 0);
    if (h->list_on) {
        h->ls_ntile = h->N / 64 + 1 + std::min(h->N, LS_NSUPER);       // tiles never leave a super-cell: at most one partial tile per non-empty super-cell
        h->ls_npad = (h->N + 63) / 64 * 64 + 64;                       // + the padding behind the list
        A(dalloc(h->ls_pts, F * 2 * h->ls_npad)); A(dalloc(h->ls_box, F * 2 * h->ls_ntile * 2)); A(dalloc(h->ls_tile, F * 2 * h->ls_ntile));
        A(dalloc(h->ls_n, F * 4)); A(dalloc(h->ls_super, (size_t)LS_TASKS * LS_NSUPER));
        if (getenv("SLAM3D_LIST_DEBUG")) A(dalloc(h->ls_dbg, (size_t)iters * LS_MAX_BLOCKS * 12));
        A(dalloc(h->ls_cnt, (size_t)LS_TASKS * LS_NCELL)); A(dalloc(h->ls_cstart, (size_t)LS_TASKS * LS_NCELL)); A(dalloc(h->ls_grp, (size_t)LS_TASKS * LS_NGROUP));
        A(dalloc(h->ls_cr, (size_t)LS_TASKS * h->N)); A(dalloc(h->ls_match, (size_t)h->maxB * h->ls_npad));
    }
    A(dalloc(h->ccounts, (size_t)h->maxB * 4)); A(dalloc(h->ticket, (size_t)h->maxB * 2));      // [maxB, 2 maxB): the list kernel's claim counters
    A(dalloc(h->corr, BS)); A(dalloc(h->flags, (size_t)h->maxB)); A(dalloc(h->cd2, BS));
    if (brute) A(dalloc(h->chunk_cnt, (size_t)h->maxB * 2 * ((h->N + 1023) / 1024)));
    if (brute) A(dalloc(h->prevq, BS));      // (the tile search keeps 8-byte slot records instead: slot_rec)
    h->nsets = p->iterations > 0 ? p->iterations : 1;
    A(dalloc(h->acc, (size_t)h->maxB * h->nsets * ACC_R * ACC_STRIDE));
    A(dalloc(h->cost, (size_t)h->maxB * tg.ntiles));
    {   // grid widths of the NN kernel, multiples of 8.  Cooperative build: as many waves as tiles -- with the row ownership every
        // XCD's list is then exactly full (640x480: 1,200 blocks of four tiles).  More, smaller-loaded blocks (20 % slack: 1,440)
        // end a pair ALONE 1.5 % sooner, but with several alignments in flight the extra blocks hold LDS and wave slots that the
        // next stream's launch could use: 56.1 k -> 60.6 k it/s pipelined (SLAM3D_NN_SLACK = percent of extra waves, developer knob).
        // Throughput build: exactly one wave per tile.
        const int slack = getenv("SLAM3D_NN_SLACK") ? atoi(getenv("SLAM3D_NN_SLACK")) : 0;
        const long long waves = ((long long)tg.ntiles * (100 + (slack < 0 ? 0 : slack)) + 99) / 100;
        h->nn_gx = (int)(((waves + NN_WAVES - 1) / NN_WAVES + 7) / 8 * 8);
        if (getenv("SLAM3D_NN_GX")) {                      // developer knob: grid width of the cooperative build
            const int gx = atoi(getenv("SLAM3D_NN_GX")) / 8 * 8;
            if ((long long)gx * NN_WAVES >= tg.ntiles) h->nn_gx = gx;
        }
        {   // row-interleaved XCD ownership: every XCD needs a slot for each tile of its rows
            const int per_xcd = (tg.nty / 8) * tg.ntx + ((tg.nty % 8) * tg.ntx + 7) / 8;
            const int min_gx = 8 * ((per_xcd + NN_WAVES - 1) / NN_WAVES);
            if (h->nn_gx < min_gx) h->nn_gx = min_gx;
        }
        h->nn_gx_d = ((tg.ntiles + NN_WAVES - 1) / NN_WAVES + 7) / 8 * 8;
    }
    A(dalloc(h->perm_d, (size_t)h->maxB * h->nn_gx_d * NN_WAVES));
    if (getenv("SLAM3D_NN_DEBUG")) A(dalloc(h->dbg, (size_t)tg.ntiles * 20));
    if (getenv("SLAM3D_NO_GRAPH") || getenv("SLAM3D_NN_DEBUG")) h->use_graph = false;
    if (getenv("SLAM3D_DENSE_BATCH")) h->dense_batch = atoi(getenv("SLAM3D_DENSE_BATCH"));
    if (getenv("SLAM3D_HEAD_SOLVE")) h->head_solve = atoi(getenv("SLAM3D_HEAD_SOLVE"));
    if (getenv("SLAM3D_CERT")) h->cert_on = atoi(getenv("SLAM3D_CERT")) != 0;
    if (getenv("SLAM3D_SEG_PERSIST")) h->seg_persist = atoi(getenv("SLAM3D_SEG_PERSIST")) != 0;
    A(dalloc(h->slot_rec, (size_t)h->maxB * tg.nslots)); A(dalloc(h->tile_cum, (size_t)h->maxB * tg.ntiles));
    A(dalloc(h->sums, (size_t)h->maxB * NRAW + 8)); A(dalloc(h->Tcur, (size_t)h->maxB * 16));
    A(dalloc(h->trace_T, (size_t)h->maxB * (iters + 1) * 16)); A(dalloc(h->trace_S, (size_t)h->maxB * iters * NSUMS));
    A(dalloc(h->d_pairs, (size_t)h->maxB));
    if (p->estimator == SLAM3D_EST_PLANE) {
        A(dalloc(h->pl_state, F)); A(dalloc(h->pl_labels, F * h->N)); A(dalloc(h->pl_ptrs, F));
        A(dalloc(h->f_planes, F)); A(dalloc(h->assoc, (size_t)h->maxB * 8));
    }
    A(dalloc(h->d_depth, (size_t)h->N)); A(dalloc(h->d_idx, (size_t)h->N)); A(dalloc(h->d_d2, (size_t)h->N));
    A(dalloc(h->d_scratch4, (size_t)h->N));
    A(hipHostMalloc((void **)&h->pin_res, sizeof(double) * RES_REC * h->maxB, hipHostMallocMapped));
    A(hipHostGetDevicePointer((void **)&h->d_res, h->pin_res, 0));
    A(hipHostMalloc((void **)&h->pin_out, sizeof(double) * (16 + NRAW) * h->maxB, hipHostMallocDefault));
    A(hipHostMalloc((void **)&h->pin_int, sizeof(int) * 5 * h->maxB, hipHostMallocDefault));
    h->ev.resize(3 + 2 * (size_t)iters);
    for (auto &evx : h->ev) A(hipEventCreate(&evx));
    if (e != hipSuccess) {
        free_all(h);
        const bool oom = (e == hipErrorOutOfMemory);
        delete h;
        return oom ? SLAM3D_E_NOMEM : SLAM3D_E_HIP;
    }
    {   // the NN kernel's per-handle constants (this device's copy of the symbol)
        NnStatic st;
        st.Tcur = h->Tcur; st.corr = h->corr; st.cd2 = h->cd2; st.cost = h->cost; st.acc = h->acc; st.dbg = h->dbg;
        st.trace_T = h->trace_T; st.trace_S = h->trace_S; st.flags = h->flags; st.slot_rec = h->slot_rec; st.tile_cum = h->tile_cum;
        st.g = h->g; st.tg = tg; st.iters = iters; st.nsets = h->nsets;
        h->dev_runs = device_state_take(p->device, h->stream);
        h->nn_slot = nn_slot_take();
        // (through the handle's own stream: with hipMemcpyToSymbol -- a default-stream operation between the creation of one handle's
        // stream and the next -- four handles in flight reached 50 k it/s instead of 73 k, 1.7 launches resident instead of 3.0:
        // presumably two of their streams then shared one of the runtime's four hardware queues)
        NnStatic *sym = nullptr;
        if (h->nn_slot < 0 || !h->dev_runs || hipGetSymbolAddress((void **)&sym, HIP_SYMBOL(c_nn_static)) != hipSuccess ||
            hipMemcpyAsync(sym + h->nn_slot, &st, sizeof st, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) {
            free_all(h);
            delete h;
            return SLAM3D_E_NOMEM;
        }
    }
    (void)hipMemsetAsync(h->ticket, 0, sizeof(unsigned int) * (size_t)h->maxB * 2, h->stream);
    if (h->list_on) {      // the counting sort's counters clean up after themselves from here on
        (void)hipMemsetAsync(h->ls_cnt, 0, sizeof(int) * (size_t)LS_TASKS * LS_NCELL, h->stream);
        (void)hipMemsetAsync(h->ls_grp, 0, sizeof(int) * (size_t)LS_TASKS * LS_NGROUP, h->stream);
        (void)hipMemsetAsync(h->ls_n, 0, sizeof(int) * 4 * F, h->stream);
    }
    h->frames.assign(h->maxF, FrameHost());
    h->pair_src.assign(h->maxB, -1); h->pair_tgt.assign(h->maxB, -1);
    h->h_pairs.assign(h->maxB, PairPtrs{}); h->up_pairs.assign(h->maxB, PairPtrs{});
    (void)hipMemsetAsync(h->f_counts, 0, sizeof(int) * 4 * F, h->stream);
    if (h->f_planes) {
        (void)hipMemsetAsync(h->f_planes, 0, sizeof(FramePlanes) * F, h->stream);
        (void)hipMemsetAsync(h->assoc, 0xFF, sizeof(int) * 8 * (size_t)h->maxB, h->stream);
    }
    (void)hipMemsetAsync(h->perm_d, 0xFF, sizeof(int) * (size_t)h->maxB * h->nn_gx_d * NN_WAVES, h->stream);
    (void)hipMemsetAsync(h->acc, 0, sizeof(long long) * (size_t)h->maxB * h->nsets * ACC_R * ACC_STRIDE, h->stream);   // k_pair_init re-zeroes at every run
    *out = h;
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ frames
static inline bool slot_ok(const slam3d_icp_handle *h, int slot) { return h && slot >= 0 && slot < h->maxB; }
static inline bool frame_ok(const slam3d_icp_handle *h, int f) { return h && f >= 0 && f < h->maxF; }
static inline float4 *frame_pool_cloud(slam3d_icp_handle *h, int f) { return h->f_cloud + (size_t)f * h->N; }

// Frame uploads run on the handle's stream.  A run that was queued on a caller stream is not ordered with it, so the
// handle's stream first waits for that run's end event: a frame is never overwritten under a run that still reads it.
static int order_after_foreign_run(slam3d_icp_handle *h)
{
    if (h->ran && h->run_stream && h->run_stream != h->stream) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev[2], 0));
    return SLAM3D_OK;
}

static void frame_touch(slam3d_icp_handle *h, int f, const float4 *cloud, bool from_depth = false)
{
    FrameHost &fr = h->frames[f];
    fr.cloud = cloud;
    fr.from_depth = from_depth;
    fr.epoch += 1;                 // both roles are stale now
}

__global__ __launch_bounds__(256) void k_fill_invalid(float4 *__restrict__ dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float qnan = __int_as_float(0x7fc00000);
    if (i < n) dst[i] = make_float4(qnan, qnan, qnan, 0.0f);
}

static int upload_cloud(slam3d_icp_handle *h, const slam3d_cloud_view *v, float4 *dst)
{
    if (!v || v->stride_bytes < 12) return SLAM3D_E_INVALID;
    // an unorganized handle (height == 1) takes clouds of ANY size up to its width: the records beyond the view are invalid
    const bool ragged = h->p.height == 1 && v->height == 1 && v->width >= 0 && v->width <= h->p.width;
    if (!ragged && (v->width != h->p.width || v->height != h->p.height)) return SLAM3D_E_INVALID;
    if (!v->data && v->width > 0) return SLAM3D_E_INVALID;
    const size_t N = ragged ? (size_t)v->width : (size_t)h->N;
    if (ragged && N < (size_t)h->N) {
        hipLaunchKernelGGL(k_fill_invalid, dim3((unsigned)((h->N - N + 255) / 256)), dim3(256), 0, h->stream, dst + N, (int)(h->N - N));
        HIPCHK(h, hipGetLastError());
        if (N == 0) return SLAM3D_OK;
    }
    if (v->stride_bytes == 16) {
        HIPCHK(h, hipMemcpyAsync(dst, v->data, N * 16, hipMemcpyHostToDevice, h->stream));
        return SLAM3D_OK;
    }
    const size_t need = N * (size_t)v->stride_bytes;
    if (need > h->raw_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_raw) { HIPCHK(h, hipFree(h->d_raw)); h->d_raw = nullptr; }
        HIPCHK(h, hipMalloc((void **)&h->d_raw, need));
        h->raw_bytes = need;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_raw, v->data, need, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_repack, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, h->d_raw, v->stride_bytes, dst, (int)N);
    HIPCHK(h, hipGetLastError());
    // d_raw is reused by the next upload: keep stream order, the copy above is stream-ordered too
    return SLAM3D_OK;
}

static int backproject_dev(slam3d_icp_handle *h, const uint16_t *d_depth, float4 *dst)
{
    hipLaunchKernelGGL(k_backproject, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, d_depth, dst, h->g);
    HIPCHK(h, hipGetLastError());
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_frame_count(const slam3d_icp_handle *h) { return h ? h->maxF : 0; }

extern "C" int slam3d_icp_frame_set_cloud_host(slam3d_icp_handle *h, int32_t frame, const slam3d_cloud_view *cloud)
{
    if (!frame_ok(h, frame)) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    if (order_after_foreign_run(h)) return SLAM3D_E_HIP;
    const int rc = upload_cloud(h, cloud, frame_pool_cloud(h, frame));
    if (rc) return rc;
    frame_touch(h, frame, frame_pool_cloud(h, frame));
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_frame_set_cloud_device(slam3d_icp_handle *h, int32_t frame, const void *d_xyz4)
{
    if (!frame_ok(h, frame) || !d_xyz4) return SLAM3D_E_INVALID;
    frame_touch(h, frame, static_cast<const float4 *>(d_xyz4));
    return SLAM3D_OK;
}

// A frame set from a borrowed device pointer (set_cloud_device / set_clouds_device) is preprocessed once per SET, not once per
// run: a caller that rewrites the buffer in place must say so, or the next run aligns against the old normals and tiles.
extern "C" int slam3d_icp_frame_invalidate(slam3d_icp_handle *h, int32_t frame)
{
    if (!frame_ok(h, frame)) return SLAM3D_E_INVALID;
    FrameHost &fr = h->frames[frame];
    if (fr.epoch == 0) return SLAM3D_E_STATE;
    frame_touch(h, frame, fr.cloud, fr.from_depth);
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_frame_set_depth_host(slam3d_icp_handle *h, int32_t frame, const uint16_t *depth)
{
    if (!frame_ok(h, frame) || !depth) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    if (order_after_foreign_run(h)) return SLAM3D_E_HIP;
    // one staging image on the handle's stream: the copy of frame k+1 is ordered behind the back-projection of frame k
    HIPCHK(h, hipMemcpyAsync(h->d_depth, depth, sizeof(uint16_t) * h->N, hipMemcpyHostToDevice, h->stream));
    const int rc = backproject_dev(h, h->d_depth, frame_pool_cloud(h, frame));
    if (rc) return rc;
    frame_touch(h, frame, frame_pool_cloud(h, frame), true);
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_frame_set_depth_device(slam3d_icp_handle *h, int32_t frame, const void *d_depth)
{
    if (!frame_ok(h, frame) || !d_depth) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    if (order_after_foreign_run(h)) return SLAM3D_E_HIP;
    const int rc = backproject_dev(h, static_cast<const uint16_t *>(d_depth), frame_pool_cloud(h, frame));
    if (rc) return rc;
    frame_touch(h, frame, frame_pool_cloud(h, frame), true);
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_set_pair(slam3d_icp_handle *h, int32_t slot, int32_t src_frame, int32_t tgt_frame)
{
    if (!slot_ok(h, slot) || !frame_ok(h, src_frame) || !frame_ok(h, tgt_frame)) return SLAM3D_E_INVALID;
    h->pair_src[slot] = src_frame; h->pair_tgt[slot] = tgt_frame;
    return SLAM3D_OK;
}

// ---- the slot-wise input calls: slot b owns the implicit frames 2b (source) and 2b + 1 (target)
extern "C" int slam3d_icp_set_clouds_host(slam3d_icp_handle *h, int32_t slot, const slam3d_cloud_view *src,
                                          const slam3d_cloud_view *tgt)
{
    if (!slot_ok(h, slot)) return SLAM3D_E_INVALID;
    int rc = slam3d_icp_frame_set_cloud_host(h, 2 * slot, src);
    if (rc) return rc;
    rc = slam3d_icp_frame_set_cloud_host(h, 2 * slot + 1, tgt);
    if (rc) return rc;
    return slam3d_icp_set_pair(h, slot, 2 * slot, 2 * slot + 1);
}

extern "C" int slam3d_icp_set_depth_host(slam3d_icp_handle *h, int32_t slot, const uint16_t *src_depth,
                                         const uint16_t *tgt_depth)
{
    if (!slot_ok(h, slot) || !src_depth || !tgt_depth) return SLAM3D_E_INVALID;
    int rc = slam3d_icp_frame_set_depth_host(h, 2 * slot, src_depth);
    if (rc) return rc;
    rc = slam3d_icp_frame_set_depth_host(h, 2 * slot + 1, tgt_depth);
    if (rc) return rc;
    return slam3d_icp_set_pair(h, slot, 2 * slot, 2 * slot + 1);
}

extern "C" int slam3d_icp_set_clouds_device(slam3d_icp_handle *h, int32_t slot, const void *d_src_xyz4, const void *d_tgt_xyz4)
{
    if (!slot_ok(h, slot) || !d_src_xyz4 || !d_tgt_xyz4) return SLAM3D_E_INVALID;
    int rc = slam3d_icp_frame_set_cloud_device(h, 2 * slot, d_src_xyz4);
    if (rc) return rc;
    rc = slam3d_icp_frame_set_cloud_device(h, 2 * slot + 1, d_tgt_xyz4);
    if (rc) return rc;
    return slam3d_icp_set_pair(h, slot, 2 * slot, 2 * slot + 1);
}

extern "C" int slam3d_icp_set_depth_device(slam3d_icp_handle *h, int32_t slot, const void *d_src_depth, const void *d_tgt_depth)
{
    if (!slot_ok(h, slot) || !d_src_depth || !d_tgt_depth) return SLAM3D_E_INVALID;
    int rc = slam3d_icp_frame_set_depth_device(h, 2 * slot, d_src_depth);
    if (rc) return rc;
    rc = slam3d_icp_frame_set_depth_device(h, 2 * slot + 1, d_tgt_depth);
    if (rc) return rc;
    return slam3d_icp_set_pair(h, slot, 2 * slot, 2 * slot + 1);
}

// dense mode's exchange step, transport-agnostic: SUM over the ranks of `count` int64 at d_buf, in place, ordered on `stream` (or
// complete when the call returns); 0 = ok.  slam3d_icp_dense_run passes RCCL's ncclAllReduce, slam3d_icp_dense_run_with the caller's.
struct DenseExchange { slam3d_allreduce_fn fn; void *ctx; };
static int rccl_allreduce_thunk(void *ctx, void *d_buf, int64_t count, void *stream)
{
    slam3d_comm *c = static_cast<slam3d_comm *>(ctx);
    if (!c || !c->comm) return 1;
    const ncclResult_t nr = s3d::rccl().AllReduce(d_buf, d_buf, (size_t)count, ncclInt64, ncclSum, c->comm, (hipStream_t)stream);
    if (nr != ncclSuccess) { c->err = std::string("ncclAllReduce failed: ") + s3d::rccl().GetErrorString(nr); return 1; }
    return 0;
}
// word DENSE_POISON of an exchanged set (the Gram totals use words 0..35 of the 40): the number of ranks that FAILED locally in this
// iteration or earlier.  A rank whose iteration k cannot be enqueued keeps taking part in the remaining exchanges with zero totals
// and this word set, so no peer is left waiting in a collective; every rank then sees a non-zero word and returns SLAM3D_E_COMM.
constexpr int DENSE_POISON = 36;
__global__ void k_dense_poison(long long *__restrict__ set, int n)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k < n) set[k] = k == DENSE_POISON ? 1 : 0;
}

// the launches of one segmentation pass over B frames (spec P1-P5): nothing returns to the host.  ptrs_dev[b] = cloud of frame b,
// lab + b * N its labels, st[b] its state (zeroed here).  Shared by slam3d_segment_planes* and the preprocessing of SLAM3D_EST_PLANE.
static int enqueue_segmentation(slam3d_icp_handle *h, int B, const float4 **ptrs_dev, int *lab, SegState *st, const slam3d_seg_params *sp, hipStream_t s,
                                bool final_launch = true /* false: the caller's next kernel closes the last round itself (k_plane_normals) */)
{
    const int N = h->N;
    const SegParams P = { sp->distance_threshold, sp->plane_percent, sp->max_planes, sp->hypotheses, sp->seed };
    const int pts = B <= 2 ? SEG_PTS : SEG_PTS_BATCH;
    const dim3 pg((N + SEG_BLOCK * pts - 1) / (SEG_BLOCK * pts), B);
    HIPCHK(h, hipMemsetAsync(st, 0, sizeof(SegState) * B, s));
    if (B <= 2 && h->seg_persist) {
        // round 6 experiment (off by default): a frame alone (or the two frames of a pair) in ONE persistent launch -- pixels resident in
        // LDS, three grid barriers per RANSAC round instead of three launches (plane_seg.hpp); <= 256 blocks, so four handles' launches
        // are always co-resident -- which is also why it loses: one wave per SIMD
        const int g_max = 256 / B;
        const int ppt = (N + SEG_BLOCK * g_max - 1) / (SEG_BLOCK * g_max);
        if (ppt <= SEG_PPT_MAX) {
            if (!h->seg_scratch && hipMalloc((void **)&h->seg_scratch, sizeof(SegScratch) * 2) != hipSuccess) { (void)hipGetLastError(); return SLAM3D_E_NOMEM; }
            HIPCHK(h, hipMemsetAsync(h->seg_scratch, 0, sizeof(SegScratch) * B, s));
            const int G = (N + SEG_BLOCK * ppt - 1) / (SEG_BLOCK * ppt);
            hipLaunchKernelGGL(k_seg_persist, dim3(G, B), dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, h->seg_scratch, N, ppt, h->g.zmax, P);
            if (final_launch) hipLaunchKernelGGL(k_seg_final, dim3(B), dim3(1), 0, s, st, P.max_planes, P.percent);
            HIPCHK(h, hipGetLastError());
            return SLAM3D_OK;
        }
    }
    if (B <= 2) hipLaunchKernelGGL(k_seg_init<SEG_PTS>, pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, h->g.zmax);
    else hipLaunchKernelGGL(k_seg_init<SEG_PTS_BATCH>, pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, h->g.zmax);
    for (int r = 0; r < P.max_planes; ++r) {
        const dim3 cg(pg.x, pg.y, (P.hypotheses + SEG_HGROUP - 1) / SEG_HGROUP);
        if (B <= 2) {
            // a frame alone is bound by launch latency: three launches per round -- bookkeeping + hypotheses + consensus | moments | refinement + labels
            hipLaunchKernelGGL((k_seg_count<true, SEG_PTS>), cg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P, r);
            hipLaunchKernelGGL(k_seg_moments<SEG_PTS>, pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P.hypotheses, r);
            hipLaunchKernelGGL((k_seg_label<true, SEG_PTS>), pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P.hypotheses, P.thr, r);
        } else {
            // a batch is bound by throughput: the heads run once per frame (five launches per round)
            hipLaunchKernelGGL(k_seg_hyp, dim3(B), dim3(64), 0, s, ptrs_dev, lab, st, N, P, r);
            hipLaunchKernelGGL((k_seg_count<false, SEG_PTS_BATCH>), cg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P, r);
            hipLaunchKernelGGL(k_seg_moments<SEG_PTS_BATCH>, pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P.hypotheses, r);
            hipLaunchKernelGGL(k_seg_refine, dim3(B), dim3(64), 0, s, st, P.hypotheses, r);
            hipLaunchKernelGGL((k_seg_label<false, SEG_PTS_BATCH>), pg, dim3(SEG_BLOCK), 0, s, ptrs_dev, lab, st, N, P.hypotheses, P.thr, r);
        }
    }
    if (final_launch) hipLaunchKernelGGL(k_seg_final, dim3(B), dim3(1), 0, s, st, P.max_planes, P.percent);
    HIPCHK(h, hipGetLastError());
    return SLAM3D_OK;
}

static bool seg_params_ok(const slam3d_seg_params *sp)
{
    return sp && sp->max_planes >= 1 && sp->max_planes <= SEG_MAXP && sp->hypotheses >= 1 && sp->hypotheses <= SEG_H &&
           sp->distance_threshold > 0.0f && sp->plane_percent >= 0.0f;
}

// ------------------------------------------------------------------------------ run
static int pick_nsplit(const slam3d_icp_handle *h, int B)
{
    // enough workgroups for 256 CUs: aim at >= ~2048 blocks (query blocks are sized for ~75 % valid)
    const int qblocks = std::max(1, ((h->N * 3) / 4 + NN_BLOCK * NN_QPT - 1) / (NN_BLOCK * NN_QPT));      // (never 0: a one-point list gave 0 and a division by zero)
    int ns = (2048 + qblocks * B - 1) / (qblocks * B);
    if (ns < 1) ns = 1;
    if (ns > 16) ns = 16;
    return ns;
}

// Preprocessing of a run: every (frame, role) the pairs [0,B) use and that is stale is rebuilt ONCE (normals, tile
// records, boxes / source slots), the pair table is refreshed when it changed (kernel arguments: nothing in flight
// reads host memory), and the pairs' iteration state is reset (T_init by kernel argument too).
static int enqueue_preprocess(slam3d_icp_handle *h, int B, const double *T_init, hipStream_t s, int count_run = 0,
                              bool list = false /* a point-list run (list_icp.hpp): sorted lists instead of tile records and compacted lists */)
{
    const Geometry &g = h->g;
    const TileGrid &tg = h->tg;
    const int use_normals = is_p2p(h) ? 1 : 0;
    const bool src_normals = use_normals && (g.min_ncos > 0.0f || g.pair_gate);       // the normal-angle gate / the plane-pair gate read the source frame's normals
    std::vector<FrameTask> tasks, ntasks;
    std::vector<char> nfull;                       // per normals task: vectors wanted (1) or plane labels only (0)
    std::unordered_map<int, size_t> ntask_of;      // frame -> its normals task of this pass
    // SLAM3D_EST_PLANE: a SOURCE frame's normals are read by the pair gate for their plane labels alone (unless the normal-angle gate
    // is on too), so its 7x7-window pass is skipped: k_plane_normals writes (plane normal, 1 + r) or nothing
    const bool src_labels_only = is_plane(h) && !(g.min_ncos > 0.0f);
    auto task_of = [&](int f, int role) {
        FrameTask t;
        t.cloud = h->frames[f].cloud;
        t.nrm = h->f_nrm + (size_t)f * h->N;
        t.tiles = role == 0 ? h->f_srcT + (size_t)f * tg.ntiles * TILE_SLOTS : h->f_tgtT + (size_t)f * tg.ntiles * TILE_REC;
        t.tq = h->f_tq + (size_t)f * h->N;
        t.tbox = h->f_tbox + (size_t)f * tg.ntiles * 2;
        t.cbox = h->f_cbox + (size_t)f * tg.ncoarse * 2;
        t.scount = h->f_scount + ((size_t)f * 2 + role) * tg.ntiles;
        t.counts = h->f_counts + (size_t)f * 4;
        t.role = role; t.row0 = h->row0; t.row1 = h->row1; t.use_normals = use_normals;
        return t;
    };
    std::vector<ListTask> ltasks;
    auto list_task_of = [&](int f, int role) {
        ListTask t;
        t.cloud = h->frames[f].cloud; t.nrm = h->f_nrm + (size_t)f * h->N;
        t.pts = h->ls_pts + ((size_t)f * 2 + role) * h->ls_npad; t.box = h->ls_box + ((size_t)f * 2 + role) * h->ls_ntile * 2;
        t.tile = h->ls_tile + ((size_t)f * 2 + role) * h->ls_ntile;
        t.n = h->ls_n + ((size_t)f * 2 + role) * 2;
        t.which = role; t.use_normals = use_normals; t.i_begin = role == 0 ? h->row0 * g.W : 0; t.i_end = role == 0 ? h->row1 * g.W : h->N;
        return t;
    };
    // every pair is validated before anything is planned, and the per-frame "built for this epoch" marks are kept in a
    // local plan that is committed only after every launch of this function was enqueued without error: a failing call
    // leaves the frames stale, so the caller's retry rebuilds them (marks set inside the loop survived an
    // E_STATE / HIP error further down and the retry aligned against unbuilt tiles)
    for (int b = 0; b < B; ++b) {
        const int fs = h->pair_src[b], ft = h->pair_tgt[b];
        if (fs < 0 || ft < 0 || h->frames[fs].epoch == 0 || h->frames[ft].epoch == 0) {
            h->err = "slam3d_icp_run: pair " + std::to_string(b) + " has no frames set";
            return SLAM3D_E_STATE;
        }
    }
    std::unordered_map<int, FrameHost> plan;
    auto planned = [&](int f) -> FrameHost & {
        auto it = plan.find(f);
        if (it == plan.end()) it = plan.emplace(f, h->frames[f]).first;
        return it->second;
    };
    auto want_normals = [&](int f, bool full) {
        FrameHost &F = planned(f);
        if (F.nrm_epoch == F.epoch && (F.nrm_full || !full)) return;
        auto it = ntask_of.find(f);
        if (it != ntask_of.end()) { if (full) { nfull[it->second] = 1; F.nrm_full = true; } return; }      // (already planned in this pass: upgrade it)
        ntask_of[f] = ntasks.size();
        ntasks.push_back(task_of(f, 1)); nfull.push_back(full ? 1 : 0);
        F.nrm_epoch = F.epoch; F.nrm_full = full;
    };
    for (int b = 0; b < B; ++b) {
        const int fs = h->pair_src[b], ft = h->pair_tgt[b];
        if (list) {
            FrameHost &S = planned(fs);
            if (S.ls_epoch[0] != S.epoch) { ltasks.push_back(list_task_of(fs, 0)); S.ls_epoch[0] = S.epoch; }
            if (src_normals) want_normals(fs, !src_labels_only);
            FrameHost &T = planned(ft);
            if (T.ls_epoch[1] != T.epoch || T.ls_normals != use_normals) {
                if (use_normals) want_normals(ft, true);
                ltasks.push_back(list_task_of(ft, 1));
                T.ls_epoch[1] = T.epoch; T.ls_normals = use_normals;
            }
        } else {
            FrameHost &S = planned(fs);
            if (S.src_epoch != S.epoch || S.src_row0 != h->row0 || S.src_row1 != h->row1) {
                tasks.push_back(task_of(fs, 0));
                S.src_epoch = S.epoch; S.src_row0 = h->row0; S.src_row1 = h->row1;
            }
            if (src_normals) want_normals(fs, !src_labels_only);
            FrameHost &T = planned(ft);
            if (T.tgt_epoch != T.epoch || T.tgt_normals != use_normals) {
                if (use_normals) want_normals(ft, true);
                tasks.push_back(task_of(ft, 1));
                T.tgt_epoch = T.epoch; T.tgt_normals = use_normals;
            }
        }
        const FrameHost &S = planned(fs), &T = planned(ft);
        PairPtrs &pp = h->h_pairs[b];
        pp.src = S.cloud; pp.tgt = T.cloud;
        pp.nrm = h->f_nrm + (size_t)ft * h->N;
        pp.snrm = h->f_nrm + (size_t)fs * h->N;
        pp.tq = (T.from_depth && h->proj_search) ? h->f_tq + (size_t)ft * h->N : nullptr;
        pp.srcT = h->f_srcT + (size_t)fs * tg.ntiles * TILE_SLOTS;
        pp.tgtT = h->f_tgtT + (size_t)ft * tg.ntiles * TILE_REC;
        pp.tbox = h->f_tbox + (size_t)ft * tg.ntiles * 2;
        pp.cbox = h->f_cbox + (size_t)ft * tg.ncoarse * 2;
        pp.src_counts = h->f_counts + (size_t)fs * 4;
        pp.tgt_counts = h->f_counts + (size_t)ft * 4;
        pp.spl = h->f_planes ? h->f_planes + fs : nullptr;
        pp.tpl = h->f_planes ? h->f_planes + ft : nullptr;
        pp.assoc = h->assoc ? h->assoc + (size_t)b * 8 : nullptr;
        pp.ls_src = pp.ls_tgt = pp.ls_tbox = nullptr; pp.ls_ns = pp.ls_nt = nullptr; pp.ls_stile = pp.ls_ttile = nullptr;
        if (h->list_on) {
            pp.ls_src = h->ls_pts + ((size_t)fs * 2 + 0) * h->ls_npad; pp.ls_tgt = h->ls_pts + ((size_t)ft * 2 + 1) * h->ls_npad;
            pp.ls_tbox = h->ls_box + ((size_t)ft * 2 + 1) * h->ls_ntile * 2;
            pp.ls_stile = h->ls_tile + ((size_t)fs * 2 + 0) * h->ls_ntile; pp.ls_ttile = h->ls_tile + ((size_t)ft * 2 + 1) * h->ls_ntile;
            pp.ls_ns = h->ls_n + ((size_t)fs * 2 + 0) * 2; pp.ls_nt = h->ls_n + ((size_t)ft * 2 + 1) * 2;
        }
    }
    const bool plane_only = is_plane(h) && (h->p.plane_flags & SLAM3D_PLANE_ONLY);
    std::vector<FrameTask> wtasks;                 // the normals tasks that need the 7x7-window pass
    for (size_t k = 0; k < ntasks.size(); ++k) if (nfull[k] && !plane_only) wtasks.push_back(ntasks[k]);
    for (size_t k0 = 0; k0 < wtasks.size(); k0 += FRAME_ARGS) {
        FrameTasks a;
        const int n = (int)std::min<size_t>(FRAME_ARGS, wtasks.size() - k0);
        for (int k = 0; k < n; ++k) a.t[k] = wtasks[k0 + k];
        dim3 grid((g.W + NRM_BX - 1) / NRM_BX, (g.H + NRM_BY - 1) / NRM_BY, n);
        if (g.win_r == 3) hipLaunchKernelGGL(k_normals<3>, grid, dim3(NRM_BX, NRM_BY), 0, s, a, g);
        else hipLaunchKernelGGL(k_normals<0>, grid, dim3(NRM_BX, NRM_BY), 0, s, a, g);
    }
    if (is_plane(h) && !ntasks.empty()) {
        // spec S2p: the frames that need normals are segmented together (scratch slot k = task k), then every labelled pixel takes
        // its plane's normal and the frame's plane table is recorded
        const int nt = (int)ntasks.size();          // <= maxF: one task per frame at most
        bool ptrs_same = (int)h->pl_ptrs_up.size() >= nt;     // (a stream of pairs through the same two frame slots names the same clouds every time)
        for (int k = 0; k < nt && ptrs_same; ++k) ptrs_same = h->pl_ptrs_up[k] == ntasks[k].cloud;
        for (int k0 = 0; k0 < nt && !ptrs_same; k0 += PTR_ARGS) {
            PtrArgs a;
            const int n = std::min(PTR_ARGS, nt - k0);
            for (int k = 0; k < n; ++k) a.p[k] = ntasks[k0 + k].cloud;
            hipLaunchKernelGGL(k_set_ptrs, dim3(1), dim3(64), 0, s, h->pl_ptrs + k0, a, n);
        }
        if (!ptrs_same) {
            if ((int)h->pl_ptrs_up.size() < nt) h->pl_ptrs_up.resize(nt);
            for (int k = 0; k < nt; ++k) h->pl_ptrs_up[k] = ntasks[k].cloud;
        }
        const int src = enqueue_segmentation(h, nt, h->pl_ptrs, h->pl_labels, h->pl_state, &h->seg_sp, s, /* final launch */ false);
        if (src) return src;
        for (int k0 = 0; k0 < nt; k0 += PLANE_ARGS) {
            PlaneTasks a;
            const int n = std::min(PLANE_ARGS, nt - k0);
            for (int k = 0; k < n; ++k) {
                const size_t f = (size_t)(ntasks[k0 + k].nrm - h->f_nrm) / (size_t)h->N;
                a.t[k].lab = h->pl_labels + (size_t)(k0 + k) * h->N; a.t[k].st = h->pl_state + (k0 + k);
                a.t[k].nrm = ntasks[k0 + k].nrm; a.t[k].out = h->f_planes + f;
                a.t[k].window = (nfull[k0 + k] && !plane_only) ? 1 : 0;
            }
            hipLaunchKernelGGL(k_plane_normals, dim3((h->N + 255) / 256, n), dim3(256), 0, s, a, h->N, h->seg_sp.max_planes, h->seg_sp.plane_percent);
        }
    }
    for (size_t k0 = 0; k0 < tasks.size(); k0 += FRAME_ARGS) {
        FrameTasks a;
        const int n = (int)std::min<size_t>(FRAME_ARGS, tasks.size() - k0);
        for (int k = 0; k < n; ++k) a.t[k] = tasks[k0 + k];
        hipLaunchKernelGGL(k_frame_tiles, dim3(tg.ntiles, n), dim3(64), 0, s, a, g, tg);
        hipLaunchKernelGGL(k_coarse_boxes, dim3(tg.ncoarse, n), dim3(64), 0, s, a, tg);
    }
    for (size_t k0 = 0; k0 < ltasks.size(); k0 += LS_TASKS) {      // the counting sort of every stale list (list_icp.hpp), LS_TASKS lists at a time
        ListTasks a;
        const int n = (int)std::min<size_t>(LS_TASKS, ltasks.size() - k0);
        for (int k = 0; k < n; ++k) a.t[k] = ltasks[k0 + k];
        const dim3 pg((h->N + 255) / 256, n);
        hipLaunchKernelGGL(k_list_bin, pg, dim3(256), 0, s, a, h->ls_cnt, h->ls_grp, h->ls_cr, h->N, g.zmax);
        hipLaunchKernelGGL(k_list_scan, dim3(LS_NGROUP, n), dim3(64), 0, s, a, h->ls_cnt, h->ls_cstart, h->ls_grp, h->ls_super);
        hipLaunchKernelGGL(k_list_scatter, pg, dim3(256), 0, s, a, h->ls_cstart, h->ls_super, h->ls_cr, h->N);
        hipLaunchKernelGGL(k_list_boxes, dim3(std::max((h->ls_ntile + 3) / 4, 4), n), dim3(256), 0, s, a, h->ls_grp);
    }
    bool same = B <= h->pairs_uploaded;                 // the device copy of the pair table is still current
    for (int b = 0; b < B && same; ++b) same = memcmp(&h->up_pairs[b], &h->h_pairs[b], sizeof(PairPtrs)) == 0;
    if (!same) {
        for (int b0 = 0; b0 < B; b0 += PAIR_ARGS) {
            PairArgs a;
            const int n = B - b0 < PAIR_ARGS ? B - b0 : PAIR_ARGS;
            for (int k = 0; k < n; ++k) a.p[k] = h->up_pairs[b0 + k] = h->h_pairs[b0 + k];
            hipLaunchKernelGGL(k_set_pairs, dim3(1), dim3(64), 0, s, h->d_pairs + b0, a, n);
        }
        h->pairs_uploaded = B;
    }
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    if (T_init) {
        for (int b0 = 0; b0 < B; b0 += TINIT_ARGS) {
            TinitArgs ti;
            const int n = B - b0 < TINIT_ARGS ? B - b0 : TINIT_ARGS;
            memcpy(ti.T, T_init + (size_t)b0 * 16, sizeof(double) * 16 * n);
            hipLaunchKernelGGL(k_pair_init, dim3(n), dim3(256), 0, s, ti, 1, b0, h->Tcur, h->trace_T, h->flags, h->acc, h->ticket, h->maxB, iters, h->nsets,
                               stamp_ring_of(h, b0 == 0), (count_run && b0 == 0) ? h->dev_runs : nullptr, g.pair_gate ? h->d_pairs : nullptr);
        }
    } else {
        TinitArgs ti;
        memset(&ti, 0, sizeof ti);
        hipLaunchKernelGGL(k_pair_init, dim3(B), dim3(256), 0, s, ti, 0, 0, h->Tcur, h->trace_T, h->flags, h->acc, h->ticket, h->maxB, iters, h->nsets,
                               stamp_ring_of(h), count_run ? h->dev_runs : nullptr, g.pair_gate ? h->d_pairs : nullptr);
    }
    if (count_run) h->run_counted = true;
    if (nn_mode_of(h) != SLAM3D_NN_TILES && !list) {
        HIPCHK(h, hipMemsetAsync(h->best, 0xFF, sizeof(unsigned long long) * (size_t)B * tg.nslots, s));
        const dim3 cgrid((h->N + 1023) / 1024, 2, B);
        hipLaunchKernelGGL(k_compact_count, cgrid, dim3(1024), 0, s, h->d_pairs, h->chunk_cnt, g, use_normals, h->row0, h->row1);
        hipLaunchKernelGGL(k_compact_scatter, cgrid, dim3(1024), 0, s, h->d_pairs, h->chunk_cnt, h->src_c, h->tgt_c, h->ccounts, g, tg,
                           use_normals, h->row0, h->row1);
        if (h->valu_filter) {
            HIPCHK(h, hipMemsetAsync(h->qmax2, 0, sizeof(unsigned int) * (size_t)B, s));
            hipLaunchKernelGGL(k_qmax2, dim3((h->N + 255) / 256, B), dim3(256), 0, s, h->tgt_c, h->ccounts, h->qmax2, h->N, 0.5f * g.zmax);
        }
        if (nn_mode_of(h) == SLAM3D_NN_BRUTE_MFMA) {
            HIPCHK(h, hipMemsetAsync(h->qmax2, 0, sizeof(unsigned int) * (size_t)B, s));
            if (h->mfma_bf16)
                hipLaunchKernelGGL(k_make_bfrag16, dim3(h->npad / 256, B), dim3(256), 0, s, h->tgt_c, h->ccounts, reinterpret_cast<uint2 *>(h->tgtB),
                                   h->qmax2, h->N, h->npad, 0.5f * g.zmax);
            else
                hipLaunchKernelGGL(k_make_bfrag, dim3(h->npad / 256, B), dim3(256), 0, s, h->tgt_c, h->ccounts, h->tgtB, h->qmax2,
                                   h->N, h->npad, 0.5f * g.zmax);
        }
    }
    HIPCHK(h, hipGetLastError());
    for (const auto &kv : plan) h->frames[kv.first] = kv.second;      // everything was enqueued: the frames count as built
    return SLAM3D_OK;
}

// one iteration's data-parallel part: NN search + normal-equation chunks, then the 29-sum reduction
// (+ solve and SE(3) update when do_solve)
#ifndef S3D_COOP_WPE
#define S3D_COOP_WPE 7
#endif
static int enqueue_iteration(slam3d_icp_handle *h, int B, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int it, int do_solve,
                             long long *raw_out = nullptr, int balance = 0, int first = 0, int counted_run = 0,
                             const DenseExchange *exchange = nullptr /* dense mode: all-reduce this iteration's accumulator set in place */,
                             bool *used_head = nullptr, bool *exchanged = nullptr /* out: this iteration's exchange was enqueued */)
{
    const TileGrid &tg = h->tg;
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    bool head = false;           // this iteration's solve runs at the head of the NEXT NN launch (icp_kernels.hpp)
    // spec S4c: the first n_coarse iterations of a run -- never its last one -- search only the sources of every fourth tile
    const int run_it = do_solve ? it : h->dense_it;       // (the three-step dense loop passes it = 0 and counts in dense_it)
    const int n_coarse = std::max(0, std::min(h->p.coarse_iterations, iters - 1));
    const int cmode = run_it < n_coarse ? 1 : ((n_coarse > 0 && run_it == n_coarse) ? 2 : 0);
    if (e0) HIPCHK(h, hipEventRecord(e0, s));
    if (nn_mode_of(h) == SLAM3D_NN_TILES) {
        // few pairs: cooperative blocks (latency bound); from 8 pairs per launch: every wave on its
        // own, 8 waves per SIMD (throughput bound); three staged tile records per wave in both
        const bool dense = B >= h->dense_batch;
        head = h->head_solve != 0 && !dense && do_solve && is_p2p(h);
        const int write_out = (!do_solve || it == iters - 1 || h->want_corr_trace) ? 1 : 0;      // corr / cd2: only the last iteration's are read
        int *perm = dense ? h->perm_d : nullptr;            // the cooperative build owns tiles by the interleaved default
        const int gx = dense ? h->nn_gx_d : h->nn_gx;
        // four instances: {throughput, cooperative} x {production, instrumented (SLAM3D_NN_DEBUG: per-tile clocks and counters)}
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(gx, B), dim3(64 * NN_WAVES), 0, s, h->d_pairs, h->nn_slot, perm, write_out,
                               do_solve ? it : (first ? 0 : 1), stamp_ring_of(h, do_solve != 0), it,
                               head ? h->head_solve : 0, (h->cert_on && do_solve) ? 1 : 0, cmode, h->dev_runs);
        };
        // (+ two with the optional S4g gates compiled in: the production instances carry none of that code)
        const bool gated = is_p2p(h) && (h->g.resid2 > 0.0f || h->g.min_ncos > 0.0f || h->g.pair_gate);
        if (gated) { if (dense) launch(k_nn_tiles_acc<3, 8, false, false, true>); else launch(k_nn_tiles_acc<3, S3D_COOP_WPE, true, false, true>); }
        else if (dense) { if (h->dbg) launch(k_nn_tiles_acc<3, 8, false, true>); else launch(k_nn_tiles_acc<3, 8, false, false>); }
        else       { if (h->dbg) launch(k_nn_tiles_acc<3, S3D_COOP_WPE, true, true>);  else launch(k_nn_tiles_acc<3, S3D_COOP_WPE, true, false>); }
        // Throughput build: costs are stable from the second iteration on, balance the blocks once per run (+6 % at 64
        // pairs).  Cooperative build: never -- on a stream of DISTINCT pairs the interleaved default ownership is as good
        // as the measured-cost deal and the 12 us of k_balance are saved: +4 % pipelined, +11 % at 1280x960 (round 1
        // measured the deal on one pair repeated, where the previous run's map was already this pair's).
        if (dense && do_solve && iters > 1 && it == std::min(n_coarse + 1, iters - 1))       // (the costs of a launch in which every tile took part; a single-iteration run has no later launch to balance)
            hipLaunchKernelGGL(k_balance, dim3(B), dim3(1024), 0, s, h->cost, perm, tg, gx);
        if (e1) HIPCHK(h, hipEventRecord(e1, s));
    } else {
        if (nn_mode_of(h) == SLAM3D_NN_BRUTE_MFMA) {
            // one wave per block; target slices so that a pair alone still gives every SIMD several waves
            // (24 k waves: three resident per SIMD make eight rounds -- with 8 k the last of three rounds ran 3/4 empty: 95.6 -> 100 TFLOP/s)
            // (the bf16 form's waves are half as long: 48 k of them -- 175 -> 183 TFLOP/s-equivalent)
            const int qblocks = (h->N + MF_Q - 1) / MF_Q * B;      // = grid.x * grid.z, never 0 (N / MF_Q was 0 for lists shorter than 128 points)
            int msplit = ((h->mfma_bf16 ? 48 : 24) * 1024 + qblocks - 1) / qblocks;
            if (msplit < 1) msplit = 1;
            if (msplit > 64) msplit = 64;      // (round 5: 64, not 32 -- the 16 k x 15 k scans of unorganized clouds have 128 query blocks: 53 -> 38 us per launch)
            if (h->mfma_split > 0) msplit = h->mfma_split;                                                               // developer knob SLAM3D_MFMA_SPLIT (read at create)
            if (h->mfma_bf16)
                hipLaunchKernelGGL(k_nn_mfma16, dim3((h->N + MF_Q - 1) / MF_Q, msplit, B), dim3(64), 0, s, h->d_pairs, h->src_c, h->tgt_c,
                                   reinterpret_cast<const uint2 *>(h->tgtB), h->qmax2, h->ccounts, h->prevq, h->Tcur, h->best, h->g, tg, h->npad,
                                   0.5f * h->g.zmax, msplit, first);
            else
                hipLaunchKernelGGL(k_nn_mfma, dim3((h->N + MF_Q - 1) / MF_Q, msplit, B), dim3(64), 0, s, h->d_pairs, h->src_c, h->tgt_c,
                                   h->tgtB, h->qmax2, h->ccounts, h->prevq, h->Tcur, h->best, h->g, tg, h->npad, 0.5f * h->g.zmax,
                                   msplit, first);
        } else {
            const int nsplit = pick_nsplit(h, B);
            const int qblocks = (h->N + NN_BLOCK * NN_QPT - 1) / (NN_BLOCK * NN_QPT);
            if (h->valu_filter)            // (eight queries per thread measured slower: 4.12 vs 3.84 ms)
                hipLaunchKernelGGL((k_nn_valu<true, NN_QPT>), dim3(qblocks, nsplit, B), dim3(NN_BLOCK), 0, s, h->d_pairs, h->src_c, h->tgt_c, h->ccounts,
                                   h->prevq, h->Tcur, h->best, h->g, tg, nsplit, first, h->qmax2, 0.5f * h->g.zmax);
            else
                hipLaunchKernelGGL((k_nn_valu<false, NN_QPT>), dim3(qblocks, nsplit, B), dim3(NN_BLOCK), 0, s, h->d_pairs, h->src_c, h->tgt_c, h->ccounts,
                                   h->prevq, h->Tcur, h->best, h->g, tg, nsplit, first, nullptr, 0.0f);
        }
        if (e1) HIPCHK(h, hipEventRecord(e1, s));
        hipLaunchKernelGGL(k_accumulate, dim3(tg.nchunks, B), dim3(CHUNK), 0, s, h->d_pairs, h->Tcur, h->best,
                           h->corr, h->cd2, h->prevq, h->acc, h->g, tg, h->nsets, cmode);
    }
    if (h->want_corr_trace && do_solve)          // this iteration's slot-order indices (index parity per iteration)
        HIPCHK(h, hipMemcpyAsync(h->corr_trace + (size_t)it * h->maxB * tg.nslots, h->corr, sizeof(int) * (size_t)B * tg.nslots,
                                 hipMemcpyDeviceToDevice, s));
    if (used_head) *used_head = head;
    if (exchange && head) {
        // Dense mode, several ranks: each rank accumulated the rows of ITS source shard into this iteration's set; summing the
        // sets element-wise over the ranks (16 replicas x 40 int64 -- 36 Gram totals, the poison word, padding --, in place, on this stream) gives every rank the same
        // integers, whose replica sum is the global total: the head of the next launch (or the final k_solve_acc) then solves
        // the identical system on every rank.  ONE exchange per iteration, no reduction or solve launch beside it.
        long long *set = h->acc + (size_t)it * ACC_R * ACC_STRIDE;
        if (exchange->fn(exchange->ctx, set, (int64_t)ACC_R * ACC_STRIDE, (void *)s) != 0) { h->err = "dense mode: the all-reduce of an iteration's totals failed"; return SLAM3D_E_COMM; }
        if (exchanged) *exchanged = true;
    }
    if (head && it < iters - 1) {
        // solved at the head of the next NN launch; only the run's last iteration keeps its k_solve_acc (result record)
    } else if (is_p2p(h))
        hipLaunchKernelGGL(k_solve_acc<0>, dim3(B), dim3(64), 0, s, h->acc, raw_out, h->Tcur, h->trace_T, h->trace_S, h->flags, h->d_pairs,
                           do_solve ? h->d_res : nullptr, it, iters, do_solve,
                           stamp_ring_of(h, do_solve != 0), iters + it,
                           h->nsets, head ? it : 0, head ? 1 : 0, (counted_run && it == iters - 1) ? h->dev_runs : nullptr, h->g.eb);
    else
        hipLaunchKernelGGL(k_solve_acc<1>, dim3(B), dim3(64), 0, s, h->acc, raw_out, h->Tcur, h->trace_T, h->trace_S, h->flags, h->d_pairs,
                           do_solve ? h->d_res : nullptr, it, iters, do_solve,
                           stamp_ring_of(h, do_solve != 0), iters + it,
                           h->nsets, 0, 0, (counted_run && it == iters - 1) ? h->dev_runs : nullptr, h->g.eb);
    HIPCHK(h, hipGetLastError());
    return SLAM3D_OK;
}

static int enqueue_iterations(slam3d_icp_handle *h, int B, hipStream_t s, int iters);

extern "C" int slam3d_icp_run(slam3d_icp_handle *h, int32_t B, const double *T_init, void *stream)
{
    if (!h || B <= 0 || B > h->maxB) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (s != h->stream) {   // inputs were staged on the handle's stream
        HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
        HIPCHK(h, hipStreamWaitEvent(s, h->ev[0], 0));
    }
    HIPCHK(h, hipEventRecord(h->ev[0], s));
    const int iters = h->p.iterations;
    if (h->want_corr_trace && !h->corr_trace) {
        const size_t n = (size_t)(iters > 0 ? iters : 1) * h->maxB * h->tg.nslots;
        if (hipMalloc((void **)&h->corr_trace, sizeof(int) * n) != hipSuccess) { (void)hipGetLastError(); return SLAM3D_E_NOMEM; }
    }
    h->run_counted = false;
    int rc = enqueue_preprocess(h, B, T_init, s, iters > 0 ? 1 : 0, h->list_on);       // (counted as in flight until the last k_solve_acc)
    if (!rc) rc = enqueue_iterations(h, B, s, iters);
    if (rc) {
        // k_pair_init may have counted this run in (run_counted: it was launched); its last k_solve_acc, which counts it out,
        // will not run.  Whatever failed -- a later launch of the preprocessing, the capture, the graph launch -- the stream is
        // out of capture mode by now (enqueue_iterations always ends a capture it began), so the correction can be enqueued
        // behind k_pair_init; a count that leaked would make every later head solve poll instead of solving locally.
        if (h->run_counted) { hipLaunchKernelGGL(k_run_uncount, dim3(1), dim3(1), 0, s, h->dev_runs); (void)hipGetLastError(); h->run_counted = false; }
        return rc;
    }
    h->run_counted = false;
    HIPCHK(h, hipEventRecord(h->ev[2], s));
    h->run_stream = s;
    h->ran = true;
    h->ran_profiled = h->profiling;
    h->ran_corr_trace = h->want_corr_trace;
    h->ran_list = h->list_on && iters > 0;
    h->res_mapped = iters > 0;
    h->last_B = B;
    return SLAM3D_OK;
}

// the iteration loop of slam3d_icp_run: a graph replay, or direct launches (profiling, correspondence trace, SLAM3D_NO_GRAPH)
static int enqueue_iterations(slam3d_icp_handle *h, int B, hipStream_t s, int iters)
{
    int rc = SLAM3D_OK;
    if (h->list_on && iters > 0) {
        // point lists: ALL iterations in one persistent launch (list_icp.hpp); at most LS_MAX_BLOCKS blocks, so that the launches of
        // four streams are always co-resident
        const int G = std::max(1, std::min(h->ls_ntile, LS_MAX_BLOCKS / B));
        const int n_coarse = std::max(0, std::min(h->p.coarse_iterations, iters - 1));
        const bool gated = is_p2p(h) && (h->g.resid2 > 0.0f || h->g.min_ncos > 0.0f || h->g.pair_gate);
        if (h->profiling) { HIPCHK(h, hipEventRecord(h->ev[1], s)); HIPCHK(h, hipEventRecord(h->ev[3], s)); }
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(G, B), dim3(64 * LS_WAVES), 0, s, h->d_pairs, h->g, iters, n_coarse | ((h->ls_test_stall_it + 1) << 16), h->nsets, h->ls_npad, h->Tcur, h->trace_T,
                               h->trace_S, h->flags, h->acc, h->ticket, h->ticket + h->maxB, h->ls_match, h->corr, h->cd2, h->want_corr_trace ? h->corr_trace : nullptr,
                               h->maxB, h->tg.nslots, h->d_res, h->dev_runs, h->ls_dbg);
        };
        if (h->ls_dbg && !is_p2p(h)) launch(k_list_icp<1, false, true>);        // SLAM3D_LIST_DEBUG=1: the instrumented instance
        else if (!is_p2p(h)) launch(k_list_icp<1, false>);
        else if (gated) launch(k_list_icp<0, true>);
        else launch(k_list_icp<0, false>);
        HIPCHK(h, hipGetLastError());
        if (h->profiling)        // one launch: its time is booked on iteration 0, the other iterations read 0
            for (int it = 0; it < iters; ++it) { if (it > 0) HIPCHK(h, hipEventRecord(h->ev[3 + 2 * it], s)); HIPCHK(h, hipEventRecord(h->ev[4 + 2 * it], s)); }
        return SLAM3D_OK;
    }
    if (iters > 0 && !h->profiling && h->use_graph && !h->want_corr_trace) {
        // The iteration loop (iterations x {NN, solve}) has launch-invariant arguments: it is captured once per B into
        // a HIP graph and replayed with one hipGraphLaunch.  What changes from run to run -- which frames need their
        // preprocessing, the pair table, T_init -- stays outside (a handful of direct launches above).
        if (!h->graph_exec || h->graph_B != B) {
            if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int it = 0; it < iters && !rc; ++it) rc = enqueue_iteration(h, B, s, nullptr, nullptr, it, 1, nullptr, 0, it == 0, 1);
            const hipError_t ce = hipStreamEndCapture(s, &graph);
            if (rc || ce != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return rc ? rc : SLAM3D_E_HIP; }
            const hipError_t ie = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { h->graph_exec = nullptr; (void)hipGetLastError(); return SLAM3D_E_HIP; }
            h->graph_B = B;
        }
        HIPCHK(h, hipGraphLaunch(h->graph_exec, s));
    } else {
        if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[1], s));
        for (int it = 0; it < iters; ++it) {
            rc = enqueue_iteration(h, B, s, h->profiling ? h->ev[3 + 2 * it] : nullptr, h->profiling ? h->ev[4 + 2 * it] : nullptr, it, 1,
                                   nullptr, 0, it == 0, 1);
            if (rc) return rc;
        }
    }
    HIPCHK(h, hipGetLastError());
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_set_profiling(slam3d_icp_handle *h, int32_t on)
{
    if (!h) return SLAM3D_E_INVALID;
    h->profiling = on != 0;
    return SLAM3D_OK;
}

// Launch stamps: every NN / solve launch of the following runs records when its first block started and its last wave
// ended on the GPU's constant-rate 100 MHz real-time counter, common to all handles and streams of the device: the
// host can then tell how many launches really were resident at once without a tracer (profiles/r03_overlap.md).
// The rows of the last `ring_runs` runs stay in device memory; nothing is copied or synchronised while runs are in flight.
extern "C" int slam3d_icp_set_stamping(slam3d_icp_handle *h, int32_t ring_runs)
{
    if (!h || ring_runs < 0 || ring_runs > 4096) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipStreamSynchronize(h->run_stream ? h->run_stream : h->stream));
    if (ring_runs > 0 && ring_runs != h->stamp_ring) {
        // the ring (buffer, modulus, rows per run) is baked into the captured launches as kernel arguments: a graph that
        // outlived a re-allocation would stamp into freed memory with the old modulus
        h->graph_B = 0;
        if (h->d_stamps) { (void)hipFree(h->d_stamps); h->d_stamps = nullptr; }
        h->stamp_rows = 2 * (h->p.iterations > 0 ? h->p.iterations : 1);
        h->stamp_ring = ring_runs;
        HIPCHK(h, hipMalloc((void **)&h->d_stamps, sizeof(unsigned long long) * (size_t)ring_runs * h->stamp_rows * STAMP_ROW));
        if (!h->d_stamp_seq) HIPCHK(h, hipMalloc((void **)&h->d_stamp_seq, sizeof(unsigned int)));
    }
    if (ring_runs > 0) {        // a fresh ring: no run recorded yet
        HIPCHK(h, hipMemset(h->d_stamps, 0, sizeof(unsigned long long) * (size_t)h->stamp_ring * h->stamp_rows * STAMP_ROW));
        HIPCHK(h, hipMemset(h->d_stamp_seq, 0, sizeof(unsigned int)));
    }
    if (h->stamping != (ring_runs > 0)) h->graph_B = 0;  // the ring is a kernel argument of the captured launches
    h->stamping = ring_runs > 0;
    return SLAM3D_OK;
}

// (start, end) ticks (10 ns) of the launches of the last runs, oldest first: out[run][row][2], rows [0, iterations) the NN
// launches, [iterations, 2 iterations) the solve launches (a launch that did not run reads (~0, 0)).  Returns through
// n_runs how many runs were written (at most max_runs and the ring size).  Synchronises the handle's stream.
extern "C" int slam3d_icp_get_stamps(slam3d_icp_handle *h, uint64_t *out, int32_t max_runs, int32_t *n_runs)
{
    if (!h || !out || !n_runs || max_runs <= 0) return SLAM3D_E_INVALID;
    if (!h->d_stamps) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipStreamSynchronize(h->run_stream ? h->run_stream : h->stream));
    unsigned int seq = 0;
    HIPCHK(h, hipMemcpy(&seq, h->d_stamp_seq, sizeof seq, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> ring((size_t)h->stamp_ring * h->stamp_rows * STAMP_ROW);
    HIPCHK(h, hipMemcpy(ring.data(), h->d_stamps, ring.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int n = (int)std::min<unsigned int>(seq, (unsigned int)std::min(max_runs, h->stamp_ring));
    for (int k = 0; k < n; ++k) {
        const unsigned int run = (seq - (unsigned int)(n - 1 - k)) % (unsigned int)h->stamp_ring;      // run numbers start at 1
        for (int r = 0; r < h->stamp_rows; ++r) {
            const unsigned long long *row = ring.data() + ((size_t)run * h->stamp_rows + r) * STAMP_ROW;
            unsigned long long t0 = ~0ull, t1 = 0ull;
            for (int j = 0; j < STAMP_R; ++j) { if (row[j] < t0) t0 = row[j]; if (row[STAMP_R + j] > t1) t1 = row[STAMP_R + j]; }
            out[((size_t)k * h->stamp_rows + r) * 2] = t0; out[((size_t)k * h->stamp_rows + r) * 2 + 1] = t1;
        }
    }
    *n_runs = n;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_set_corr_trace(slam3d_icp_handle *h, int32_t on)
{
    if (!h) return SLAM3D_E_INVALID;
    h->want_corr_trace = on != 0;
    return SLAM3D_OK;
}

// S6 on the host: norm, thresholds, failure == Identity (src/GraphicEnd.cpp:599,618,621,173)
static void finish_result(const slam3d_icp_params &p, const double *T, const double *last_sums, int degenerate,
                          int n_src, int n_tgt, slam3d_icp_result *r)
{
    memcpy(r->T, T, sizeof(double) * 16);
    memcpy(r->T_raw, T, sizeof(double) * 16);
    const double tr = T[0] + T[5] + T[10];
    double ca = (tr - 1.0) / 2.0;
    if (ca > 1.0) ca = 1.0;
    if (ca < -1.0) ca = -1.0;
    const double ang = acos(ca);
    const double tn = sqrt(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    r->norm = fabs(fmin(ang, 2.0 * M_PI - ang)) + 0.9 * fabs(tn);
    r->inliers = last_sums ? (int)last_sums[27] : 0;
    r->rmse = (last_sums && last_sums[27] > 0.0) ? sqrt(last_sums[28] / last_sums[27]) : 0.0;
    r->iterations = p.iterations;
    r->n_src = n_src; r->n_tgt = n_tgt; r->_pad = 0;
    r->status = SLAM3D_OK;
    if (r->inliers < p.min_inliers) r->status = SLAM3D_TOO_FEW_INLIERS;
    else if (degenerate) r->status = SLAM3D_DEGENERATE;
    else if (r->norm > p.error_threshold) r->status = SLAM3D_NORM_EXCEEDED;
    if (r->status != SLAM3D_OK) identity16(r->T);
}

static int pair_counts(slam3d_icp_handle *h, int b, int which /* 0 n_src, 1 n_tgt */)
{
    return h->pin_int[4 * b + which];
}

extern "C" int slam3d_icp_fetch_results(slam3d_icp_handle *h, int32_t B, slam3d_icp_result *out)
{
    if (!h || !out || B <= 0 || B > h->maxB) return SLAM3D_E_INVALID;
    if (!h->ran || B > h->last_B) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = h->run_stream;
    if (h->res_mapped) {        // the final k_solve_acc wrote the records into host-mapped memory
        // wait for THIS run only (its end event), not for whatever else the caller queued on the stream since:
        // two handles can then alternate on one stream and the host never leaves the GPU idle between runs
        HIPCHK(h, hipEventSynchronize(h->ev[2]));
        bool gave_up = false;
        for (int b = 0; b < B; ++b) {
            const double *r = h->pin_res + (size_t)b * RES_REC;
            finish_result(h->p, r, r + 16, (int)r[45], (int)r[46], (int)r[47], out + b);
            gave_up = gave_up || (((int)r[45]) & 4) != 0;
        }
        if (gave_up) {      // (list_icp.hpp, LS_ABORT: the pairs concerned carry SLAM3D_DEGENERATE and the identity)
            h->err = "point-list ICP: a grid barrier of the persistent launch timed out (its blocks were not all resident: another process on the device?); the run was given up";
            return SLAM3D_E_HIP;
        }
        return SLAM3D_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->pin_out, h->Tcur, sizeof(double) * 16 * B, hipMemcpyDeviceToHost, s));
    for (int b = 0; b < B; ++b) {
        HIPCHK(h, hipMemcpyAsync(h->pin_int + 4 * b, h->f_counts + (size_t)h->pair_src[b] * 4, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->pin_int + 4 * b + 1, h->f_counts + (size_t)h->pair_tgt[b] * 4 + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipMemcpyAsync(h->pin_int + 4 * (size_t)h->maxB, h->flags, sizeof(int) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b)
        finish_result(h->p, h->pin_out + 16 * (size_t)b, nullptr, h->pin_int[4 * (size_t)h->maxB + b], pair_counts(h, b, 0),
                      pair_counts(h, b, 1), out + b);
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ one-call API
static int first_bad_status(const slam3d_icp_result *out, int B)
{
    for (int b = 0; b < B; ++b) if (out[b].status != SLAM3D_OK) return out[b].status;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_align_batch(slam3d_icp_handle *h, int32_t B, const slam3d_cloud_view *src,
                                      const slam3d_cloud_view *tgt, const double *T_init, slam3d_icp_result *out)
{
    if (!h || !src || !tgt || !out || B <= 0 || B > h->maxB) return SLAM3D_E_INVALID;
    for (int b = 0; b < B; ++b) {
        const int rc = slam3d_icp_set_clouds_host(h, b, src + b, tgt + b);
        if (rc) return rc;
    }
    int rc = slam3d_icp_run(h, B, T_init, nullptr);
    if (rc) return rc;
    rc = slam3d_icp_fetch_results(h, B, out);
    if (rc) return rc;
    return B == 1 ? out[0].status : (first_bad_status(out, B) ? first_bad_status(out, B) : SLAM3D_OK);
}

extern "C" int slam3d_icp_align(slam3d_icp_handle *h, const slam3d_cloud_view *src, const slam3d_cloud_view *tgt,
                                const double *T_init, slam3d_icp_result *out)
{
    return slam3d_icp_align_batch(h, 1, src, tgt, T_init, out);
}

extern "C" int slam3d_icp_align_depth_batch(slam3d_icp_handle *h, int32_t B, const uint16_t *const *src_depth,
                                            const uint16_t *const *tgt_depth, const double *T_init,
                                            slam3d_icp_result *out)
{
    if (!h || !src_depth || !tgt_depth || !out || B <= 0 || B > h->maxB) return SLAM3D_E_INVALID;
    for (int b = 0; b < B; ++b) {
        const int rc = slam3d_icp_set_depth_host(h, b, src_depth[b], tgt_depth[b]);
        if (rc) return rc;
    }
    int rc = slam3d_icp_run(h, B, T_init, nullptr);
    if (rc) return rc;
    rc = slam3d_icp_fetch_results(h, B, out);
    if (rc) return rc;
    return first_bad_status(out, B);
}

// ------------------------------------------------------------------------------ introspection
static int corr_to_host(slam3d_icp_handle *h, int slot, const int *corr_slot, const float *cd2_slot, int32_t *idx, float *d2)
{
    hipStream_t s = h->run_stream;
    const int N = h->N;
    hipLaunchKernelGGL(k_fill_corr, dim3((N + 255) / 256), dim3(256), 0, s, h->d_idx, h->d_d2, N);
    if (h->p.iterations > 0 && h->list_on && h->ran_list)
        hipLaunchKernelGGL(k_scatter_corr_list, dim3((N + 255) / 256), dim3(256), 0, s, h->d_pairs, corr_slot, cd2_slot, slot, N, h->g.zmax, h->d_idx, h->d_d2);
    else if (h->p.iterations > 0)
        hipLaunchKernelGGL(k_scatter_corr, dim3((h->tg.nslots + 255) / 256), dim3(256), 0, s, h->d_pairs, corr_slot, cd2_slot,
                           slot, h->tg, h->d_idx, h->d_d2);
    HIPCHK(h, hipGetLastError());
    if (idx) HIPCHK(h, hipMemcpyAsync(idx, h->d_idx, sizeof(int) * N, hipMemcpyDeviceToHost, s));
    if (d2) HIPCHK(h, hipMemcpyAsync(d2, h->d_d2, sizeof(float) * N, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_correspondences(slam3d_icp_handle *h, int32_t slot, int32_t *idx, float *d2)
{
    if (!slot_ok(h, slot)) return SLAM3D_E_INVALID;
    if (!h->ran || slot >= h->last_B) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    return corr_to_host(h, slot, h->corr + (size_t)slot * h->tg.nslots, h->cd2 + (size_t)slot * h->tg.nslots, idx, d2);
}

extern "C" int slam3d_icp_get_correspondences_at(slam3d_icp_handle *h, int32_t slot, int32_t it, int32_t *idx)
{
    if (!slot_ok(h, slot) || !idx || it < 0 || it >= h->p.iterations) return SLAM3D_E_INVALID;
    if (!h->ran || !h->ran_corr_trace || !h->corr_trace || slot >= h->last_B) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    return corr_to_host(h, slot, h->corr_trace + ((size_t)it * h->maxB + slot) * h->tg.nslots, nullptr, idx, nullptr);
}

extern "C" int slam3d_icp_get_trace(slam3d_icp_handle *h, int32_t slot, double *T_trace, double *sums_trace)
{
    if (!slot_ok(h, slot)) return SLAM3D_E_INVALID;
    if (!h->ran || slot >= h->last_B) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = h->run_stream;
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    if (T_trace)
        HIPCHK(h, hipMemcpyAsync(T_trace, h->trace_T + (size_t)slot * (iters + 1) * 16,
                                 sizeof(double) * 16 * (h->p.iterations + 1), hipMemcpyDeviceToHost, s));
    if (sums_trace && h->p.iterations > 0)
        HIPCHK(h, hipMemcpyAsync(sums_trace, h->trace_S + (size_t)slot * iters * NSUMS,
                                 sizeof(double) * NSUMS * h->p.iterations, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_clouds(slam3d_icp_handle *h, int32_t slot, float *src_xyz4, float *tgt_xyz4, float *tgt_nrm4)
{
    if (!slot_ok(h, slot) || h->pair_src[slot] < 0 || h->pair_tgt[slot] < 0) return SLAM3D_E_INVALID;
    const FrameHost &S = h->frames[h->pair_src[slot]], &T = h->frames[h->pair_tgt[slot]];
    if (!S.cloud || !T.cloud) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = h->ran ? h->run_stream : h->stream;
    const size_t bytes = sizeof(float) * 4 * (size_t)h->N;
    if (s != h->stream) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (src_xyz4) HIPCHK(h, hipMemcpyAsync(src_xyz4, S.cloud, bytes, hipMemcpyDeviceToHost, s));
    if (tgt_xyz4) HIPCHK(h, hipMemcpyAsync(tgt_xyz4, T.cloud, bytes, hipMemcpyDeviceToHost, s));
    if (tgt_nrm4) {
        if (!h->ran || !is_p2p(h)) return SLAM3D_E_STATE;
        HIPCHK(h, hipMemcpyAsync(tgt_nrm4, h->f_nrm + (size_t)h->pair_tgt[slot] * h->N, bytes, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipStreamSynchronize(s));
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_timings(slam3d_icp_handle *h, float ms[4])
{
    if (!h || !ms) return SLAM3D_E_INVALID;
    if (!h->ran) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipEventSynchronize(h->ev[2]));
    float pre = 0, tot = 0, nn = 0;
    if (h->ran_profiled) HIPCHK(h, hipEventElapsedTime(&pre, h->ev[0], h->ev[1]));
    HIPCHK(h, hipEventElapsedTime(&tot, h->ev[0], h->ev[2]));
    for (int it = 0; h->ran_profiled && it < h->p.iterations; ++it) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, h->ev[3 + 2 * it], h->ev[4 + 2 * it]));
        nn += t;
    }
    ms[0] = pre; ms[1] = nn; ms[2] = tot - pre - nn; ms[3] = tot;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_nn_debug(slam3d_icp_handle *h, int64_t *out /* ntiles*8 */, int32_t n)
{
    if (!h || !out) return SLAM3D_E_INVALID;
    if (h->ls_dbg && h->ran_list) {         // a list handle: block 0's stamps, iterations x 4
        const int m = (h->p.iterations > 0 ? h->p.iterations : 1) * LS_MAX_BLOCKS * 12;
        if (n < m) return SLAM3D_E_STATE;
        HIPCHK(h, hipSetDevice(h->p.device));
        HIPCHK(h, hipStreamSynchronize(h->run_stream));
        HIPCHK(h, hipMemcpy(out, h->ls_dbg, sizeof(long long) * (size_t)m, hipMemcpyDeviceToHost));
        return SLAM3D_OK;
    }
    if (!h->dbg || !h->ran || n < h->tg.ntiles * 20) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipStreamSynchronize(h->run_stream));
    HIPCHK(h, hipMemcpy(out, h->dbg, sizeof(long long) * (size_t)h->tg.ntiles * 20, hipMemcpyDeviceToHost));
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_iteration_timings(slam3d_icp_handle *h, float *nn_ms)
{
    if (!h || !nn_ms) return SLAM3D_E_INVALID;
    if (!h->ran || !h->ran_profiled) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipEventSynchronize(h->ev[2]));
    for (int it = 0; it < h->p.iterations; ++it)
        HIPCHK(h, hipEventElapsedTime(&nn_ms[it], h->ev[3 + 2 * it], h->ev[4 + 2 * it]));
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ building blocks
extern "C" int slam3d_backproject_u16(slam3d_icp_handle *h, const uint16_t *depth, float *xyz4)
{
    if (!h || !depth || !xyz4) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipMemcpyAsync(h->d_depth, depth, sizeof(uint16_t) * h->N, hipMemcpyHostToDevice, h->stream));
    const int rc = backproject_dev(h, h->d_depth, h->d_scratch4);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(xyz4, h->d_scratch4, sizeof(float) * 4 * (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLAM3D_OK;
}

extern "C" int slam3d_fit_planes(slam3d_icp_handle *h, const slam3d_cloud_view *cloud, const int32_t *labels,
                                 int32_t nplanes, slam3d_plane *planes)
{
    if (!h || !cloud || !labels || !planes || nplanes <= 0 || nplanes > FIT_MAXP) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    const int N = h->N;
    int rc = upload_cloud(h, cloud, h->d_scratch4);
    if (rc) return rc;
    if (!h->fit_state) {
        if (hipMalloc((void **)&h->fit_state, sizeof(FitState)) != hipSuccess ||
            hipHostMalloc((void **)&h->pin_fit, sizeof(FitState), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SLAM3D_E_NOMEM;
        }
    }
    hipStream_t s = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_idx, labels, sizeof(int) * N, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemsetAsync(h->fit_state, 0, sizeof(FitState), s));
    hipLaunchKernelGGL(k_fit_moments, dim3((N + SEG_BLOCK * SEG_PTS - 1) / (SEG_BLOCK * SEG_PTS)), dim3(SEG_BLOCK), 0, s, h->d_scratch4,
                       h->d_idx, N, nplanes, h->fit_state);
    hipLaunchKernelGGL(k_fit_refine, dim3(1), dim3(64), 0, s, h->fit_state, nplanes);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->pin_fit, h->fit_state, sizeof(FitState), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int pl = 0; pl < nplanes; ++pl) {
        const SegPlane &q = h->pin_fit->out[pl];
        slam3d_plane &P = planes[pl];
        memset(&P, 0, sizeof P);
        P.coeff[0] = q.a; P.coeff[1] = q.b; P.coeff[2] = q.c; P.coeff[3] = q.d;
        P.count = q.count;
        P.centroid[0] = q.cx; P.centroid[1] = q.cy; P.centroid[2] = q.cz;
    }
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ a9: plane association
extern "C" int slam3d_match_planes(const slam3d_plane *p1, int32_t n1, const slam3d_plane *p2, int32_t n2, int32_t *train_idx,
                                   float *distance)
{
    if (n1 < 0 || n2 < 0 || (n1 > 0 && (!p1 || !train_idx)) || (n2 > 0 && !p2)) return SLAM3D_E_INVALID;
    for (int i = 0; i < n1; ++i) {
        int best = -1;
        float bd = __builtin_inff();
        for (int j = 0; j < n2; ++j) {
            float d2 = 0.0f;
            for (int k = 0; k < 4; ++k) { const float e = p1[i].coeff[k] - p2[j].coeff[k]; d2 = fmaf(e, e, d2); }
            if (d2 < bd) { bd = d2; best = j; }
        }
        train_idx[i] = best;
        if (distance) distance[i] = best >= 0 ? sqrtf(bd) : __builtin_inff();
    }
    return SLAM3D_OK;
}

extern "C" int slam3d_plane_gate(const slam3d_plane *p1, int32_t n1, const slam3d_plane *p2, int32_t n2, const double *T,
                                 float max_dist, int32_t *n_matched)
{
    if (!T || !n_matched || n1 < 0 || n2 < 0 || n1 > 8 || (n1 > 0 && !p1) || (n2 > 0 && !p2)) return SLAM3D_E_INVALID;
    *n_matched = 0;
    if (n1 == 0 || n2 == 0) return SLAM3D_OK;
    slam3d_plane moved[8];
    for (int i = 0; i < n1; ++i) {
        const double a = p1[i].coeff[0], b = p1[i].coeff[1], c = p1[i].coeff[2], d = p1[i].coeff[3];
        // plane n.X + d = 0 in frame 1; X_2 = R X_1 + t  =>  n' = R n, d' = d - n'.t
        double n[3];
        for (int r = 0; r < 3; ++r) n[r] = (T[r * 4] * a + T[r * 4 + 1] * b) + T[r * 4 + 2] * c;
        double dd = d - ((n[0] * T[3] + n[1] * T[7]) + n[2] * T[11]);
        if (dd < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; dd = -dd; }        // src/GraphicEnd.cpp:383-387
        moved[i] = p1[i];
        moved[i].coeff[0] = (float)n[0]; moved[i].coeff[1] = (float)n[1]; moved[i].coeff[2] = (float)n[2]; moved[i].coeff[3] = (float)dd;
    }
    int32_t idx[8];
    float dist[8];
    const int rc = slam3d_match_planes(moved, n1, p2, n2, idx, dist);
    if (rc) return rc;
    for (int i = 0; i < n1; ++i) if (idx[i] >= 0 && dist[i] <= max_dist) *n_matched += 1;
    return SLAM3D_OK;
}

extern "C" int slam3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ------------------------------------------------------------------------------ frame ingestion filters (f-1)
static void vox_free(slam3d_icp_handle *h)
{
    auto F = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->vox_mem); F(h->vox_lkey); F(h->vox_lslot); F(h->vox_m); F(h->vox_out); F(h->vox_gkey); F(h->vox_gslot); F(h->vox_hist); F(h->vox_frames); F(h->vox_bits); F(h->vox_rowbits); F(h->vox_flags);
    if (h->pin_vox_m) { (void)hipHostFree(h->pin_vox_m); h->pin_vox_m = nullptr; }
    h->vox_B = 0;
}

// tables for B frames per launch sequence (a later, larger batch re-allocates: nothing of a call survives it -- the tables are
// self-cleaning).  Per frame at 640x480: 64 MB of slots (every point its own voxel still fits), 12 MB of lists and rows.
static int vox_alloc(slam3d_icp_handle *h, int B = 1)
{
    if (h->vox_mem && B <= h->vox_B) return SLAM3D_OK;
    if (h->vox_mem) {
        if (h->vox_done_valid) { (void)hipEventSynchronize(h->vox_done); h->vox_done_valid = false; }
        vox_free(h);
    }
    int cap = 1024;
    while (cap < h->N) cap <<= 1;                         // every point its own voxel still fits; typical load ~ 0.1
    // insert blocks: runs of 256 records, or 16x16 tiles of the organized image (ragged edges need a few more)
    int nblk = std::max((h->N + VOX_BLOCK - 1) / VOX_BLOCK, ((h->p.width + VOX_TW - 1) / VOX_TW) * ((h->p.height + VOX_TW - 1) / VOX_TW));
    nblk = (nblk + VOX_SEG_ALIGN - 1) / VOX_SEG_ALIGN * VOX_SEG_ALIGN;      // list blocks own whole segments of the claim lists
    const size_t Bz = (size_t)B;
    if (hipMalloc((void **)&h->vox_mem, Bz * cap * sizeof(VoxSlot)) != hipSuccess ||
        hipMalloc((void **)&h->vox_lkey, sizeof(unsigned long long) * Bz * nblk * VOX_BLOCK) != hipSuccess ||
        hipMalloc((void **)&h->vox_lslot, sizeof(int) * Bz * nblk * VOX_BLOCK) != hipSuccess ||
        hipMalloc((void **)&h->vox_m, sizeof(int) * Bz * (nblk + 1)) != hipSuccess ||            // per frame: [0] kept-count of pass_transform, [1..] claims per insert block
        hipMalloc((void **)&h->vox_gkey, sizeof(unsigned long long) * Bz * h->N) != hipSuccess ||
        hipMalloc((void **)&h->vox_gslot, sizeof(int) * Bz * h->N) != hipSuccess ||
        hipMalloc((void **)&h->vox_hist, sizeof(int) * Bz * VOX_HIST_INTS) != hipSuccess ||
        hipMalloc((void **)&h->vox_bits, sizeof(unsigned long long) * Bz * VOX_BINS * VOX_BW) != hipSuccess ||
        hipMalloc((void **)&h->vox_rowbits, sizeof(unsigned long long) * Bz * VOX_BINS * VOX_BW) != hipSuccess ||
        hipMalloc((void **)&h->vox_flags, sizeof(int) * Bz) != hipSuccess ||
        hipMalloc((void **)&h->vox_out, sizeof(float4) * h->N) != hipSuccess ||
        hipMalloc((void **)&h->vox_frames, sizeof(VoxFrame) * Bz) != hipSuccess ||
        hipHostMalloc((void **)&h->pin_vox_m, sizeof(int) * Bz, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&h->pin_vox_m_dev, h->pin_vox_m, 0) != hipSuccess) {
        (void)hipGetLastError();
        vox_free(h);
        return SLAM3D_E_NOMEM;
    }
    h->vox.slot = reinterpret_cast<VoxSlot *>(h->vox_mem);
    h->vox.cap = cap;
    h->vox_B = B;
    h->vox_nblk = nblk;
    h->vox_dirty = true;                                  // first use: slots and histogram are cleared once
    return SLAM3D_OK;
}

static VoxLayout vox_layout(const slam3d_icp_handle *h)
{
    VoxLayout L;
    L.t = h->vox;
    L.lkey = h->vox_lkey; L.lslot = h->vox_lslot; L.bcount = h->vox_m; L.blk_stride = h->vox_nblk;
    L.hist = h->vox_hist; L.hist_stride = VOX_HIST_INTS;
    L.gkey = h->vox_gkey; L.gslot = h->vox_gslot; L.g_stride = h->N;
    L.m_host = h->pin_vox_m_dev;
    L.bits = h->vox_bits; L.rowbits = h->vox_rowbits; L.flags = h->vox_flags;
    return L;
}

constexpr int VOXF_ARGS = 128;
struct VoxFrameArgs { VoxFrame f[VOXF_ARGS]; };
__global__ void k_set_voxframes(VoxFrame *__restrict__ dst, VoxFrameArgs a, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = a.f[threadIdx.x];
}

// PassThrough z in [zmin, zmax] + VoxelGrid(leaf) of B frames in ONE launch sequence (grid.y = frame).  frames[b] = (records, output,
// count); B == 1: the record travels as a kernel argument.  n_out[b] on the host when the call returns; the records follow in
// stream order (stream != NULL) or are complete (stream == NULL: the handle's stream, drained).
static int voxel_grid_impl(slam3d_icp_handle *h, int B, const VoxFrame *frames, float leaf, float zmin, float zmax, int32_t *n_out, void *stream)
{
    if (!h || !frames || !n_out || B < 1 || B > 4096 || !(leaf > 0.0f)) return SLAM3D_E_INVALID;
    int nmax = 0;
    bool all_full = true;
    for (int b = 0; b < B; ++b) {
        if (frames[b].n < 0 || frames[b].n > h->N || (frames[b].n > 0 && (!frames[b].pts || !frames[b].out))) return SLAM3D_E_INVALID;
        nmax = std::max(nmax, frames[b].n);
        all_full = all_full && frames[b].n == h->N;
        n_out[b] = 0;
    }
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = vox_alloc(h, B);
    if (rc) return rc;
    if (nmax == 0) return SLAM3D_OK;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // The handle's tables (hash slots, histogram, claim lists) serve one call at a time.  With a caller's stream a call returns
    // as soon as the count is known, the rest of its finalize launch still running: whatever stream the NEXT call runs on first
    // waits for the end of those launches (two streams raced on the tables).
    if (!h->vox_done) HIPCHK(h, hipEventCreateWithFlags(&h->vox_done, hipEventDisableTiming));
    if (h->vox_done_valid) HIPCHK(h, hipStreamWaitEvent(s, h->vox_done, 0));
    if (h->vox_dirty) {      // allocation, or an earlier call failed half way: from then on every call cleans up after itself
        hipLaunchKernelGGL(k_voxel_clear, dim3((h->vox.cap + VOX_BLOCK - 1) / VOX_BLOCK, h->vox_B), dim3(VOX_BLOCK), 0, s, h->vox);
        HIPCHK(h, hipMemsetAsync(h->vox_hist, 0, sizeof(int) * (size_t)h->vox_B * VOX_HIST_INTS, s));     // histograms AND the scan tickets
        HIPCHK(h, hipMemsetAsync(h->vox_bits, 0, sizeof(unsigned long long) * (size_t)h->vox_B * VOX_BINS * VOX_BW, s));
        HIPCHK(h, hipMemsetAsync(h->vox_flags, 0, sizeof(int) * (size_t)h->vox_B, s));
    }
    h->vox_dirty = true;     // until this call has run to its end
    const VoxLayout L = vox_layout(h);
    const VoxFrame *d_frames = nullptr;
    if (B > 1) {
        for (int b0 = 0; b0 < B; b0 += VOXF_ARGS) {
            VoxFrameArgs a;
            const int n = std::min(VOXF_ARGS, B - b0);
            memcpy(a.f, frames + b0, sizeof(VoxFrame) * n);
            hipLaunchKernelGGL(k_set_voxframes, dim3(1), dim3(VOXF_ARGS), 0, s, h->vox_frames + b0, a, n);
        }
        d_frames = h->vox_frames;
    }
    // an organized cloud (all width x height records present) is cut into 16x16-pixel tiles, anything else into runs of 256
    const bool org = all_full;
    // a list block sums `passes` runs of 1,024 records before it touches the global table (voxel.hpp): 4 when the batch fills the chip anyway
    // (55 blocks per 221 k-record frame), 2 for one or two lists (109 blocks each: the call's latency is the longest block)
    int passes = org ? 1 : (B >= 4 ? VOX_MAX_PASSES : 2);      // (measured on the reference's 221 k-record frames: tools/quick_voxel.py)
    if (!org) { if (const char *e = getenv("SLAM3D_VOX_PASSES")) passes = std::min(VOX_MAX_PASSES, std::max(1, atoi(e))); }      // developer knob
    const int run = org ? VOX_BLOCK : VOX_LIST_BLOCK, segs = run / VOX_BLOCK;
    const int nins = org ? ((h->p.width + VOX_TW - 1) / VOX_TW) * ((h->p.height + VOX_TW - 1) / VOX_TW) : (nmax + passes * run - 1) / (passes * run);
    const int nblk = nins * passes * segs;                // segments of the claim lists = blocks of the launches that walk them
    if (org) hipLaunchKernelGGL(k_voxel_insert<true>, dim3(nins, B), dim3(VOX_BLOCK), 0, s, frames[0], d_frames, h->p.width, h->p.height, 1.0f / leaf, zmin, zmax, L, passes);
    else hipLaunchKernelGGL(k_voxel_insert<false>, dim3(nins, B), dim3(VOX_LIST_BLOCK), 0, s, frames[0], d_frames, h->p.width, h->p.height, 1.0f / leaf, zmin, zmax, L, passes);
    for (int b = 0; b < B; ++b) ((volatile int *)h->pin_vox_m)[b] = -1;
    // the slabs PassThrough lets through (vox_key's arithmetic: floorf(z * inv_leaf), monotone in z) -> the scan blocks that can hold a bit
    int sb0 = 0, sb1 = VOX_SCAN_BLOCKS;
    {
        const float inv_leaf = 1.0f / leaf;
        const float lo = floorf(zmin * inv_leaf), hi = floorf(zmax * inv_leaf);
        if (lo == lo && lo > -1.0e6f) sb0 = std::min(VOX_SCAN_BLOCKS - 1, std::max(0, ((int)lo + (int)zbias) * VOX_BY / 1024));
        if (hi == hi && hi < 1.0e6f) sb1 = std::max(sb0 + 1, std::min(VOX_SCAN_BLOCKS, ((int)hi + (int)zbias) * VOX_BY / 1024 + 1));
    }
    hipLaunchKernelGGL(k_voxel_scan<true>, dim3(sb1 - sb0, B), dim3(1024), 0, s, L, sb0);
    hipLaunchKernelGGL(k_voxel_finalize, dim3(nblk, B), dim3(VOX_BLOCK), 0, s, frames[0], d_frames, L, nins, passes * segs, sb0, sb1);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->vox_done, s));
    h->vox_done_valid = true;
    // the general ordering path for the frames the dense one flagged (a claimed voxel outside the bitmap's key range: clouds in a world
    // frame, huge leaves): histogram from the claim lists, scan, scatter, rank -- the round-5 sequence, on those frames only
    auto general_path = [&]() -> int {
        for (int b = 0; b < B; ++b) if (((volatile int *)h->pin_vox_m)[b] == -2) ((volatile int *)h->pin_vox_m)[b] = -1;
        hipLaunchKernelGGL(k_voxel_hist, dim3(nblk, B), dim3(VOX_BLOCK), 0, s, L);
        hipLaunchKernelGGL(k_voxel_scan<false>, dim3(VOX_SCAN_BLOCKS, B), dim3(1024), 0, s, L, 0);
        hipLaunchKernelGGL(k_voxel_scatter, dim3(nblk, B), dim3(VOX_BLOCK), 0, s, L);
        hipLaunchKernelGGL(k_voxel_rank, dim3(nblk, B), dim3(VOX_BLOCK), 0, s, frames[0], d_frames, L);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemsetAsync(h->vox_flags, 0, sizeof(int) * (size_t)B, s));
        HIPCHK(h, hipEventRecord(h->vox_done, s));
        return SLAM3D_OK;
    };
    for (int pass = 0; pass < 2; ++pass) {
    bool flagged = false;
    if (stream) {
        // A caller's stream: the records are ready IN STREAM ORDER, and the call returns as soon as the counts are known --
        // the first workgroup of k_voxel_finalize writes them into host-mapped memory while the others still emit records.
        for (int b = 0; b < B; ++b) {
            int m = -1;
            for (unsigned spins = 1; (m = ((volatile int *)h->pin_vox_m)[b]) == -1; ++spins) {
                if ((spins & 0x3ff) != 0) continue;
                const hipError_t q = hipStreamQuery(s);                 // a failed launch must not leave us spinning
                if (q == hipErrorNotReady) continue;
                HIPCHK(h, q);
                m = ((volatile int *)h->pin_vox_m)[b];
                if (m == -1) { h->err = "voxel grid: the stream drained without a count"; return SLAM3D_E_HIP; }
                break;
            }
            n_out[b] = m;
            flagged = flagged || m == -2;
        }
    } else {
        HIPCHK(h, hipStreamSynchronize(s));      // the voxel counts are in host-mapped memory by now (written by k_voxel_finalize)
        for (int b = 0; b < B; ++b) { n_out[b] = ((volatile int *)h->pin_vox_m)[b]; flagged = flagged || n_out[b] == -2; }
    }
    if (pass == 0) ++(flagged ? h->vox_calls_general : h->vox_calls_dense);
    if (!flagged) break;
    if (pass == 1) { h->err = "voxel grid: the general path left a frame flagged"; return SLAM3D_E_HIP; }
    { const int rc2 = general_path(); if (rc2) return rc2; }
    }
    h->vox_dirty = false;
    return SLAM3D_OK;
}

extern "C" int slam3d_voxel_grid_path_counts(slam3d_icp_handle *h, int64_t counts[2])
{
    if (!h || !counts) return SLAM3D_E_INVALID;
    counts[0] = h->vox_calls_dense; counts[1] = h->vox_calls_general;
    return SLAM3D_OK;
}

static int voxel_grid_one(slam3d_icp_handle *h, const void *d_points16, int32_t n, float leaf, float zmin, float zmax, void *d_out16,
                          int32_t *n_out, void *stream)
{
    if (!h || !d_points16 || !d_out16 || !n_out || n < 0 || n > h->N) return SLAM3D_E_INVALID;
    VoxFrame f = { static_cast<const float4 *>(d_points16), static_cast<float4 *>(d_out16), n, 0 };
    return voxel_grid_impl(h, 1, &f, leaf, zmin, zmax, n_out, stream);
}

extern "C" int slam3d_voxel_grid_device(slam3d_icp_handle *h, const void *d_points16, int32_t n, float leaf, void *d_out16,
                                        int32_t *n_out, void *stream)
{
    if (!h) return SLAM3D_E_INVALID;
    return voxel_grid_one(h, d_points16, n, leaf, 0.0f, h->g.zmax, d_out16, n_out, stream);      // PassThrough z in [0, z_filter]
}

// B frames in one launch sequence: the keyframes saveOutput merges (src/saveOutput.cpp:58-96), the 30 loop-closure candidates
extern "C" int slam3d_voxel_grid_batch_device(slam3d_icp_handle *h, int32_t B, const void *const *d_points16, const int32_t *n, float leaf,
                                              void *const *d_out16, int32_t *n_out, void *stream)
{
    if (!h || !d_points16 || !n || !d_out16 || !n_out || B < 1 || B > 4096) return SLAM3D_E_INVALID;
    std::vector<VoxFrame> fr((size_t)B);
    for (int b = 0; b < B; ++b) fr[b] = VoxFrame{ static_cast<const float4 *>(d_points16[b]), static_cast<float4 *>(d_out16[b]), n[b], 0 };
    return voxel_grid_impl(h, B, fr.data(), leaf, 0.0f, h->g.zmax, n_out, stream);
}

static int voxel_host(slam3d_icp_handle *h, const void *points16, int32_t n, float leaf, float zmin, float zmax, void *out16, int32_t *n_out)
{
    if (!h || !points16 || !out16 || !n_out || n < 0 || n > h->N) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = vox_alloc(h);
    if (rc) return rc;
    if (n > 0) HIPCHK(h, hipMemcpyAsync(h->d_scratch4, points16, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    rc = voxel_grid_one(h, h->d_scratch4, n, leaf, zmin, zmax, h->vox_out, n_out, h->stream);
    if (rc) return rc;
    if (*n_out > 0) HIPCHK(h, hipMemcpyAsync(out16, h->vox_out, (size_t)*n_out * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SLAM3D_OK;
}

// pcl::VoxelGrid alone: every finite point, no PassThrough (src/saveOutput.cpp:80-83 and :97-100)
extern "C" int slam3d_voxel_grid_only(slam3d_icp_handle *h, const void *points16, int32_t n, float leaf, void *out16, int32_t *n_out)
{
    const float inf = __builtin_inff();
    return voxel_host(h, points16, n, leaf, -inf, inf, out16, n_out);
}

// src/saveOutput.cpp:84-92: PassThrough z in [0, z_max] + pcl::transformPointCloud by T (row-major 4x4); out16[i] is the
// transformed record or NaN when record i was dropped; *n_kept = records kept
extern "C" int slam3d_pass_transform(slam3d_icp_handle *h, const void *points16, int32_t n, float z_max, const double *T,
                                     void *out16, int32_t *n_kept)
{
    if (!h || !points16 || !out16 || !n_kept || !T || n < 0 || n > h->N) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = vox_alloc(h);
    if (rc) return rc;
    hipStream_t s = h->stream;
    Pose34 P;
    for (int k = 0; k < 12; ++k) P.m[k] = T[k];
    // (vox_m doubles as the claim counts of the voxel grid: a device-resident call on a caller's stream may still have its scatter
    //  launch queued -- wait for it like the next voxel call would)
    if (h->vox_done_valid) HIPCHK(h, hipStreamWaitEvent(s, h->vox_done, 0));
    HIPCHK(h, hipMemsetAsync(h->vox_m, 0, sizeof(int), s));
    if (n > 0) {
        HIPCHK(h, hipMemcpyAsync(h->d_scratch4, points16, (size_t)n * 16, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pass_transform, dim3((n + VOX_BLOCK - 1) / VOX_BLOCK), dim3(VOX_BLOCK), 0, s, h->d_scratch4, n, z_max, P,
                           h->vox_out, h->vox_m);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out16, h->vox_out, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(h, hipMemcpyAsync(h->pin_vox_m, h->vox_m, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    *n_kept = *h->pin_vox_m;
    return SLAM3D_OK;
}

extern "C" int slam3d_voxel_grid(slam3d_icp_handle *h, const void *points16, int32_t n, float leaf, void *out16, int32_t *n_out)
{
    if (!h) return SLAM3D_E_INVALID;
    return voxel_host(h, points16, n, leaf, 0.0f, h->g.zmax, out16, n_out);
}

// ------------------------------------------------------------------------------ plane segmentation (f-2)
extern "C" void slam3d_seg_default_params(slam3d_seg_params *sp)
{
    if (!sp) return;
    sp->distance_threshold = 0.08f;     // parameters.yaml:45 distance_threshold
    sp->plane_percent = 0.2f;           // parameters.yaml:46 plane_percent
    sp->max_planes = 3;                 // parameters.yaml:47 max_planes
    sp->hypotheses = 64;
    sp->seed = 1;
}

static int seg_alloc(slam3d_icp_handle *h)
{
    if (h->seg_state) return SLAM3D_OK;
    if (hipMalloc((void **)&h->seg_state, sizeof(SegState) * h->maxB) != hipSuccess ||
        hipMalloc((void **)&h->seg_labels, sizeof(int) * (size_t)h->maxB * h->N) != hipSuccess ||
        hipMalloc((void **)&h->seg_ptrs, sizeof(float4 *) * h->maxB) != hipSuccess ||
        hipHostMalloc((void **)&h->pin_seg, sizeof(SegState) * h->maxB, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return SLAM3D_E_NOMEM;
    }
    return SLAM3D_OK;
}

extern "C" int slam3d_segment_planes_device(slam3d_icp_handle *h, int32_t B, const void *const *d_clouds,
                                            const slam3d_seg_params *sp, slam3d_plane *planes, int32_t *nplanes,
                                            int32_t *d_labels, void *stream)
{
    if (!h || !d_clouds || !sp || !planes || !nplanes || B <= 0 || B > h->maxB) return SLAM3D_E_INVALID;
    if (!seg_params_ok(sp)) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = seg_alloc(h);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    for (int b0 = 0; b0 < B; b0 += PTR_ARGS) {
        PtrArgs a;
        const int n = B - b0 < PTR_ARGS ? B - b0 : PTR_ARGS;
        for (int k = 0; k < n; ++k) {
            if (!d_clouds[b0 + k]) return SLAM3D_E_INVALID;
            a.p[k] = static_cast<const float4 *>(d_clouds[b0 + k]);
        }
        hipLaunchKernelGGL(k_set_ptrs, dim3(1), dim3(64), 0, s, h->seg_ptrs + b0, a, n);
    }
    int *lab = d_labels ? d_labels : h->seg_labels;
    rc = enqueue_segmentation(h, B, h->seg_ptrs, lab, h->seg_state, sp, s);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->pin_seg, h->seg_state, sizeof(SegState) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) {
        const SegState &st = h->pin_seg[b];
        if (st.n_valid < 0) { h->err = "plane segmentation: a grid barrier of the persistent launch (SLAM3D_SEG_PERSIST) timed out; the pass was given up"; return SLAM3D_E_HIP; }
        nplanes[b] = st.nplanes;
        for (int r = 0; r < sp->max_planes; ++r) {
            slam3d_plane &o = planes[(size_t)b * sp->max_planes + r];
            memset(&o, 0, sizeof o);
            if (r >= st.nplanes) continue;
            const SegPlane &q = st.planes[r];
            o.coeff[0] = q.a; o.coeff[1] = q.b; o.coeff[2] = q.c; o.coeff[3] = q.d;
            o.count = q.count;
            o.centroid[0] = q.cx; o.centroid[1] = q.cy; o.centroid[2] = q.cz;
        }
    }
    return SLAM3D_OK;
}

extern "C" int slam3d_segment_planes(slam3d_icp_handle *h, const slam3d_cloud_view *cloud, const slam3d_seg_params *sp,
                                     slam3d_plane *planes, int32_t *nplanes, int32_t *labels)
{
    if (!h || !cloud || !sp || !planes || !nplanes) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = upload_cloud(h, cloud, h->d_scratch4);
    if (rc) return rc;
    const void *ptr = h->d_scratch4;
    rc = slam3d_segment_planes_device(h, 1, &ptr, sp, planes, nplanes, nullptr, h->stream);
    if (rc) return rc;
    if (labels) HIPCHK(h, hipMemcpy(labels, h->seg_labels, sizeof(int) * (size_t)h->N, hipMemcpyDeviceToHost));
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ SLAM3D_EST_PLANE: parameters / introspection
extern "C" int slam3d_icp_set_seg_params(slam3d_icp_handle *h, const slam3d_seg_params *sp)
{
    if (!h || !seg_params_ok(sp)) return SLAM3D_E_INVALID;
    if (!is_plane(h)) return SLAM3D_E_STATE;
    h->seg_sp = *sp;
    for (auto &fr : h->frames) { fr.nrm_epoch = 0; fr.tgt_epoch = 0; }      // normals and target tiles of every frame are stale
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_frame_planes(slam3d_icp_handle *h, int32_t frame, slam3d_plane *planes, int32_t *nplanes)
{
    if (!frame_ok(h, frame) || !planes || !nplanes) return SLAM3D_E_INVALID;
    *nplanes = 0;
    memset(planes, 0, sizeof(slam3d_plane) * 8);
    if (!h->f_planes) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    if (h->ran && h->run_stream) HIPCHK(h, hipStreamSynchronize(h->run_stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    FramePlanes fp;
    HIPCHK(h, hipMemcpy(&fp, h->f_planes + frame, sizeof fp, hipMemcpyDeviceToHost));
    if (h->frames[frame].nrm_epoch != h->frames[frame].epoch || h->frames[frame].epoch == 0) return SLAM3D_OK;      // not built (for this content)
    *nplanes = fp.n;
    for (int r = 0; r < fp.n && r < 8; ++r) {
        planes[r].coeff[0] = fp.pl[r].a; planes[r].coeff[1] = fp.pl[r].b; planes[r].coeff[2] = fp.pl[r].c; planes[r].coeff[3] = fp.pl[r].d;
        planes[r].count = fp.pl[r].count;
        planes[r].centroid[0] = fp.pl[r].cx; planes[r].centroid[1] = fp.pl[r].cy; planes[r].centroid[2] = fp.pl[r].cz;
    }
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_get_plane_assoc(slam3d_icp_handle *h, int32_t slot, int32_t *assoc)
{
    if (!slot_ok(h, slot) || !assoc) return SLAM3D_E_INVALID;
    if (!h->assoc || !h->g.pair_gate || !h->ran || slot >= h->last_B) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    HIPCHK(h, hipStreamSynchronize(h->run_stream));
    HIPCHK(h, hipMemcpy(assoc, h->assoc + (size_t)slot * 8, sizeof(int) * 8, hipMemcpyDeviceToHost));
    return SLAM3D_OK;
}

// ------------------------------------------------------------------------------ dense mode
extern "C" int slam3d_icp_dense_set_rows(slam3d_icp_handle *h, int32_t row_begin, int32_t row_end)
{
    if (!h || row_begin < 0 || row_end > h->p.height || row_begin > row_end) return SLAM3D_E_INVALID;
    h->row0 = row_begin; h->row1 = row_end;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_begin(slam3d_icp_handle *h, const double *T_init, void *stream)
{
    if (!h) return SLAM3D_E_INVALID;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (s != h->stream) {
        HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
        HIPCHK(h, hipStreamWaitEvent(s, h->ev[0], 0));
    }
    const int rc = enqueue_preprocess(h, 1, T_init, s);
    if (rc) return rc;
    h->dense_it = 0;
    HIPCHK(h, hipEventRecord(h->ev[0], s));
    if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[1], s));
    h->run_stream = s; h->ran = true; h->last_B = 1; h->res_mapped = false; h->ran_list = false; h->ran_profiled = h->profiling;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_partial(slam3d_icp_handle *h, int64_t sums[SLAM3D_ICP_NRAW], void *stream)
{
    if (!h || !sums) return SLAM3D_E_INVALID;
    if (!h->ran) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->run_stream;
    const int rc = enqueue_iteration(h, 1, s, nullptr, nullptr, 0, 0, h->sums, 0, h->dense_it == 0);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->pin_out, h->sums, sizeof(int64_t) * NRAW, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    memcpy(sums, h->pin_out, sizeof(int64_t) * NRAW);
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_update(slam3d_icp_handle *h, const int64_t sums[SLAM3D_ICP_NRAW], void *stream)
{
    if (!h || !sums) return SLAM3D_E_INVALID;
    if (!h->ran || h->dense_it >= (h->p.iterations > 0 ? h->p.iterations : 1)) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->run_stream;
    memcpy(h->pin_out, sums, sizeof(int64_t) * NRAW);           // pin_out holds (16+36)*maxB 8-byte words
    HIPCHK(h, hipMemcpyAsync(h->sums, h->pin_out, sizeof(int64_t) * NRAW, hipMemcpyHostToDevice, s));
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    if (is_p2p(h))
        hipLaunchKernelGGL(k_solve<0>, dim3(1), dim3(64), 0, s, h->sums, h->Tcur, h->trace_T, h->trace_S, h->flags, h->dense_it, iters, h->g.eb);
    else
        hipLaunchKernelGGL(k_solve<1>, dim3(1), dim3(64), 0, s, h->sums, h->Tcur, h->trace_T, h->trace_S, h->flags, h->dense_it, iters, h->g.eb);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(s));   // pin_out is reused by the next call
    h->dense_it++;
    return SLAM3D_OK;
}

// Device-resident forms of the three calls above: the 29 sums stay in a caller-owned device buffer, so the
// exchange (an RCCL all-reduce on the same stream) needs no host round trip; nothing here synchronises.
extern "C" int slam3d_icp_dense_partial_device(slam3d_icp_handle *h, int64_t *d_sums, void *stream)
{
    if (!h || !d_sums) return SLAM3D_E_INVALID;
    if (!h->ran) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->run_stream;
    // with profiling on, the NN launch of iteration dense_it is bracketed by HIP events like in slam3d_icp_run
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    const bool ev = h->ran_profiled && h->dense_it < iters;
    return enqueue_iteration(h, 1, s, ev ? h->ev[3 + 2 * h->dense_it] : nullptr, ev ? h->ev[4 + 2 * h->dense_it] : nullptr, 0, 0,
                             reinterpret_cast<long long *>(d_sums), 0, h->dense_it == 0);
}

extern "C" int slam3d_icp_dense_update_device(slam3d_icp_handle *h, const int64_t *d_sums, void *stream)
{
    if (!h || !d_sums) return SLAM3D_E_INVALID;
    const int iters = h->p.iterations > 0 ? h->p.iterations : 1;
    if (!h->ran || h->dense_it >= iters) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->run_stream;
    if (is_p2p(h))
        hipLaunchKernelGGL(k_solve<0>, dim3(1), dim3(64), 0, s, reinterpret_cast<const long long *>(d_sums), h->Tcur, h->trace_T, h->trace_S, h->flags, h->dense_it, iters, h->g.eb);
    else
        hipLaunchKernelGGL(k_solve<1>, dim3(1), dim3(64), 0, s, reinterpret_cast<const long long *>(d_sums), h->Tcur, h->trace_T, h->trace_S, h->flags, h->dense_it, iters, h->g.eb);
    HIPCHK(h, hipGetLastError());
    h->dense_it++;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_finish_device(slam3d_icp_handle *h, const int64_t *d_last_sums, void *stream,
                                              slam3d_icp_result *out)
{
    if (!h || !out || !d_last_sums) return SLAM3D_E_INVALID;
    if (!h->ran) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = stream ? (hipStream_t)stream : h->run_stream;
    double *ps = h->pin_out + 16;
    HIPCHK(h, hipMemcpyAsync(h->pin_out, h->Tcur, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(ps, d_last_sums, sizeof(int64_t) * NRAW, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipEventRecord(h->ev[2], s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int, h->f_counts + (size_t)h->pair_src[0] * 4, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int + 1, h->f_counts + (size_t)h->pair_tgt[0] * 4 + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int + 4, h->flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    double ls[NSUMS];
    for (int k = 0; k < NSUMS; ++k) ls[k] = derive_sum(row_estimator(h), h->g.eb, k, reinterpret_cast<const long long *>(ps));
    finish_result(h->p, h->pin_out, ls, h->pin_int[4], h->pin_int[0], h->pin_int[1], out);
    out->iterations = h->dense_it;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_finish(slam3d_icp_handle *h, const int64_t last_sums[SLAM3D_ICP_NRAW], slam3d_icp_result *out)
{
    if (!h || !out) return SLAM3D_E_INVALID;
    if (!h->ran) return SLAM3D_E_STATE;
    HIPCHK(h, hipSetDevice(h->p.device));
    hipStream_t s = h->run_stream;
    HIPCHK(h, hipMemcpyAsync(h->pin_out, h->Tcur, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int, h->f_counts + (size_t)h->pair_src[0] * 4, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int + 1, h->f_counts + (size_t)h->pair_tgt[0] * 4 + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(h->pin_int + 4, h->flags, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    double ls[NSUMS];
    for (int k = 0; k < NSUMS; ++k) ls[k] = last_sums ? derive_sum(row_estimator(h), h->g.eb, k, reinterpret_cast<const long long *>(last_sums)) : 0.0;
    finish_result(h->p, h->pin_out, ls, h->pin_int[4], h->pin_int[0], h->pin_int[1], out);
    out->iterations = h->dense_it;
    return SLAM3D_OK;
}

// BASELINE config 5 inside the library: source rows sharded over the ranks of `comm`, ONE ncclAllReduce(SUM) per iteration
// on the handle's own stream and nothing else between two NN launches (round 3): launch k accumulates this rank's rows into
// accumulator set k, the all-reduce sums set k over the ranks in place (16 x 40 int64), and the head of launch k+1 solves
// -- the same integers, hence the same pose bits, on every rank (round 2: reduce launch -> all-reduce of 29 words -> solve
// launch).  The svd estimator and SLAM3D_HEAD_SOLVE=0 keep that three-step form.  A rank that fails locally aborts ITS side of
// the communicator (ncclCommAbort) and marks the slam3d_comm dead (every later call with it returns SLAM3D_E_COMM).  Whether the
// peers' pending collectives then return is up to RCCL's transport -- it is not guaranteed intra-node --, so a host program
// that must survive a rank's failure should run its ranks under a watchdog (torchrun does: a dead rank ends the job).
// SLAM3D_DENSE_FORCE_COLLECTIVE=1 runs the collective with one rank too (developer knob: exercises RCCL on a single GPU).
//
// What is NOT sharded, and why: the target's preprocessing (normals, tile records: ~0.15 ms of a 2.0 ms alignment at
// 1280x960).  Its products are 62 MB (19.7 MB normals, 22 MB tile records, 19.7 MB image-order records, boxes) from 2.4 MB of
// depth image: all-gathering them over xGMI costs several times what every rank needs to recompute them from the image it
// already holds.  The same holds for the H2D of the pair (each rank uploads both images over its own PCIe link).
static int dense_run_impl(slam3d_icp_handle *h, int rank, int world, const DenseExchange *ex, const double *T_init, slam3d_icp_result *out)
{
    const bool collective = ex != nullptr;
    int r0 = 0, r1 = h->p.height;
    slam3d_shard_range(h->p.height, world, rank, &r0, &r1);
    int rc = slam3d_icp_dense_set_rows(h, r0, r1);
    if (rc) return rc;
    hipStream_t s = h->stream;
    rc = slam3d_icp_dense_begin(h, T_init, s);
    const int iters = h->p.iterations;
    const bool head_flow = h->head_solve != 0 && is_p2p(h) && nn_mode_of(h) == SLAM3D_NN_TILES && 1 < h->dense_batch && iters > 0;
    int64_t *d_sums = reinterpret_cast<int64_t *>(h->sums);
    const int fail_at = h->dense_fail_at;      // failure injection (tests; slam3d_icp_set_fault_injection): this handle's iteration fail_at cannot be enqueued
    int failed_it = -1;
    bool exchanged = false;                    // iteration failed_it's exchange was already enqueued when the failure happened
    // a rank that fails BEFORE the loop (dense_begin, the memset below) has no exchange enqueued yet, while its peers queue all of
    // theirs: it is a failure of iteration 0 and is drained like any other (it used to skip the drain and leave the peers
    // in their first all-reduce)
    if (!rc && fail_at == -2) { h->err = "dense mode: injected failure before the first iteration (slam3d_icp_set_fault_injection)"; rc = SLAM3D_E_HIP; }
    if (rc) failed_it = 0;
    const bool three_step_exchange = !head_flow && collective;
    for (int it = 0; it < iters && !rc; ++it) {
        exchanged = false;
        if (it == fail_at) { h->err = "dense mode: injected failure (slam3d_icp_set_fault_injection)"; rc = SLAM3D_E_HIP; }
        else if (head_flow) {
            const bool ev = h->ran_profiled;
            bool used = false;
            rc = enqueue_iteration(h, 1, s, ev ? h->ev[3 + 2 * it] : nullptr, ev ? h->ev[4 + 2 * it] : nullptr, it, 1, nullptr, 0, it == 0, 0, ex, &used, &exchanged);
            if (!rc && !used) rc = SLAM3D_E_STATE;
            if (!rc && fail_at == 1000 + it) { h->err = "dense mode: injected failure behind the exchange (slam3d_icp_set_fault_injection)"; rc = SLAM3D_E_HIP; }
        } else {
            // the three-step exchange carries the poison word behind the 36 totals; cleared before every partial, so that what the
            // all-reduce leaves there is the number of ranks that failed in THIS exchange (not a running re-sum of earlier ones)
            if (three_step_exchange && hipMemsetAsync(d_sums + NRAW, 0, sizeof(int64_t) * 8, s) != hipSuccess) { (void)hipGetLastError(); h->err = "dense mode: hipMemsetAsync failed"; rc = SLAM3D_E_HIP; }
            if (!rc) rc = slam3d_icp_dense_partial_device(h, d_sums, s);
            if (!rc && collective) {
                if (ex->fn(ex->ctx, d_sums, NRAW + 1, (void *)s) != 0) { h->err = "dense mode: the all-reduce of an iteration's totals failed"; rc = SLAM3D_E_COMM; }
                else exchanged = true;
            }
            if (!rc && fail_at == 1000 + it) { h->err = "dense mode: injected failure behind the exchange (slam3d_icp_set_fault_injection)"; rc = SLAM3D_E_HIP; }
            if (!rc) rc = slam3d_icp_dense_update_device(h, d_sums, s);
        }
        if (rc) failed_it = it;
    }
    if (rc < 0 && rc != SLAM3D_E_COMM && collective && failed_it >= 0) {
        // This rank leaves the loop early.  Its peers have every remaining exchange queued already (nothing synchronises with the
        // host before the result): take part in all of them with zero totals and the poison word set, so that nobody waits for a
        // rank that is gone and everybody learns of the failure.  If even that cannot be enqueued the transport is aborted below.
        // The drain starts BEHIND an exchange this rank already enqueued for the failing iteration (a failure of the update or of
        // the final solve after the all-reduce): one exchange too many would block this rank in a collective nobody joins.
        bool drained = true;
        for (int it = failed_it + (exchanged ? 1 : 0); it < iters && drained; ++it) {
            long long *set = head_flow ? h->acc + (size_t)it * ACC_R * ACC_STRIDE : reinterpret_cast<long long *>(d_sums);
            const int n = head_flow ? ACC_R * ACC_STRIDE : NRAW + 1;
            hipLaunchKernelGGL(k_dense_poison, dim3((n + 63) / 64), dim3(64), 0, s, set, n);
            drained = hipGetLastError() == hipSuccess && ex->fn(ex->ctx, set, n, (void *)s) == 0;
        }
        if (drained) { (void)hipStreamSynchronize(s); h->err += " (the remaining exchanges were completed with the failure flag set: peers return SLAM3D_E_COMM)"; }
        else rc = rc == SLAM3D_E_COMM ? rc : SLAM3D_E_COMM - 100;      // marks "could not drain" for the caller below
    }
    if (!rc) {
        if (head_flow) {
            h->dense_it = iters;
            if (hipEventRecord(h->ev[2], s) != hipSuccess) rc = SLAM3D_E_HIP;
            h->res_mapped = true;
            if (!rc) rc = slam3d_icp_fetch_results(h, 1, out);
            if (!rc) out->iterations = iters;
        } else rc = slam3d_icp_dense_finish_device(h, d_sums, s, out);
    }
    if (rc >= 0 && collective && iters > 0) {
        // did a peer fail?  (the poison words of every exchanged set; the stream is drained by now)
        std::vector<long long> w((size_t)iters, 0);
        hipError_t ce;
        if (head_flow) ce = hipMemcpy2D(w.data(), sizeof(long long), h->acc + DENSE_POISON, sizeof(long long) * ACC_R * ACC_STRIDE, sizeof(long long), (size_t)iters, hipMemcpyDeviceToHost);
        else ce = hipMemcpy(w.data(), d_sums + DENSE_POISON, sizeof(long long), hipMemcpyDeviceToHost);
        if (ce != hipSuccess) { (void)hipGetLastError(); rc = SLAM3D_E_HIP; }
        else {
            for (int it = 0; it < iters; ++it)
                if (w[it] != 0) {
                    h->err = "dense mode: " + std::to_string((long long)w[it]) + " peer rank(s) failed in or before iteration " + std::to_string(head_flow ? it : iters - 1) + "; the pose is not valid";
                    identity16(out->T); out->status = SLAM3D_DEGENERATE;
                    rc = SLAM3D_E_COMM;
                    break;
                }
        }
    }
    (void)slam3d_icp_dense_set_rows(h, 0, h->p.height);
    return rc;
}

extern "C" int slam3d_icp_set_fault_injection(slam3d_icp_handle *h, int32_t dense_fail_at)
{
    if (!h) return SLAM3D_E_INVALID;
    if (dense_fail_at >= 2000 && dense_fail_at < 2256) { h->ls_test_stall_it = dense_fail_at - 2000; return SLAM3D_OK; }      // the point-list watchdog's hook
    h->ls_test_stall_it = -1;
    h->dense_fail_at = dense_fail_at;
    return SLAM3D_OK;
}

extern "C" int slam3d_icp_dense_run_with(slam3d_icp_handle *h, int32_t rank, int32_t world, slam3d_allreduce_fn allreduce, void *ctx,
                                         const double *T_init, slam3d_icp_result *out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce)) return SLAM3D_E_INVALID;
    DenseExchange ex = { allreduce, ctx };
    int rc = dense_run_impl(h, rank, world, allreduce ? &ex : nullptr, T_init, out);
    if (rc == SLAM3D_E_COMM - 100) rc = SLAM3D_E_COMM;
    return rc;
}

extern "C" int slam3d_icp_dense_run(slam3d_icp_handle *h, slam3d_comm *comm, const double *T_init, slam3d_icp_result *out)
{
    if (!h || !out) return SLAM3D_E_INVALID;
    if (comm && comm->device != h->p.device) return SLAM3D_E_INVALID;
    if (comm && !comm->comm) { h->err = "slam3d_icp_dense_run: the communicator was aborted by an earlier failure; create a new one"; return SLAM3D_E_COMM; }
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    const bool collective = comm && (world > 1 || getenv("SLAM3D_DENSE_FORCE_COLLECTIVE"));
    DenseExchange ex = { rccl_allreduce_thunk, comm };
    int rc = dense_run_impl(h, rank, world, collective ? &ex : nullptr, T_init, out);
    if (rc == SLAM3D_E_COMM - 100 || (rc == SLAM3D_E_COMM && comm && !comm->err.empty() && h->err.find("peer rank") == std::string::npos)) {
        // the exchanges themselves failed (or could not be completed after a local failure): release this rank's side; the
        // slam3d_comm is dead from here on.  Whether the peers' pending collectives then return is up to RCCL's transport.
        if (collective && world > 1 && s3d::rccl().CommAbort && comm->comm) {
            (void)s3d::rccl().CommAbort(comm->comm);
            comm->comm = nullptr;
            h->err += " (communicator aborted)";
        }
        rc = SLAM3D_E_COMM;
    }
    return rc;
}

#ifdef VOX_DBG
extern "C" int slam3d_debug_vox_phases(long long *out, int n_ll) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s3d::g_vox_dbg), sizeof(long long) * (size_t)n_ll); }
#endif

#ifdef SEGC_DBG
extern "C" int slam3d_debug_segc_phases(long long *out, int n_ll) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(s3d::g_segc_dbg), sizeof(long long) * (size_t)n_ll); }
#endif

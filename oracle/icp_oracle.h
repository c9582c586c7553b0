/*
 * icp_oracle.h -- CPU ORACLE for the plane-ICP registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (slam3d_gx_amd/,
 * include/) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the reference (gaoxiang12/slam3d_gx) contains no ICP of its
 * own and its PCL/OpenCV dependencies cannot be built
 * here (F3), so no golden vector of the reference pins this oracle.  What IS
 * pinned against reference artefacts:
 *   - orc_backproject vs data/exp1/pcd/{1,2}.pcd (written by the reference's
 *     src/convert2PCD.cpp:54-72) -- tests/test_golden.py::test_backprojection_pinned_by_reference_pcd
 * What hardens the rest although the reference cannot pin it: an INDEPENDENT numpy / scipy restatement of
 * the iteration (cKDTree candidates + canonical float32 re-evaluation for the indices, numpy lstsq on the
 * explicit [p x n, n] rows, numpy SVD for Kabsch, numpy eigh for the normals) must give the same indices
 * exactly and the same sums / poses to the fixed-point bound / 1e-9 at every iterate --
 * tests/test_oracle_independent.py, tests/golden/make_independent_golden.py.
 * Everything else follows the normative restatement in DESIGN.md section 3
 * (SURVEY.md App. C), which re-uses the reference's conventions:
 *   - pinhole back-projection         src/convert2PCD.cpp:65-69,
 *                                     src/GraphicEnd.cpp:452-455,
 *                                     src/planarFeatures.cpp:108-111
 *   - PassThrough z in (0, z_filter]  src/GraphicEnd.cpp:283-285
 *   - 7x7 organized patch, ">40" planar inliers at 0.01 m
 *                                     src/planarFeatures.cpp:92,118-128
 *   - pose direction / norm / thresholds of multiPnP
 *                                     src/GraphicEnd.cpp:557-659 (:599,:618,:621)
 *   - pose-error metric               src/exp1/exp1_2.cpp:167-171,285-289
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSUMS 29          /* 21 upper-tri AtA + 6 Atb + count + sum r^2 (derived from the Gram matrix)  */
#define ORC_NRAW  36          /* upper triangle of the 8x8 integer Gram matrix of the quantised row vectors */
#define ORC_CHUNK 256         /* (historic) launch block of four tiles            */

/* ORC_EST_PLANE (round 5, spec S2p): the point-to-plane rows of ORC_EST_POINT2PLANE with the target normals taken from the
 * frame's PLANES instead of its 7x7 windows: the target frame is segmented (seg_oracle.c, P1-P5), every labelled pixel takes
 * its plane's (a, b, c), unlabelled pixels are no targets (SURVEY.md App. C2 "points take their plane's normal";
 * src/GraphicEnd.cpp:353-430 extracts the planes per frame, :557-659 derives the pose from plane-wise correspondences). */
enum { ORC_EST_POINT2PLANE = 0, ORC_EST_SVD = 1, ORC_EST_PLANE = 2 };
enum { ORC_NN_BRUTE = 0, ORC_NN_KDTREE = 1 };
enum { ORC_OK = 0, ORC_TOO_FEW_INLIERS = 1, ORC_NORM_EXCEEDED = 2, ORC_DEGENERATE = 3 };

typedef struct orc_params {
    int    width, height;
    double fx, fy, cx, cy, depth_factor;  /* back-projection intrinsics            */
    double z_filter;                      /* validity: 0 < z <= z_filter           */
    int    iterations;                    /* fixed count, no early exit            */
    double max_corr_dist;                 /* gate on the NN distance (metres)      */
    int    estimator;                     /* ORC_EST_*                             */
    int    normal_window;                 /* odd, 7 in the reference's isPlanar    */
    int    normal_min_inliers;            /* 41  (= "> 40")                        */
    double normal_inlier_dist;            /* 0.01 m                                */
    int    min_inliers;                   /* 12, multiPnP default                  */
    double error_threshold;               /* 1.0, parameters.yaml:39               */
    int    nn_method;                     /* ORC_NN_*  (identical results)         */
    int    threads;                       /* OpenMP threads, <=0 -> all            */
    /* optional correspondence gates of the point-to-plane estimator (0 = off; spec S4g).  A correspondence that
     * passed the distance gate is dropped when
     *   max_plane_residual2 > 0 and e*e > max_plane_residual2, e = n.(q - p') the signed point-to-plane residual
     *     (the per-pixel test of src/GraphicEnd.cpp~:448-502, `e*=e; if (e > _min_error_plane)`, parameters.yaml:45);
     *   min_normal_cos > 0 and (R n_src).n_tgt < min_normal_cos, or the source pixel has no valid normal
     *     (the outlier-rejection role of solvePnPRansac's inlier subset, src/GraphicEnd.cpp:522-554).           */
    double max_plane_residual2;
    double min_normal_cos;
    /* spec S4c (round 4): the first coarse_iterations iterations -- never the last one of a run -- take only the sources of
     * every fourth 8x8-pixel tile, (tile_x + 2 tile_y) mod 4 == 0: while the pose still moves by centimetres a quarter of
     * the rows gives the same update, and those are the iterations whose searches are widest.  0 = every iteration uses
     * every source.  Default 3. */
    int    coarse_iterations;
    /* ---- ORC_EST_PLANE (spec S2p): the segmentation that yields the planes (defaults = parameters.yaml: 0.08 / 0.2 / 3; 64
     * hypotheses, seed 1), and the optional plane-PAIR gate (src/GraphicEnd.cpp:459-484,:572: PnP only
     * inside matched plane pairs): the source frame is segmented too, its planes are carried into the target frame by the
     * run's initial pose and associated with the target's planes exactly as GraphicEnd::match does (nearest (a, b, c, d));
     * a correspondence is kept iff the target pixel's plane is the one associated with the source pixel's plane (a pixel on no
     * plane: label -1, associated with -1 -- clutter only matches clutter).
     * plane_only: pixels on no plane are NOT targets (the literal per-plane variant of App. C2); default 0: they keep their 7x7
     * window normal, because two or three planes rarely constrain all six degrees of freedom. */
    float  seg_distance_threshold, seg_plane_percent;
    int    seg_max_planes, seg_hypotheses;
    uint64_t seg_seed;
    int    plane_pair_gate;
    int    plane_only;
} orc_params;

typedef struct orc_result {
    double T[16];        /* row-major 4x4, X_target = T * X_source */
    double norm;         /* |min(a, 2pi-a)| + 0.9*||t||            */
    double rmse;         /* sqrt(sum r^2 / count) at last iteration */
    int    inliers;      /* gated correspondences at last iteration */
    int    status;       /* ORC_*                                   */
    int    iterations;   /* iterations actually executed            */
    int    n_src, n_tgt; /* compacted list sizes                    */
} orc_result;

void orc_default_params(orc_params *p);

/* S1: u16 depth -> organized float4 cloud {x,y,z,1}; invalid pixels = NaN,NaN,NaN,0 */
void orc_backproject(const uint16_t *depth, const orc_params *p, float *xyz4);

/* S2: per-pixel normals of an organized cloud; out {nx,ny,nz,1} or {0,0,0,0} */
void orc_normals(const float *xyz4, const orc_params *p, float *nrm4);

/* S2p: per-plane normals of an organized cloud: out {a,b,c, 1 + plane} for a pixel labelled with plane r; a pixel on no plane
 * keeps its S2 window normal as {nx,ny,nz, 0.75} (plane_only: {0,0,0,0}); {0,0,0,0} where there is neither;
 * planes8 (seg_max_planes * 8: a b c d cx cy cz count) and labels (N) nullable; returns the number of planes */
int orc_plane_normals(const float *xyz4, const orc_params *p, float *nrm4, float *planes8, int32_t *labels);
/* plane association of the pair gate: planes of frame 1 (n1 x 8 floats, a b c d first) carried by T (row-major 4x4, nullable =
 * Identity) and matched to the planes of frame 2; assoc[i] = index in frame 2 or -1 */
void orc_plane_assoc(const float *planes1, int n1, const float *planes2, int n2, const double *T, int32_t *assoc);

/* S3..S6: full ICP.  src4/tgt4 organized float4 clouds (w ignored).
 * idx_out[N] (original linear indices, -1 none) / d2_out[N] of the LAST iteration,
 * T_trace[(iterations+1)*16], sums_trace[iterations*29] -- all nullable. */
int orc_icp(const float *src4, const float *tgt4, const orc_params *p,
            const double *T_init, orc_result *res,
            int32_t *idx_out, float *d2_out,
            double *T_trace, double *sums_trace);

/* one NN pass only (for index-parity tests / CPU baseline timing):
 * T (row-major 4x4 double) applied to src, search in tgt (point validity only when
 * use_normals==0, else tgt normal validity too). */
/* spec S4 (round 4): exponent of the residual component, the Gram accumulation of one row vector, and the 29 derived sums */
int  orc_b_exponent(double max_corr_dist);
void orc_accumulate_gram(const int64_t v[8], int64_t G[ORC_NRAW]);
void orc_derive_sums(const int64_t G[ORC_NRAW], int estimator, int eb, double s[ORC_NSUMS]);

int orc_nn_once(const float *src4, const float *tgt4, const orc_params *p,
                const double *T, int use_normals, int32_t *idx_out, float *d2_out);
int orc_nn_once_ex(const float *src4, const float *tgt4, const orc_params *p, const double *T,
                   int use_normals, int coarse, int32_t *idx_out, float *d2_out);

/* per-plane fit (row a6): labels[N] in {-1,0..nplanes-1}; out planes[nplanes*4]=(a,b,c,d), d>=0
 * counts[nplanes] */
void orc_fit_planes(const float *xyz4, const int32_t *labels, int n, int nplanes,
                    float *planes, int32_t *counts);

/* batched plane segmentation (row f-2, seg_oracle.c): labels[n] out: -2 invalid pixel, -1 valid on no plane,
 * r = plane r; planes[max_planes*8] = a b c d cx cy cz count; returns the number of planes */
#define ORC_SEG_DRAWS 32
typedef struct {
    float distance_threshold;   /* parameters.yaml distance_threshold 0.08 (src/GraphicEnd.cpp:365) */
    float plane_percent;        /* plane_percent 0.2 (:372) */
    int32_t max_planes;         /* max_planes 3 (:424) */
    int32_t hypotheses;         /* per round, <= 64 */
    uint64_t seed;
} orc_seg_params;
int orc_segment_planes(const float *xyz4, int n, float zmax, const orc_seg_params *sp, float *planes,
                       int32_t *labels);

/* frame ingestion filters (row f-1, voxel_oracle.c): PassThrough z in [0,zmax] + VoxelGrid(leaf) on n records
 * {x,y,z,rgba bits}; out has room for n records; returns the number of voxels (ascending (iz,iy,ix)) */
int orc_voxel_grid(const float *pts, int n, float leaf, float zmax, float *out);
int orc_voxel_grid_range(const float *pts, int n, float leaf, float zmin, float zmax, float *out);
/* keyframe cloud merge of src/saveOutput.cpp:84-92: PassThrough + rigid transform, dropped records -> NaN; returns kept */
int orc_pass_transform(const float *pts, int n, float zmax, const double *T, float *out);
uint64_t orc_voxel_key(float x, float y, float z, float inv_leaf);

/* pose error metric a14: E = Tref^-1 * T ; trans = ||E_t||, rot = acos(clamp((tr-1)/2)) */
void orc_pose_error(const double *Tref, const double *T, double *rot_err, double *trans_err);

/* small solvers exposed for property tests */
void orc_eig3(const double A[6] /*xx,xy,xz,yy,yz,zz*/, double evals[3], double evecs[9] /*cols*/);
int  orc_smallest_evec3(const double C[6] /*xx,xy,xz,yy,yz,zz*/, double n[3]);   /* spec S2: 1 = unit vector written, 0 = no direction */
int  orc_solve6(const double AtA21[21], const double Atb[6], double x[6]);
void orc_svd3_rotation(const double H[9], double R[9]);
void orc_sincos(double x, double *s, double *c);

#ifdef __cplusplus
}
#endif
#endif

// GraphicEndICP.h -- host-side mirror of the reference's front end for the plane-ICP path.
//
// The reference's plug-in seam is virtual-method override of GraphicEnd (src/GraphicEnd.h:74-216); its
// second front end GraphicEnd2 (src/GraphicEnd.h:262-275, src/GraphicEnd2.cpp) overrides
// init/run/readimage/multiPnP and main() picks the subclass (src/run_SLAM.cpp:21 vs
// src/run_SLAM_imageonly.cpp:21).  GraphicEndICP has the same method names, defaults and result type, but
// its frames are organized depth images (kept on the GPU as organized clouds) and multiPnP forwards to the
// C-ABI slam3d_icp_* (include/slam3d_icp.h).  It is self-contained (no PCL / OpenCV / g2o / Eigen): those
// are not available here, and PLANE/KEYFRAME carry OpenCV members (src/GraphicEnd.h:41-57).  INTEGRATION.md
// shows the ~40-line subclass that plugs the same calls into the real GraphicEnd.
#pragma once
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/slam3d_icp.h"
#include "ParameterReader.h"
#include "PoseGraph.h"
#include "pcd_io.h"

// RESULT_OF_MULTIPNP {T, norm, inliers} (src/GraphicEnd.h:59-69); T row-major instead of Eigen::Isometry3d
struct RESULT_OF_MULTIPNP {
    RESULT_OF_MULTIPNP();
    double T[16];
    double norm;
    int inliers;
    // what an ICP "inlier" count needs beside it to be discriminative: the reference's inliers were
    // descriptor matches that survived PnP RANSAC, an ICP inlier is any source pixel with a target within the gate
    double rmse;                      // point-to-plane RMS residual of the inliers at the last iteration
    int n_src;                        // valid source points: inliers / n_src is the overlap ratio
    bool isIdentity() const;          // the reference's failure test (src/GraphicEnd.cpp:173)
};

struct FRAME {                        // stands in for KEYFRAME / vector<PLANE> (src/GraphicEnd.h:51-57)
    int id = 0;
    int frame_index = 0;
    std::vector<uint16_t> depth;      // organized 16-bit depth, width*height
    std::vector<slam3d_plane> planes; // PLANE::coff of src/GraphicEnd.h:43 (filled when icp_extract_planes: yes)
    std::vector<int> connect;         // loop-closure partners, KEYFRAME::connect (src/GraphicEnd.cpp:760)
    std::vector<PointXYZRGBA16> cloud; // _currCloud after PassThrough + VoxelGrid (src/GraphicEnd.cpp:283-295), icp_read_pcd: yes
};

void mat4_identity(double *T);
void mat4_mul(const double *A, const double *B, double *C);
void mat4_inverse_rigid(const double *T, double *Ti);

class GraphicEndICP {
 public:
    GraphicEndICP();
    virtual ~GraphicEndICP();

    virtual void init(const std::string &param_file = "./parameters.yaml");   // src/GraphicEnd.cpp:77-148
    virtual int run();                                                         // src/GraphicEnd.cpp:150-264
    virtual int readimage();                                                   // src/GraphicEnd.cpp:266-302
    virtual void generateKeyFrame(const double *T, int frame_index = -1);      // src/GraphicEnd.cpp:304-351 (-1: the current _index)
    virtual void saveFinalResult(const std::string &fileaddr);                 // src/GraphicEnd.cpp:661-682
    virtual void loopClosure();                                                // src/GraphicEnd.cpp:685-762
    virtual void lostRecovery();                                               // src/GraphicEnd.cpp:764-838
    virtual void findMoreLoops();                                              // src/GraphicEnd.cpp:868-890
    virtual bool check(int frame1, int frame2);                                // src/GraphicEnd.cpp:892-917
    virtual std::vector<int> checknearby(int source, int target);              // src/GraphicEnd.cpp:919-947
    // plane list of a frame: the SACSegmentation loop of extractPlanesAndGenerateImage (src/GraphicEnd.cpp:353-430)
    virtual std::vector<slam3d_plane> extractPlanes(const FRAME &frame);
    // DMatch list of GraphicEnd::match(vector<PLANE>&, vector<PLANE>&) (src/GraphicEnd.cpp:459-484): trainIdx per plane of p1
    virtual std::vector<int> match(const std::vector<slam3d_plane> &p1, const std::vector<slam3d_plane> &p2);
    // same call shape and defaults as GraphicEnd::multiPnP (src/GraphicEnd.h:134)
    virtual RESULT_OF_MULTIPNP multiPnP(FRAME &frame1, FRAME &frame2, bool loopclosure = false, int frame_index = 0,
                                        int minimum_inliers = 12);
    // loop-closure candidates are independent pairs (src/GraphicEnd.cpp:685-762): one batched launch per GPU, the
    // batch dealt over the GPUs in contiguous blocks (slam3d_shard_range), one host thread per GPU
    // T_init: optional initial guesses, 16 doubles per pair (row-major, X_frame2 = T X_frame1); null = Identity
    std::vector<RESULT_OF_MULTIPNP> multiPnPBatch(const std::vector<const FRAME *> &f1, const std::vector<const FRAME *> &f2,
                                                  int minimum_inliers = 12, bool loopclosure = false, const double *T_init = nullptr);
    // plane-association gate (rows a9/a11): the planes of frame1 carried into frame2 by T must meet planes of frame2
    // (GraphicEnd::match on (a,b,c,d), src/GraphicEnd.cpp:459-484); true when the gate is off or a frame has no planes
    virtual bool planeGate(const FRAME &frame1, const FRAME &frame2, const double *T);
    int deviceCount() const { return (int)_devs.size(); }

    int index() const { return _index; }
    const double *robot() const { return _robot; }
    const std::vector<FRAME> &keyframes() const { return _keyframes; }
    int lostCount() const { return _lost; }
    const PoseGraph &graph() const { return _graph; }
    int moreLoops() const { return _moreLoops; }

 protected:
    ParameterReader *_reader = nullptr;
    slam3d_icp_handle *_icp = nullptr;               // = _devs[0].icp: the single-frame calls (planes, voxel grid)
    // one handle per GPU (hip_devices); each keeps the depth frames it has seen RESIDENT (frame id = frame_index):
    // a keyframe is uploaded and preprocessed once, not once per multiPnP call
    struct Device {
        slam3d_icp_handle *icp = nullptr;
        slam3d_icp_handle *icp_list = nullptr;       // icp_cloud: voxel -- a point-list handle (height 1) for readimage's voxel clouds
        int device = 0, first_frame = 0;
        std::vector<int> key;                        // resident slot -> frame_index (-1 free)
        std::vector<unsigned long long> used;        // LRU stamps
        unsigned long long clock = 0;
    };
    std::vector<Device> _devs;
    int residentFrame(Device &d, const FRAME &f, unsigned long long pin);
    void alignOnDevice(Device &d, const std::vector<const FRAME *> &f1, const std::vector<const FRAME *> &f2, int b0, int b1,
                       int minimum_inliers, bool loopclosure, std::vector<RESULT_OF_MULTIPNP> &out, const double *T_init);
    // icp_motion_model: the pose of the previous frame against the SAME keyframe starts the present frame's alignment
    // (an ICP needs a starting pose where the reference's PnP did not; the first iterations of a run, whose neighbours are
    // centimetres away, are the expensive ones).  Off by default; reset whenever the keyframe changes.
    bool _motion_model = false, _have_guess = false;
    double _T_guess[16];
    double _min_inlier_ratio = 0.3, _max_rmse = 0.05;             // tracking gates (icp_min_inlier_ratio, icp_max_rmse)
    double _loop_min_inlier_ratio = 0.6, _loop_max_rmse = 0.02;   // loop-closure gates (icp_loop_min_inlier_ratio, icp_loop_max_rmse)
    bool _plane_gate = false; double _plane_match_dist = 0.15;    // icp_plane_gate, icp_plane_match_dist
    slam3d_icp_params _params;
    std::string _depPath;
    std::ofstream _errorfile, _trajfile;
    int _index = 0, _start_index = 1, _end_index = 1, _lost = 0, _lost_frames = 10, _max_batch = 1;
    double _max_pos_change = 0.25, _error_threshold = 1.0, _loop_closure_error = 1.5;
    int _loop_closure_inliers = 30;
    FRAME _present, _currKF, _last;
    std::vector<FRAME> _keyframes;
    std::vector<std::vector<double> > _kf_poses;     // keyframe poses (row-major 4x4), what the reference keeps in g2o
    double _robot[16], _kf_pos[16], _graph_pose[16];
    void writeTrajectoryLine(int frame_index, const double *T);
    // accept test shared by loopClosure / lostRecovery / check (:701-706): not Identity, norm, inliers
    bool acceptLoop(const RESULT_OF_MULTIPNP &r) const;
    PoseGraph _graph;
    bool _loop_closure_detection = false, _extract_planes = false;
    int _loopclosure_frames = 30, _moreLoops = 0;
    unsigned long long _lc_state = 1;                 // counter-based PRNG replacing rand() (src/GraphicEnd.cpp:69,725)
    slam3d_seg_params _seg;
    std::ofstream _lcfile, _planefile;
    bool _read_pcd = false;
    bool _cloud_voxel = false;        // icp_cloud: voxel -- multiPnP aligns readimage's voxel clouds (point lists), as the reference hands them on (src/GraphicEnd.cpp:158)
    int _cloud_max_points = 32768;    // icp_cloud_max_points: capacity of the point-list handle
    std::string _pclPath;
    float _grid_leaf = 0.03f;
};

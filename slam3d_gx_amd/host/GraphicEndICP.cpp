// GraphicEndICP.cpp -- see GraphicEndICP.h.  Control flow follows GraphicEnd::run (src/GraphicEnd.cpp:150-264);
// the pose arithmetic lives behind the C-ABI (HIP kernels), never here.
#include "GraphicEndICP.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <thread>

#include "png16.h"

using namespace std;

RESULT_OF_MULTIPNP::RESULT_OF_MULTIPNP() : norm(0.0), inliers(0), rmse(0.0), n_src(0) { mat4_identity(T); }

bool RESULT_OF_MULTIPNP::isIdentity() const
{
    for (int k = 0; k < 16; ++k)
        if (T[k] != ((k % 5 == 0) ? 1.0 : 0.0)) return false;
    return true;
}

void mat4_identity(double *T) { for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0; }

void mat4_mul(const double *A, const double *B, double *C)
{
    double R[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[r * 4 + k] * B[k * 4 + c];
            R[r * 4 + c] = s;
        }
    memcpy(C, R, sizeof R);
}

void mat4_inverse_rigid(const double *T, double *Ti)
{
    double R[16];
    mat4_identity(R);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 4 + c] = T[c * 4 + r];
    for (int r = 0; r < 3; ++r) R[r * 4 + 3] = -(R[r * 4] * T[3] + R[r * 4 + 1] * T[7] + R[r * 4 + 2] * T[11]);
    memcpy(Ti, R, sizeof R);
}

GraphicEndICP::GraphicEndICP()
{
    mat4_identity(_robot);
    mat4_identity(_kf_pos);
    mat4_identity(_graph_pose);
    slam3d_icp_default_params(&_params);
}

GraphicEndICP::~GraphicEndICP()
{
    for (size_t k = 0; k < _devs.size(); ++k) {
        if (_devs[k].icp_list) slam3d_icp_destroy(_devs[k].icp_list);
        if (_devs[k].icp) slam3d_icp_destroy(_devs[k].icp);
    }
    delete _reader;
    if (g_pParaReader == _reader) g_pParaReader = nullptr;
}

void GraphicEndICP::init(const string &param_file)
{
    _reader = new ParameterReader(param_file);
    g_pParaReader = _reader;
    // same keys as GraphicEnd::init (src/GraphicEnd.cpp:82-104)
    _start_index = _reader->GetInt("start_index", 1);
    _end_index = _reader->GetInt("end_index", 1);
    _index = _start_index;
    const string source = _reader->GetPara("data_source");
    _depPath = source + "/dep_index/";
    _max_pos_change = _reader->GetDouble("max_pos_change", 0.25);
    _error_threshold = _reader->GetDouble("error_threshold", 1.0);
    _lost_frames = _reader->GetInt("lost_frames", 10);
    _loop_closure_error = _reader->GetDouble("loop_closure_error", 1.5);
    _loop_closure_inliers = _reader->GetInt("loop_closure_inliers", 30);
    _loop_closure_detection = _reader->Has("loop_closure_detection") && _reader->GetPara("loop_closure_detection") == "yes";
    _loopclosure_frames = _reader->GetInt("loopclosure_frames", 30);
    _lc_state = (unsigned long long)_reader->GetInt("loopclosure_seed", 1);
    _read_pcd = _reader->Has("icp_read_pcd") && _reader->GetPara("icp_read_pcd") == "yes";
    _pclPath = source + "/pcd/";                                                       // src/GraphicEnd.cpp:85
    _grid_leaf = (float)_reader->GetDouble("grid_leaf", 0.03);                         // :288
    // icp_cloud: depth (default) -- multiPnP aligns the organized depth frames;  voxel -- it aligns the cloud readimage made of the
    // frame (PCD -> PassThrough -> VoxelGrid, ~15 k points; needs icp_read_pcd: yes), which is what the reference hands on
    // (src/GraphicEnd.cpp:279-295 -> :158): point lists, svd estimator, one persistent launch per alignment (csrc/list_icp.hpp)
    _cloud_voxel = _reader->Has("icp_cloud") && _reader->GetPara("icp_cloud") == "voxel";
    _cloud_max_points = _reader->GetInt("icp_cloud_max_points", 32768);
    if (_cloud_voxel && !_read_pcd) { cerr << "icp_cloud: voxel needs icp_read_pcd: yes (the voxel cloud is made from the frame's PCD file)" << endl; exit(1); }
    _extract_planes = _reader->Has("icp_extract_planes") && _reader->GetPara("icp_extract_planes") == "yes";
    slam3d_seg_default_params(&_seg);
    _seg.distance_threshold = (float)_reader->GetDouble("distance_threshold", 0.08);   // src/GraphicEnd.cpp:89
    _seg.plane_percent = (float)_reader->GetDouble("plane_percent", 0.2);              // :93
    _seg.max_planes = _reader->GetInt("max_planes", 3);                                // :95
    _seg.hypotheses = _reader->GetInt("ransac_hypotheses", 64);
    if (_reader->Has("detector_name") && _reader->GetPara("detector_name") != "ICP")
        cout << "note: detector_name/descriptor_name are ignored by the ICP front end" << endl;

    slam3d_icp_default_params(&_params);
    _params.width = _reader->GetInt("image_width", 640);
    _params.height = _reader->GetInt("image_height", 480);
    _params.fx = camera_fx; _params.fy = camera_fy; _params.cx = camera_cx; _params.cy = camera_cy;
    _params.depth_factor = camera_factor;
    _params.z_filter = _reader->GetDouble("z_filter", 7.0);
    // new keys, defaults when absent
    _params.iterations = _reader->GetInt("icp_iterations", 20);
    _params.max_corr_dist = _reader->GetDouble("icp_max_corr_dist", 0.10);
    // icp_estimator: point2plane (7x7-window normals) | svd | plane -- plane-ICP proper: the frames' planes give the normals, as the
    // reference derives the pose from the planes it extracts per frame (src/GraphicEnd.cpp:158,168,557-659); with
    // icp_plane_pair_gate: yes correspondences are kept only inside associated plane pairs (src/GraphicEnd.cpp:459-484,:572),
    // icp_plane_only: yes drops the pixels on no plane from the targets
    const string est_name = _reader->Has("icp_estimator") ? _reader->GetPara("icp_estimator") : string("point2plane");
    _params.estimator = est_name == "svd" ? SLAM3D_EST_SVD : (est_name == "plane" ? SLAM3D_EST_PLANE : SLAM3D_EST_POINT2PLANE);
    if (_params.estimator == SLAM3D_EST_PLANE) {
        if (_reader->Has("icp_plane_pair_gate") && _reader->GetPara("icp_plane_pair_gate") == "yes") _params.plane_flags |= SLAM3D_PLANE_PAIR_GATE;
        if (_reader->Has("icp_plane_only") && _reader->GetPara("icp_plane_only") == "yes") _params.plane_flags |= SLAM3D_PLANE_ONLY;
    }
    _params.normal_window = _reader->GetInt("icp_normal_window", 7);
    // optional correspondence gates, off unless asked for:
    //   icp_plane_residual_gate: yes -> e^2 <= min_error_plane, the reference's own key and test (src/GraphicEnd.cpp~:484-489,
    //                            parameters.yaml:45; src/GraphicEnd.cpp:91 reads it and never uses it)
    //   icp_normal_angle_deg: d  -> angle(R n_src, n_tgt) <= d degrees (role of the RANSAC inlier subset, src/GraphicEnd.cpp:542)
    if (_reader->Has("icp_plane_residual_gate") && _reader->GetPara("icp_plane_residual_gate") == "yes")
        _params.max_plane_residual2 = (float)_reader->GetDouble("min_error_plane", 0.02);
    {
        const double deg = _reader->GetDouble("icp_normal_angle_deg", 0.0);
        if (deg > 0.0 && deg < 90.0) _params.min_normal_cos = (float)cos(deg * M_PI / 180.0);
    }
    _params.min_inliers = _reader->GetInt("icp_min_inliers", 12);
    _params.coarse_iterations = _reader->GetInt("icp_coarse_iterations", 3);     // spec S4c: the per-frame call has no initial guess (src/GraphicEnd.cpp:168)
    _params.error_threshold = _error_threshold;
    _max_batch = _loopclosure_frames + 2;              // random candidates + the two adjacent keyframes, one launch
    if (_max_batch < 1) _max_batch = 1;
    _params.max_batch = _max_batch;
    // ICP-meaningful acceptance: overlap ratio inliers / n_src and the RMS residual, beside the reference's
    // inlier-count / norm thresholds; 0 switches a gate off
    _min_inlier_ratio = _reader->GetDouble("icp_min_inlier_ratio", 0.3);
    _max_rmse = _reader->GetDouble("icp_max_rmse", 0.05);
    _loop_min_inlier_ratio = _reader->GetDouble("icp_loop_min_inlier_ratio", 0.6);
    _loop_max_rmse = _reader->GetDouble("icp_loop_max_rmse", 0.02);
    _plane_gate = _reader->Has("icp_plane_gate") && _reader->GetPara("icp_plane_gate") == "yes";
    _motion_model = _reader->Has("icp_motion_model") && _reader->GetPara("icp_motion_model") == "yes";
    _plane_match_dist = _reader->GetDouble("icp_plane_match_dist", 0.15);
    // one handle per GPU: hip_devices = N (or "all"), starting at hip_device; resident frame slots beyond the batch
    const int first_dev = _reader->GetInt("hip_device", 0);
    int ndev = 1;
    if (_reader->Has("hip_devices"))
        ndev = _reader->GetPara("hip_devices") == "all" ? slam3d_device_count() - first_dev : _reader->GetInt("hip_devices", 1);
    if (ndev < 1) ndev = 1;
    _params.extra_frames = 2 * _max_batch + 8;     // a chunk of max_batch pairs can name 2 * max_batch distinct frames (all pinned), + room for keyframes to stay
    // hip_devices_share: yes (tests): all handles on the first GPU -- the threaded sharding path on a one-GPU box
    const bool share = _reader->Has("hip_devices_share") && _reader->GetPara("hip_devices_share") == "yes";
    for (int k = 0; k < ndev; ++k) {
        Device d;
        d.device = share ? first_dev : first_dev + k;
        _params.device = d.device;
        const int rc = slam3d_icp_create(&_params, &d.icp);
        if (rc != SLAM3D_OK) {
            cerr << "slam3d_icp_create (device " << d.device << ") failed: " << slam3d_strerror(rc) << endl;
            if (k == 0) exit(1);                                   // the reference exits on fatal config errors (:113)
            break;                                                 // fewer GPUs than asked for: go on with what there is
        }
        if (_params.estimator == SLAM3D_EST_PLANE) {     // the library segments with the reference's plane parameters (fixed seed: a frame's planes do not depend on who aligns it)
            slam3d_seg_params sp = _seg;
            sp.seed = 1;
            // the ICP's own segmentation threshold (spec S2p) -- not the plane-extraction key distance_threshold (0.08, src/GraphicEnd.cpp:89)
            sp.distance_threshold = (float)_reader->GetDouble("icp_seg_distance_threshold", 0.04);
            const int src = slam3d_icp_set_seg_params(d.icp, &sp);
            if (src != SLAM3D_OK) {      // e.g. max_planes / ransac_hypotheses above the library's limits: the library would segment with ITS defaults
                cerr << "slam3d_icp_set_seg_params failed: " << slam3d_strerror(src) << " (" << slam3d_last_error(d.icp) << "); plane parameters of parameters.yaml are out of the library's range" << endl;
                exit(1);                 // while extractPlanes used _seg -- a fatal config error like the reference's (:113)
            }
        }
        if (_cloud_voxel) {
            slam3d_icp_params lp = _params;
            lp.width = _cloud_max_points; lp.height = 1;
            lp.estimator = SLAM3D_EST_SVD; lp.plane_flags = 0;                 // a list has no 7x7 windows: point-to-point (Kabsch)
            lp.max_plane_residual2 = 0.0f; lp.min_normal_cos = 0.0f;
            lp.nn_mode = SLAM3D_NN_AUTO;
            const int lrc = slam3d_icp_create(&lp, &d.icp_list);
            if (lrc != SLAM3D_OK) { cerr << "slam3d_icp_create (point lists, device " << d.device << ") failed: " << slam3d_strerror(lrc) << endl; exit(1); }
        }
        d.first_frame = 2 * _max_batch;
        d.key.assign(_params.extra_frames, -1);
        d.used.assign(_params.extra_frames, 0);
        _devs.push_back(d);
    }
    _icp = _devs[0].icp;
    cout << "ICP front end on " << _devs.size() << " GPU(s)" << endl;
    _errorfile.open("./data/error_of_transform.log");              // src/GraphicEnd.cpp:153
    _trajfile.open("./data/trajectory_icp.txt");
    _lcfile.open("./data/lc.txt");                                 // displayLC, src/GraphicEnd.cpp:843
    if (_extract_planes) _planefile.open("./data/planes.txt");

    // first frame = keyframe 0 at the origin (src/GraphicEnd.cpp:106-145)
    readimage();
    _currKF = _present;
    _currKF.id = 0;
    _currKF.frame_index = _index;
    _keyframes.push_back(_currKF);
    _kf_poses.push_back(vector<double>(_kf_pos, _kf_pos + 16));
    _graph.addVertex(0, _kf_pos, true);                            // the first vertex is fixed (src/SLAMEnd: setFixed(true))
    _last = _present;
    writeTrajectoryLine(_index, _robot);
    _index++;
}

vector<slam3d_plane> GraphicEndICP::extractPlanes(const FRAME &frame)
{
    // depth -> organized cloud on the GPU -> batched RANSAC segmentation (both behind the C-ABI)
    vector<float> xyz((size_t)_params.width * _params.height * 4);
    vector<slam3d_plane> planes(_seg.max_planes), out;
    if (slam3d_backproject_u16(_icp, frame.depth.data(), xyz.data()) != SLAM3D_OK) return out;
    slam3d_cloud_view v = { xyz.data(), 16, _params.width, _params.height };
    int n = 0;
    _seg.seed = (uint64_t)frame.frame_index + 1;
    if (slam3d_segment_planes(_icp, &v, &_seg, planes.data(), &n, nullptr) != SLAM3D_OK) return out;
    out.assign(planes.begin(), planes.begin() + n);
    cout << "Total planes: " << n << endl;                         // src/GraphicEnd.cpp:428
    return out;
}

int GraphicEndICP::readimage()
{
    cout << "loading image " << _index << endl;
    stringstream ss;
    ss << _depPath << _index << ".png";
    int w = 0, h = 0;
    string err;
    _present.depth.clear();
    if (!read_png_gray16(ss.str(), w, h, _present.depth, err) || w != _params.width || h != _params.height) {
        cerr << "readimage: " << (err.empty() ? "unexpected image size" : err) << endl;
        _present.depth.assign((size_t)_params.width * _params.height, 0);
        _present.frame_index = _index;
        return -1;
    }
    _present.frame_index = _index;
    _present.connect.clear();
    _present.planes.clear();
    _present.cloud.clear();
    if (_read_pcd) {
        // the reference's cloud path: loadPCDFile -> PassThrough z in [0, z_filter] -> VoxelGrid(grid_leaf) (:279-295),
        // the two filters on the GPU behind slam3d_voxel_grid
        stringstream ps;
        ps << _pclPath << _index << ".pcd";
        vector<PointXYZRGBA16> raw;
        int pw = 0, ph = 0;
        string perr;
        if (!read_pcd(ps.str(), raw, pw, ph, perr)) cerr << "readimage: " << perr << endl;
        else if ((long long)raw.size() > (long long)_params.width * _params.height) cerr << "readimage: cloud larger than the image" << endl;
        else {
            _present.cloud.resize(raw.size());
            int m = 0;
            if (slam3d_voxel_grid(_icp, raw.data(), (int)raw.size(), _grid_leaf, _present.cloud.data(), &m) == SLAM3D_OK) {
                _present.cloud.resize((size_t)m);
                cout << "cloud " << raw.size() << " -> " << m << " points after PassThrough + VoxelGrid" << endl;
            } else {
                _present.cloud.clear();
            }
        }
    }
    if (_extract_planes) {
        _present.planes = extractPlanes(_present);
        _planefile << _index << " " << _present.planes.size();
        for (size_t k = 0; k < _present.planes.size(); ++k) {
            const slam3d_plane &p = _present.planes[k];
            _planefile << " " << p.coeff[0] << " " << p.coeff[1] << " " << p.coeff[2] << " " << p.coeff[3] << " " << p.count;
        }
        _planefile << endl;
    }
    cout << "load ok." << endl;
    return 0;
}

RESULT_OF_MULTIPNP GraphicEndICP::multiPnP(FRAME &frame1, FRAME &frame2, bool loopclosure, int /*frame_index*/,
                                           int minimum_inliers)
{
    vector<const FRAME *> a(1, &frame1), b(1, &frame2);
    return multiPnPBatch(a, b, minimum_inliers, loopclosure)[0];
}

// resident slot of frame f on device d (uploaded when absent; least recently used slot evicted, never one of the
// current batch: those carry the stamp `pin`)
int GraphicEndICP::residentFrame(Device &d, const FRAME &f, unsigned long long pin)
{
    int free_slot = -1, lru = -1;
    for (size_t k = 0; k < d.key.size(); ++k) {
        if (d.key[k] == f.frame_index) { d.used[k] = pin; return d.first_frame + (int)k; }
        if (d.key[k] < 0) { if (free_slot < 0) free_slot = (int)k; }
        else if (d.used[k] != pin && (lru < 0 || d.used[k] < d.used[lru])) lru = (int)k;
    }
    const int victim = free_slot >= 0 ? free_slot : lru;
    if (victim < 0) return -1;
    if (_cloud_voxel) {
        if (f.cloud.empty() || (int)f.cloud.size() > _cloud_max_points) return -1;          // no PCD for this frame, or more voxels than the handle holds
        const slam3d_cloud_view v = { f.cloud.data(), 16, (int32_t)f.cloud.size(), 1 };    // {x, y, z, rgba} records: xyz are read
        if (slam3d_icp_frame_set_cloud_host(d.icp_list, d.first_frame + victim, &v) != SLAM3D_OK) return -1;
    } else if (slam3d_icp_frame_set_depth_host(d.icp, d.first_frame + victim, f.depth.data()) != SLAM3D_OK) return -1;
    d.key[victim] = f.frame_index;
    d.used[victim] = pin;
    return d.first_frame + victim;
}

void GraphicEndICP::alignOnDevice(Device &d, const vector<const FRAME *> &f1, const vector<const FRAME *> &f2, int b0, int b1,
                                  int minimum_inliers, bool loopclosure, vector<RESULT_OF_MULTIPNP> &out, const double *T_init)
{
    slam3d_icp_handle *icp = _cloud_voxel ? d.icp_list : d.icp;
    for (int c0 = b0; c0 < b1; c0 += _max_batch) {
        const int nb = min(_max_batch, b1 - c0);
        const unsigned long long pin = ++d.clock;
        bool ok = true;
        for (int k = 0; k < nb && ok; ++k) {
            const int fs = residentFrame(d, *f1[c0 + k], pin), ft = residentFrame(d, *f2[c0 + k], pin);
            ok = fs >= 0 && ft >= 0 && slam3d_icp_set_pair(icp, k, fs, ft) == SLAM3D_OK;
        }
        vector<slam3d_icp_result> res(nb);
        int rc = ok ? slam3d_icp_run(icp, nb, T_init ? T_init + (size_t)c0 * 16 : nullptr, nullptr) : SLAM3D_E_STATE;
        if (rc == SLAM3D_OK) rc = slam3d_icp_fetch_results(icp, nb, res.data());
        if (rc < 0) {
            cerr << "slam3d_icp (device " << d.device << "): " << slam3d_strerror(rc) << " " << slam3d_last_error(icp) << endl;
            continue;                                              // results stay Identity = "not matched"
        }
        const double min_ratio = loopclosure ? _loop_min_inlier_ratio : _min_inlier_ratio;
        const double max_rmse = loopclosure ? _loop_max_rmse : _max_rmse;
        for (int k = 0; k < nb; ++k) {
            RESULT_OF_MULTIPNP &r = out[c0 + k];
            r.inliers = res[k].inliers;
            r.norm = res[k].norm;
            r.rmse = res[k].rmse;
            r.n_src = res[k].n_src;
            // thresholds of multiPnP: inliers (src/GraphicEnd.cpp:599), norm (:621); plus what makes an ICP result
            // trustworthy: enough of the source overlaps (ratio) and the residual is at noise level (rmse)
            bool good = res[k].status == SLAM3D_OK && res[k].inliers >= minimum_inliers;
            if (good && min_ratio > 0.0 && r.n_src > 0 && (double)r.inliers < min_ratio * (double)r.n_src) good = false;
            if (good && max_rmse > 0.0 && r.rmse > max_rmse) good = false;
            if (good && !planeGate(*f1[c0 + k], *f2[c0 + k], res[k].T)) good = false;
            if (good) memcpy(r.T, res[k].T, sizeof r.T);
        }
    }
}

vector<RESULT_OF_MULTIPNP> GraphicEndICP::multiPnPBatch(const vector<const FRAME *> &f1, const vector<const FRAME *> &f2,
                                                        int minimum_inliers, bool loopclosure, const double *T_init)
{
    const int B = (int)f1.size();
    vector<RESULT_OF_MULTIPNP> out(B);
    const int ndev = (int)min<size_t>(_devs.size(), (size_t)max(B, 1));
    if (ndev <= 1) {
        alignOnDevice(_devs[0], f1, f2, 0, B, minimum_inliers, loopclosure, out, T_init);
    } else {
        // pairs are independent: contiguous blocks per GPU, one host thread each, no exchange but
        // the results (host memory shared by the threads -- the in-process counterpart of the RCCL pose gather)
        vector<thread> th;
        for (int k = 0; k < ndev; ++k) {
            int b0 = 0, b1 = 0;
            slam3d_shard_range(B, ndev, k, &b0, &b1);
            th.emplace_back([this, k, b0, b1, &f1, &f2, minimum_inliers, loopclosure, &out, T_init]() {
                alignOnDevice(_devs[k], f1, f2, b0, b1, minimum_inliers, loopclosure, out, T_init);
            });
        }
        for (size_t k = 0; k < th.size(); ++k) th[k].join();
    }
    for (int b = 0; b < B; ++b)
        cout << "multiICP::inliers = " << out[b].inliers << " / " << out[b].n_src << ", norm = " << out[b].norm << ", rmse = " << out[b].rmse
             << (out[b].isIdentity() ? " (rejected)" : "") << endl;
    return out;
}

bool GraphicEndICP::planeGate(const FRAME &frame1, const FRAME &frame2, const double *T)
{
    if (!_plane_gate || frame1.planes.empty() || frame2.planes.empty()) return true;
    // planes of frame 1 in frame-2 coordinates: X_2 = R X_1 + t  =>  n' = R n, d' = d - n'.t, sign rule d' >= 0 (:383-387)
    vector<slam3d_plane> moved(frame1.planes);
    for (size_t i = 0; i < moved.size(); ++i) {
        const float *c = frame1.planes[i].coeff;
        double n[3];
        for (int r = 0; r < 3; ++r) n[r] = (T[r * 4] * c[0] + T[r * 4 + 1] * c[1]) + T[r * 4 + 2] * c[2];
        double d = c[3] - ((n[0] * T[3] + n[1] * T[7]) + n[2] * T[11]);
        if (d < 0.0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; d = -d; }
        moved[i].coeff[0] = (float)n[0]; moved[i].coeff[1] = (float)n[1]; moved[i].coeff[2] = (float)n[2]; moved[i].coeff[3] = (float)d;
    }
    const vector<int> idx = match(moved, frame2.planes);           // GraphicEnd::match(vector<PLANE>&, vector<PLANE>&), :459-484
    int matched = 0;
    for (size_t i = 0; i < moved.size(); ++i) {
        if (idx[i] < 0) continue;
        double d2 = 0.0;
        for (int k = 0; k < 4; ++k) { const double e = (double)moved[i].coeff[k] - frame2.planes[idx[i]].coeff[k]; d2 += e * e; }
        if (sqrt(d2) <= _plane_match_dist) matched++;
    }
    return matched >= 1;
}

void GraphicEndICP::generateKeyFrame(const double *T, int frame_index)
{
    // T maps the current keyframe's pose to the present one (src/GraphicEnd.cpp:304-351); the g2o vertex/edge the
    // reference adds there becomes a stored pose here (pose-graph back end is out of scope)
    _currKF = _present;
    _currKF.id = (int)_keyframes.size();
    _currKF.frame_index = frame_index >= 0 ? frame_index : _index;      // (the last-frame branch stamps _index - 1, :198)
    // T = inverse of the multiPnP result = pose of this camera in the previous keyframe's frame, so the
    // camera-to-world chain is kf_pos * T -- the EdgeSE3 convention the g2o vertices below use.  (The reference
    // writes T * _kf_pos, :231,:245; there the value only feeds the viewer.)
    double P[16];
    mat4_mul(_kf_pos, T, P);
    memcpy(_kf_pos, P, sizeof P);
    _keyframes.push_back(_currKF);
    _kf_poses.push_back(vector<double>(_kf_pos, _kf_pos + 16));
    // VertexSE3 + EdgeSE3(previous keyframe -> this one, measurement T, information 100) (:319-338).  The
    // reference starts every vertex at Identity (:325); here the initial estimate is the chain of the edge
    // measurements (V_new = V_prev * T, the EdgeSE3 convention), so the file is consistent before optimisation.
    memcpy(_graph_pose, _kf_pos, sizeof _graph_pose);
    _graph.addVertex(_currKF.id, _graph_pose);
    _graph.addEdge(_currKF.id - 1, _currKF.id, T, 100.0);
}

vector<int> GraphicEndICP::match(const vector<slam3d_plane> &p1, const vector<slam3d_plane> &p2)
{
    vector<int> idx(p1.size(), -1);
    if (!p1.empty()) slam3d_match_planes(p1.data(), (int)p1.size(), p2.data(), (int)p2.size(), idx.data(), nullptr);
    return idx;
}

bool GraphicEndICP::acceptLoop(const RESULT_OF_MULTIPNP &r) const
{
    // the reference's three tests (:701-706); the overlap-ratio / rmse / plane gates were applied by multiPnPBatch
    // (loopclosure = true) and show up here as T == Identity
    return !r.isIdentity() && r.norm <= _loop_closure_error && r.inliers >= _loop_closure_inliers;
}

void GraphicEndICP::loopClosure()
{
    if (_keyframes.size() <= 3) return;                            // :687
    cout << "Checking loop closure." << endl;
    // candidates: the two adjacent keyframes (:694-722) and up to loopclosure_frames distinct random earlier
    // ones (:725-761).  They are independent pairs against the current keyframe: ONE batched launch.
    vector<int> cand;
    for (int i = -3; i > -5; i--) {
        const int n = (int)_keyframes.size() + i;
        if (n >= 0) cand.push_back(n);
    }
    const size_t n_adjacent = cand.size();
    vector<int> checked;
    for (int i = 0; i < _loopclosure_frames; i++) {
        _lc_state += 0x9E3779B97F4A7C15ull;
        unsigned long long z = _lc_state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const int frame = (int)(z % (unsigned long long)(_keyframes.size() - 3));
        bool seen = false;
        for (size_t k = 0; k < checked.size(); ++k) seen = seen || checked[k] == frame;
        if (seen) continue;                                        // :728
        checked.push_back(frame);
        cand.push_back(frame);
    }
    vector<const FRAME *> f1(cand.size()), f2(cand.size(), &_currKF);
    for (size_t k = 0; k < cand.size(); ++k) f1[k] = &_keyframes[cand[k]];
    vector<RESULT_OF_MULTIPNP> res = multiPnPBatch(f1, f2, _loop_closure_inliers, true);
    for (size_t k = 0; k < cand.size(); ++k) {
        if (!acceptLoop(res[k])) continue;
        double Ti[16];
        mat4_inverse_rigid(res[k].T, Ti);                          // :707
        _graph.addEdge(_keyframes[cand[k]].id, _currKF.id, Ti, 100.0, true);
        if (k >= n_adjacent) {
            _lcfile << _keyframes[cand[k]].frame_index << " " << _currKF.frame_index << " " << res[k].norm << " "
                    << res[k].inliers << endl;                     // displayLC :865
            _keyframes.back().connect.push_back(cand[k]);          // :760
        }
    }
}

void GraphicEndICP::lostRecovery()
{
    // the present frame becomes a keyframe with no edge to its predecessor (:764-838)
    cout << "Lost Recovery..." << endl;
    _currKF = _present;
    _currKF.id = (int)_keyframes.size();
    _currKF.frame_index = _index;
    memcpy(_kf_pos, _robot, sizeof _kf_pos);                       // :772
    ofstream fout("./data/lost.txt", ofstream::app);
    fout << _currKF.id << " " << _currKF.frame_index << endl;      // :774-776
    fout.close();
    _keyframes.push_back(_currKF);
    _kf_poses.push_back(vector<double>(_kf_pos, _kf_pos + 16));
    memcpy(_graph_pose, _kf_pos, sizeof _graph_pose);
    _graph.addVertex(_currKF.id, _graph_pose);                     // no edge to the predecessor: position unknown (:793)
    // against every earlier keyframe (:808-836): independent pairs, batched
    const int n = (int)_keyframes.size() - 1;
    vector<const FRAME *> f1(n), f2(n, &_currKF);
    for (int i = 0; i < n; ++i) f1[i] = &_keyframes[i];
    vector<RESULT_OF_MULTIPNP> res = multiPnPBatch(f1, f2, _loop_closure_inliers, true);
    for (int i = 0; i < n; ++i) {
        if (!acceptLoop(res[i])) continue;
        double Ti[16];
        mat4_inverse_rigid(res[i].T, Ti);
        _graph.addEdge(_keyframes[i].id, _currKF.id, Ti, 100.0, true);
        _keyframes.back().connect.push_back(i);
    }
    _lost = 0;
}

bool GraphicEndICP::check(int frame1, int frame2)
{
    cout << "checking " << frame1 << ", " << frame2 << endl;
    RESULT_OF_MULTIPNP result = multiPnP(_keyframes[frame1], _keyframes[frame2], true, _keyframes[frame1].frame_index,
                                         _loop_closure_inliers);
    if (!acceptLoop(result)) return false;
    double Ti[16];
    mat4_inverse_rigid(result.T, Ti);
    _graph.addEdge(_keyframes[frame1].id, _keyframes[frame2].id, Ti, 100.0, true);
    _moreLoops++;
    return true;
}

vector<int> GraphicEndICP::checknearby(int source, int target)
{
    vector<int> checked;
    int index = target;
    while (index > 0) {                                            // walk down until a pair fails (:925-934)
        index--;
        if (index == source) continue;
        if (check(source, index)) checked.push_back(index); else break;
    }
    index = target;
    while (index < (int)_keyframes.size() - 1) {                   // walk up (:936-945)
        index++;
        if (index == source) continue;
        if (check(source, index)) checked.push_back(index); else break;
    }
    return checked;
}

void GraphicEndICP::findMoreLoops()
{
    cout << "Find more loops" << endl;
    _moreLoops = 0;
    for (size_t i = 0; i < _keyframes.size(); i++) {
        if (_keyframes[i].connect.empty()) continue;
        for (size_t j = 0; j < _keyframes[i].connect.size(); j++) {
            vector<int> checked = checknearby((int)i, _keyframes[i].connect[j]);
            for (size_t k = 0; k < checked.size(); k++) checknearby(checked[k], (int)i);
        }
    }
    cout << "Total " << _moreLoops << " loops found. " << endl;
}

int GraphicEndICP::run()
{
    cout << "********************" << endl;
    readimage();
    // present -> current keyframe (src/GraphicEnd.cpp:168-170: the result is inverted by the caller)
    RESULT_OF_MULTIPNP result;
    if (_motion_model && _have_guess) {              // start from the previous frame's pose against this keyframe
        vector<const FRAME *> a(1, &_currKF), b(1, &_present);
        result = multiPnPBatch(a, b, 12, false, _T_guess)[0];
    } else
        result = multiPnP(_currKF, _present);
    double T[16];
    mat4_inverse_rigid(result.T, T);
    _have_guess = false;                              // only an ordinary tracked frame (last branch) leaves a guess behind
    if (result.isIdentity()) {
        _errorfile << "9999" << endl;                               // :176
        cout << "This frame lost" << endl;
        RESULT_OF_MULTIPNP r = multiPnP(_last, _present);           // :187
        if (r.isIdentity() || r.inliers < _loop_closure_inliers || r.norm > _loop_closure_error) {
            _lost++;
        } else {
            // the previous ordinary frame becomes a keyframe, then the present one (:193-228)
            _lost = 0;
            RESULT_OF_MULTIPNP rr = multiPnP(_currKF, _last);
            double Tl[16];
            mat4_inverse_rigid(rr.T, Tl);
            FRAME keep = _present;
            _present = _last;
            generateKeyFrame(Tl, _last.frame_index);                // _index - 1 in the reference (:198)
            _present = keep;
            double Tp[16];
            mat4_inverse_rigid(r.T, Tp);
            generateKeyFrame(Tp);
            memcpy(_robot, _kf_pos, sizeof _robot);
            _last = _present;
        }
    } else if (result.norm > _max_pos_change) {
        _errorfile << result.norm << endl;                          // :232
        mat4_mul(_kf_pos, T, _robot);
        generateKeyFrame(T);
        if (_loop_closure_detection) loopClosure();                // :236-237
        _lost = 0;
        _last = _present;
    } else {
        _errorfile << result.norm << endl;                          // :243
        mat4_mul(_kf_pos, T, _robot);
        _lost = 0;
        _last = _present;
        memcpy(_T_guess, result.T, sizeof _T_guess);
        _have_guess = true;
    }
    if (_lost > _lost_frames) {                                     // :250-255
        cerr << "the robot lost. Perform lost recovery." << endl;
        lostRecovery();
        _last = _present;
    }
    writeTrajectoryLine(_index, _robot);
    _index++;
    _present.frame_index = _index;
    return 1;
}

void GraphicEndICP::writeTrajectoryLine(int frame_index, const double *T)
{
    // TUM trajectory line "timestamp tx ty tz qx qy qz qw" (src/generateTrajectory.cpp:62-72)
    double qw, qx, qy, qz;
    rot_to_quat(T, qx, qy, qz, qw);
    char buf[256];
    snprintf(buf, sizeof buf, "%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f", frame_index, T[3], T[7], T[11], qx, qy, qz, qw);
    _trajfile << buf << endl;
}

void GraphicEndICP::saveFinalResult(const string &fileaddr)
{
    // closing loops around the accepted ones, then the graph hand-off (the reference optimises here and saves
    // final_after.g2o, :664-680; the optimiser is g2o's, out of scope, so the un-optimised graph is written)
    if (_loop_closure_detection) findMoreLoops();
    _graph.save("./data/final.g2o");
    // keyframe.txt: "id frame_index" (src/GraphicEnd.cpp:673-681)
    ofstream fout(fileaddr.c_str());
    for (size_t i = 0; i < _keyframes.size(); ++i) fout << _keyframes[i].id << " " << _keyframes[i].frame_index << endl;
    fout.close();
    _errorfile.flush();
    _trajfile.flush();
}

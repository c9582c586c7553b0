// GraphicEndICP.cpp -- see GraphicEndICP.h.  Control flow follows GraphicEnd::run (src/GraphicEnd.cpp:150-264);
// the pose arithmetic lives behind the C-ABI (HIP kernels), never here.
#include "GraphicEndICP.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

#include "png16.h"

using namespace std;

RESULT_OF_MULTIPNP::RESULT_OF_MULTIPNP() : norm(0.0), inliers(0) { mat4_identity(T); }

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
    slam3d_icp_default_params(&_params);
}

GraphicEndICP::~GraphicEndICP()
{
    if (_icp) slam3d_icp_destroy(_icp);
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
    _params.estimator = (_reader->Has("icp_estimator") && _reader->GetPara("icp_estimator") == "svd") ? SLAM3D_EST_SVD
                                                                                                       : SLAM3D_EST_POINT2PLANE;
    _params.normal_window = _reader->GetInt("icp_normal_window", 7);
    _params.min_inliers = _reader->GetInt("icp_min_inliers", 12);
    _params.error_threshold = _error_threshold;
    _max_batch = _reader->GetInt("loopclosure_frames", 30);
    _params.max_batch = _max_batch;
    _params.device = _reader->GetInt("hip_device", 0);
    const int rc = slam3d_icp_create(&_params, &_icp);
    if (rc != SLAM3D_OK) {
        cerr << "slam3d_icp_create failed: " << slam3d_strerror(rc) << endl;
        exit(1);                                                   // the reference exits on fatal config errors (:113)
    }
    _errorfile.open("./data/error_of_transform.log");              // src/GraphicEnd.cpp:153
    _trajfile.open("./data/trajectory_icp.txt");

    // first frame = keyframe 0 at the origin (src/GraphicEnd.cpp:106-145)
    readimage();
    _currKF = _present;
    _currKF.id = 0;
    _currKF.frame_index = _index;
    _keyframes.push_back(_currKF);
    _kf_poses.push_back(vector<double>(_kf_pos, _kf_pos + 16));
    _last = _present;
    writeTrajectoryLine(_index, _robot);
    _index++;
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
    cout << "load ok." << endl;
    return 0;
}

RESULT_OF_MULTIPNP GraphicEndICP::multiPnP(FRAME &frame1, FRAME &frame2, bool /*loopclosure*/, int /*frame_index*/,
                                           int minimum_inliers)
{
    vector<const FRAME *> a(1, &frame1), b(1, &frame2);
    return multiPnPBatch(a, b, minimum_inliers)[0];
}

vector<RESULT_OF_MULTIPNP> GraphicEndICP::multiPnPBatch(const vector<const FRAME *> &f1, const vector<const FRAME *> &f2,
                                                        int minimum_inliers)
{
    const int B = (int)f1.size();
    vector<RESULT_OF_MULTIPNP> out(B);
    for (int b0 = 0; b0 < B; b0 += _max_batch) {
        const int nb = min(_max_batch, B - b0);
        vector<const uint16_t *> s(nb), t(nb);
        for (int k = 0; k < nb; ++k) { s[k] = f1[b0 + k]->depth.data(); t[k] = f2[b0 + k]->depth.data(); }
        vector<slam3d_icp_result> res(nb);
        const int rc = slam3d_icp_align_depth_batch(_icp, nb, s.data(), t.data(), nullptr, res.data());
        if (rc < 0) {
            cerr << "slam3d_icp_align_depth_batch: " << slam3d_strerror(rc) << " " << slam3d_last_error(_icp) << endl;
            continue;                                              // results stay Identity = "not matched"
        }
        for (int k = 0; k < nb; ++k) {
            RESULT_OF_MULTIPNP &r = out[b0 + k];
            r.inliers = res[k].inliers;
            r.norm = res[k].norm;
            // thresholds of multiPnP: inliers (src/GraphicEnd.cpp:599), norm (:621); library used params.min_inliers
            const bool ok = res[k].status == SLAM3D_OK && res[k].inliers >= minimum_inliers;
            if (ok) memcpy(r.T, res[k].T, sizeof r.T);
            cout << "multiICP::inliers = " << r.inliers << ", norm = " << r.norm << ", status = " << res[k].status << endl;
        }
    }
    return out;
}

void GraphicEndICP::generateKeyFrame(const double *T)
{
    // T maps the current keyframe's pose to the present one (src/GraphicEnd.cpp:304-351); the g2o vertex/edge the
    // reference adds there becomes a stored pose here (pose-graph back end is out of scope)
    _currKF = _present;
    _currKF.id = (int)_keyframes.size();
    _currKF.frame_index = _index;
    double P[16];
    mat4_mul(T, _kf_pos, P);
    memcpy(_kf_pos, P, sizeof P);
    _keyframes.push_back(_currKF);
    _kf_poses.push_back(vector<double>(_kf_pos, _kf_pos + 16));
}

int GraphicEndICP::run()
{
    cout << "********************" << endl;
    readimage();
    // present -> current keyframe (src/GraphicEnd.cpp:168-170: the result is inverted by the caller)
    RESULT_OF_MULTIPNP result = multiPnP(_currKF, _present);
    double T[16];
    mat4_inverse_rigid(result.T, T);
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
            generateKeyFrame(Tl);
            _present = keep;
            double Tp[16];
            mat4_inverse_rigid(r.T, Tp);
            generateKeyFrame(Tp);
            memcpy(_robot, _kf_pos, sizeof _robot);
            _last = _present;
        }
    } else if (result.norm > _max_pos_change) {
        _errorfile << result.norm << endl;                          // :232
        mat4_mul(T, _kf_pos, _robot);
        generateKeyFrame(T);
        _lost = 0;
        _last = _present;
    } else {
        _errorfile << result.norm << endl;                          // :243
        mat4_mul(T, _kf_pos, _robot);
        _lost = 0;
        _last = _present;
    }
    if (_lost > _lost_frames) cerr << "the robot lost. (lost recovery is out of scope of the ICP path)" << endl;
    writeTrajectoryLine(_index, _robot);
    _index++;
    _present.frame_index = _index;
    return 1;
}

void GraphicEndICP::writeTrajectoryLine(int frame_index, const double *T)
{
    // TUM trajectory line "timestamp tx ty tz qx qy qz qw" (src/generateTrajectory.cpp:62-72)
    const double tr = T[0] + T[5] + T[10];
    double qw, qx, qy, qz;
    if (tr > 0) {
        const double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (T[9] - T[6]) / s; qy = (T[2] - T[8]) / s; qz = (T[4] - T[1]) / s;
    } else if (T[0] > T[5] && T[0] > T[10]) {
        const double s = sqrt(1.0 + T[0] - T[5] - T[10]) * 2; qw = (T[9] - T[6]) / s; qx = 0.25 * s; qy = (T[1] + T[4]) / s; qz = (T[2] + T[8]) / s;
    } else if (T[5] > T[10]) {
        const double s = sqrt(1.0 + T[5] - T[0] - T[10]) * 2; qw = (T[2] - T[8]) / s; qx = (T[1] + T[4]) / s; qy = 0.25 * s; qz = (T[6] + T[9]) / s;
    } else {
        const double s = sqrt(1.0 + T[10] - T[0] - T[5]) * 2; qw = (T[4] - T[1]) / s; qx = (T[2] + T[8]) / s; qy = (T[6] + T[9]) / s; qz = 0.25 * s;
    }
    char buf[256];
    snprintf(buf, sizeof buf, "%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f", frame_index, T[3], T[7], T[11], qx, qy, qz, qw);
    _trajfile << buf << endl;
}

void GraphicEndICP::saveFinalResult(const string &fileaddr)
{
    // keyframe.txt: "id frame_index" (src/GraphicEnd.cpp:673-681)
    ofstream fout(fileaddr.c_str());
    for (size_t i = 0; i < _keyframes.size(); ++i) fout << _keyframes[i].id << " " << _keyframes[i].frame_index << endl;
    fout.close();
    _errorfile.flush();
    _trajfile.flush();
}

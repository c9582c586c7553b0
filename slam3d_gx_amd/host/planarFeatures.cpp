// planarFeatures <dep.png> [keypoints.txt] -- BASELINE config 1's binary (src/planarFeatures.cpp:26-136) without the
// OpenCV parts this image cannot build: no window, no FAST detector.  Which pixels sit on a locally planar 7x7 depth
// patch -- the reference's isPlanar (:88-136: 49 back-projected points, plane within 0.01 m, more than 40 inliers) --
// is answered for EVERY pixel at once by the library's k_normals (spec S2 of DESIGN.md: least-squares plane of the valid
// window points instead of a RANSAC plane, >= 41 of 49 within 0.01 m), through the C-ABI.
//   candidates  = the key points of keypoints.txt ("u v" per line, e.g. exported from cv::FAST), else every pixel whose
//                 7x7 patch lies inside the image (the reference takes dep(Range(v-3, v+4), Range(u-3, u+4)), :92);
//   valid       = depth != 0 at the key point (:58-62);
//   planar      = spec S2's flag;  planar_nozero = planar AND no zero depth in the patch (the reference returns false
//                 at the first zero, :103-107).
// Prints the reference's summary line (:82) and the two planar counts.  Intrinsics are the reference's constants
// (:13-14: fx = fy = 525, cx = 320, cy = 235.5, factor 1000) unless given as five more arguments.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/slam3d_icp.h"
#include "png16.h"

using namespace std;

static void usage() { cout << "planarFeatures dep [keypoints.txt] [fx fy cx cy factor]" << endl; }

int main(int argc, char **argv)
{
    if (argc < 2) { usage(); return -1; }
    int W = 0, H = 0;
    vector<uint16_t> dep;
    string err;
    if (!read_png_gray16(argv[1], W, H, dep, err)) { cerr << "cannot read " << argv[1] << ": " << err << endl; return -1; }
    string kp_path;
    int a = 2;
    if (argc > a && (argc - a) % 5 != 0) kp_path = argv[a++];
    double fx = 525.0, fy = 525.0, cx = 320.0, cy = 235.5, factor = 1000.0;       // src/planarFeatures.cpp:13-14
    if (argc - a >= 5) { fx = atof(argv[a]); fy = atof(argv[a + 1]); cx = atof(argv[a + 2]); cy = atof(argv[a + 3]); factor = atof(argv[a + 4]); }

    slam3d_icp_params p;
    slam3d_icp_default_params(&p);
    p.width = W; p.height = H; p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.depth_factor = factor;
    p.z_filter = 10.0;                // the reference's isPlanar has no range limit; 10 m is beyond a Kinect's and within what
                                      // the library's fixed-point sums admit at 640x480 (slam3d_icp_create checks)
    p.iterations = 1; p.max_batch = 1;
    slam3d_icp_handle *h = nullptr;
    int rc = slam3d_icp_create(&p, &h);
    if (rc) { cerr << "slam3d_icp_create: " << slam3d_strerror(rc) << endl; return -1; }
    // the frame as both roles of one pair: the run builds its normals (target role), the single iteration is irrelevant
    slam3d_icp_result res;
    rc = slam3d_icp_frame_set_depth_host(h, 0, dep.data());
    if (!rc) rc = slam3d_icp_set_pair(h, 0, 0, 0);
    if (!rc) rc = slam3d_icp_run(h, 1, nullptr, nullptr);
    if (!rc) rc = slam3d_icp_fetch_results(h, 1, &res);
    vector<float> nrm((size_t)W * H * 4);
    if (!rc) rc = slam3d_icp_get_clouds(h, 0, nullptr, nullptr, nrm.data());
    if (rc) { cerr << "library call failed: " << slam3d_strerror(rc) << " " << slam3d_last_error(h) << endl; slam3d_icp_destroy(h); return -1; }
    slam3d_icp_destroy(h);

    vector<pair<int, int>> kp;
    if (!kp_path.empty()) {
        ifstream fin(kp_path);
        double u, v;
        while (fin >> u >> v) kp.emplace_back((int)u, (int)v);          // the reference truncates (:90-91)
    } else {
        for (int v = 3; v + 3 < H; ++v)
            for (int u = 3; u + 3 < W; ++u) kp.emplace_back(u, v);
    }
    size_t valid = 0, planar = 0, planar_nozero = 0;
    for (const auto &k : kp) {
        const int u = k.first, v = k.second;
        if (u < 3 || v < 3 || u + 3 >= W || v + 3 >= H) continue;      // (the reference asserts inside OpenCV there)
        if (dep[(size_t)v * W + u] == 0) continue;
        ++valid;
        if (nrm[((size_t)v * W + u) * 4 + 3] <= 0.5f) continue;
        ++planar;
        bool zero = false;
        for (int j = -3; j <= 3 && !zero; ++j)
            for (int i = -3; i <= 3; ++i)
                if (dep[(size_t)(v + j) * W + (u + i)] == 0) { zero = true; break; }
        if (!zero) ++planar_nozero;
    }
    cout << "total kp: " << kp.size() << ", valid: " << valid << ", planar: " << planar << endl;
    cout << "planar with no zero depth in the 7x7 patch (the reference's rule): " << planar_nozero << endl;
    return 0;
}

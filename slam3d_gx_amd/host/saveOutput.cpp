// saveOutput keyframe.txt final.g2o [pass_z] -- the reference's map builder (src/saveOutput.cpp:11-105): for every
// keyframe "id frame" of keyframe.txt load <data_source>/pcd/<frame>.pcd, VoxelGrid(grid_leaf) (:80-83), PassThrough
// z in [0, pass_z = 5.0] (:84-87), transform by the pose of vertex id in the g2o file (:89), append (:90); VoxelGrid
// the sum once more (:97-100) and write result.pcd (:101).  The three cloud operations run on the GPU behind the
// C-ABI (slam3d_voxel_grid_only, slam3d_pass_transform); the g2o file is read as text (no g2o).
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "../../include/slam3d_icp.h"
#include "ParameterReader.h"
#include "PoseGraph.h"
#include "pcd_io.h"

using namespace std;

static void usage() { cout << "saveOutput keyframe.txt final.g2o [ pass_z ]" << endl; }

int main(int argc, char **argv)
{
    if (argc < 3) { usage(); return -1; }
    ParameterReader reader("./parameters.yaml");
    g_pParaReader = &reader;
    PoseGraph graph;
    if (!graph.load(argv[2])) { cerr << "cannot read " << argv[2] << endl; return 1; }
    ifstream fin(argv[1]);
    if (!fin) { cerr << "cannot read " << argv[1] << endl; return 1; }
    const float grid_leaf = (float)reader.GetDouble("grid_leaf", 0.03);
    const string pclPath = reader.GetPara("data_source") + "/pcd/";
    const float z = argc == 4 ? (float)atof(argv[3]) : 5.0f;

    // first pass: the keyframe clouds (their sizes fix the capacity of the device-side work buffers)
    struct KF { int id, frame; vector<PointXYZRGBA16> cloud; };
    vector<KF> kfs;
    size_t biggest = 0, total = 0;
    int id, frame;
    while (fin >> id >> frame) {
        KF k; k.id = id; k.frame = frame;
        const string path = pclPath + to_string(frame) + ".pcd";
        cout << "loading " << path << endl;
        int w = 0, h = 0; string err;
        if (!read_pcd(path, k.cloud, w, h, err)) { cerr << err << endl; continue; }
        if (!graph.vertex(id)) { cout << "cannot find vertex: " << id << endl; continue; }      // :66-70
        biggest = max(biggest, k.cloud.size()); total += k.cloud.size();
        kfs.push_back(std::move(k));
    }
    if (kfs.empty()) { cerr << "no keyframe cloud could be loaded" << endl; return 1; }

    slam3d_icp_params p;
    slam3d_icp_default_params(&p);
    const size_t cap = max(biggest, total);                        // the merged cloud is at most the sum of the inputs
    p.width = 2048; p.height = (int)((cap + 2047) / 2048); if (p.height < 8) p.height = 8;
    p.iterations = 1; p.max_batch = 1;
    p.fx = p.fy = 1e9; p.cx = p.cy = 0.0; p.z_filter = 1.0;        // no camera behind this handle: it only hosts the cloud kernels
    p.device = reader.GetInt("hip_device", 0);
    slam3d_icp_handle *hdl = nullptr;
    int rc = slam3d_icp_create(&p, &hdl);
    if (rc != SLAM3D_OK) { cerr << "slam3d_icp_create failed: " << slam3d_strerror(rc) << endl; return 1; }

    vector<PointXYZRGBA16> merged, tmp, moved;
    for (size_t i = 0; i < kfs.size(); ++i) {
        const KF &k = kfs[i];
        tmp.resize(k.cloud.size());
        int m = 0, kept = 0;
        rc = slam3d_voxel_grid_only(hdl, k.cloud.data(), (int)k.cloud.size(), grid_leaf, tmp.data(), &m);
        if (rc) { cerr << "voxel grid: " << slam3d_strerror(rc) << endl; return 1; }
        moved.resize((size_t)m);
        rc = slam3d_pass_transform(hdl, tmp.data(), m, z, graph.vertex(k.id)->T, moved.data(), &kept);
        if (rc) { cerr << "transform: " << slam3d_strerror(rc) << endl; return 1; }
        cout << "keyframe " << k.id << " frame " << k.frame << ": " << k.cloud.size() << " -> " << m << " -> " << kept << " points" << endl;
        merged.insert(merged.end(), moved.begin(), moved.end());   // dropped records are NaN; the final VoxelGrid ignores them
    }
    vector<PointXYZRGBA16> out(merged.size());
    int n_out = 0;
    rc = slam3d_voxel_grid_only(hdl, merged.data(), (int)merged.size(), grid_leaf, out.data(), &n_out);
    if (rc) { cerr << "final voxel grid: " << slam3d_strerror(rc) << endl; return 1; }
    slam3d_icp_destroy(hdl);
    string err;
    if (!write_pcd_ascii("result.pcd", out.data(), (size_t)n_out, n_out, 1, err)) { cerr << err << endl; return 1; }
    cout << "final result saved: " << n_out << " points." << endl;                              // :102
    g_pParaReader = nullptr;
    return 0;
}

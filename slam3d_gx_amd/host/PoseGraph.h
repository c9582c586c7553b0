// PoseGraph.h -- the pose hand-off the reference keeps inside g2o's SparseOptimizer (SLAMEnd::globalOptimizer):
// VertexSE3 per keyframe (src/GraphicEnd.cpp:319-326) and EdgeSE3 per accepted alignment (:328-338, :708-719,
// :744-757, :826-838, :900-914), information = 100 on the diagonal.  No optimiser here (the g2o back end is out
// of scope): save() writes the graph in g2o's text format so stock g2o tools can load and
// optimise it, which is what the reference's saveFinalResult does after optimize() (:661-682).
#pragma once
#include <string>
#include <vector>

struct PoseVertex { int id; double T[16]; bool fixed; };
struct PoseEdge { int from, to; double T[16]; double info_diag[6]; bool robust; };

void rot_to_quat(const double *T, double &qx, double &qy, double &qz, double &qw);
void quat_to_mat4(const double *t3, double qx, double qy, double qz, double qw, double *T);

class PoseGraph {
 public:
    void addVertex(int id, const double *T, bool fixed = false);
    // measurement T: pose of vertex `to` expressed in the frame of vertex `from` (EdgeSE3 convention)
    void addEdge(int from, int to, const double *T, double info = 100.0, bool robust = false);
    bool save(const std::string &path) const;              // VERTEX_SE3:QUAT / EDGE_SE3:QUAT / FIX
    bool load(const std::string &path);                    // the same text format (what saveOutput reads, src/saveOutput.cpp:29)
    const PoseVertex *vertex(int id) const;
    const std::vector<PoseVertex> &vertices() const { return _v; }
    const std::vector<PoseEdge> &edges() const { return _e; }

 private:
    std::vector<PoseVertex> _v;
    std::vector<PoseEdge> _e;
};

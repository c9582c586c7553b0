#include "PoseGraph.h"

#include <cmath>
#include <cstdio>
#include <cstring>

void rot_to_quat(const double *T, double &qx, double &qy, double &qz, double &qw)
{
    const double tr = T[0] + T[5] + T[10];
    if (tr > 0) {
        const double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (T[9] - T[6]) / s; qy = (T[2] - T[8]) / s; qz = (T[4] - T[1]) / s;
    } else if (T[0] > T[5] && T[0] > T[10]) {
        const double s = sqrt(1.0 + T[0] - T[5] - T[10]) * 2; qw = (T[9] - T[6]) / s; qx = 0.25 * s; qy = (T[1] + T[4]) / s; qz = (T[2] + T[8]) / s;
    } else if (T[5] > T[10]) {
        const double s = sqrt(1.0 + T[5] - T[0] - T[10]) * 2; qw = (T[2] - T[8]) / s; qx = (T[1] + T[4]) / s; qy = 0.25 * s; qz = (T[6] + T[9]) / s;
    } else {
        const double s = sqrt(1.0 + T[10] - T[0] - T[5]) * 2; qw = (T[4] - T[1]) / s; qx = (T[2] + T[8]) / s; qy = (T[6] + T[9]) / s; qz = 0.25 * s;
    }
}

void PoseGraph::addVertex(int id, const double *T, bool fixed)
{
    PoseVertex v;
    v.id = id; v.fixed = fixed;
    memcpy(v.T, T, sizeof v.T);
    _v.push_back(v);
}

void PoseGraph::addEdge(int from, int to, const double *T, double info, bool robust)
{
    PoseEdge e;
    e.from = from; e.to = to; e.robust = robust;
    memcpy(e.T, T, sizeof e.T);
    for (int k = 0; k < 6; ++k) e.info_diag[k] = info;
    _e.push_back(e);
}

bool PoseGraph::save(const std::string &path) const
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    double qx, qy, qz, qw;
    for (size_t i = 0; i < _v.size(); ++i) {
        const PoseVertex &v = _v[i];
        rot_to_quat(v.T, qx, qy, qz, qw);
        fprintf(f, "VERTEX_SE3:QUAT %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", v.id, v.T[3], v.T[7], v.T[11], qx, qy, qz, qw);
    }
    for (size_t i = 0; i < _v.size(); ++i)
        if (_v[i].fixed) fprintf(f, "FIX %d\n", _v[i].id);
    for (size_t i = 0; i < _e.size(); ++i) {
        const PoseEdge &e = _e[i];
        rot_to_quat(e.T, qx, qy, qz, qw);
        fprintf(f, "EDGE_SE3:QUAT %d %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g", e.from, e.to, e.T[3], e.T[7], e.T[11], qx, qy, qz, qw);
        for (int r = 0; r < 6; ++r)                      // upper triangle of the 6x6 information matrix, row-major
            for (int c = r; c < 6; ++c) fprintf(f, " %.9g", r == c ? e.info_diag[r] : 0.0);
        fprintf(f, "\n");
    }
    fclose(f);
    return true;
}

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

void quat_to_mat4(const double *t, double x, double y, double z, double w, double *T)
{
    const double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    T[0] = 1 - 2 * (y * y + z * z); T[1] = 2 * (x * y - z * w);     T[2] = 2 * (x * z + y * w);     T[3] = t[0];
    T[4] = 2 * (x * y + z * w);     T[5] = 1 - 2 * (x * x + z * z); T[6] = 2 * (y * z - x * w);     T[7] = t[1];
    T[8] = 2 * (x * z - y * w);     T[9] = 2 * (y * z + x * w);     T[10] = 1 - 2 * (x * x + y * y); T[11] = t[2];
    T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
}

const PoseVertex *PoseGraph::vertex(int id) const
{
    for (size_t i = 0; i < _v.size(); ++i) if (_v[i].id == id) return &_v[i];
    return nullptr;
}

bool PoseGraph::load(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    _v.clear(); _e.clear();
    char line[2048], tag[64];
    while (fgets(line, sizeof line, f)) {
        if (sscanf(line, "%63s", tag) != 1) continue;
        if (!strcmp(tag, "VERTEX_SE3:QUAT")) {
            int id; double t[3], q[4];
            if (sscanf(line, "%*s %d %lf %lf %lf %lf %lf %lf %lf", &id, &t[0], &t[1], &t[2], &q[0], &q[1], &q[2], &q[3]) != 8) continue;
            PoseVertex v; v.id = id; v.fixed = false;
            quat_to_mat4(t, q[0], q[1], q[2], q[3], v.T);
            _v.push_back(v);
        } else if (!strcmp(tag, "EDGE_SE3:QUAT")) {
            int a, b; double t[3], q[4], info[21];
            int off = 0;
            if (sscanf(line, "%*s %d %d %lf %lf %lf %lf %lf %lf %lf%n", &a, &b, &t[0], &t[1], &t[2], &q[0], &q[1], &q[2], &q[3], &off) != 9) continue;
            PoseEdge e; e.from = a; e.to = b; e.robust = false;
            quat_to_mat4(t, q[0], q[1], q[2], q[3], e.T);
            const char *p = line + off;
            for (int k = 0; k < 21; ++k) { int used = 0; info[k] = 0.0; if (sscanf(p, "%lf%n", &info[k], &used) == 1) p += used; }
            const int diag[6] = { 0, 6, 11, 15, 18, 20 };
            for (int k = 0; k < 6; ++k) e.info_diag[k] = info[diag[k]];
            _e.push_back(e);
        } else if (!strcmp(tag, "FIX")) {
            int id;
            if (sscanf(line, "%*s %d", &id) == 1)
                for (size_t i = 0; i < _v.size(); ++i) if (_v[i].id == id) _v[i].fixed = true;
        }
    }
    fclose(f);
    return true;
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

#include "pcd_io.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace std;

namespace {
struct Field { string name; int size; char type; int count; int offset; };

vector<string> split(const string &s)
{
    vector<string> out;
    istringstream is(s);
    string t;
    while (is >> t) out.push_back(t);
    return out;
}
}  // namespace

bool read_pcd(const string &path, vector<PointXYZRGBA16> &pts, int &width, int &height, string &err)
{
    pts.clear();
    width = height = 0;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    vector<Field> fields;
    long long points = -1;
    string data;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        vector<string> t = split(line);
        if (t.empty() || t[0][0] == '#') continue;
        if (t[0] == "FIELDS") { fields.resize(t.size() - 1); for (size_t i = 1; i < t.size(); ++i) { fields[i - 1].name = t[i]; fields[i - 1].size = 4; fields[i - 1].type = 'F'; fields[i - 1].count = 1; } }
        else if (t[0] == "SIZE") { for (size_t i = 1; i < t.size() && i - 1 < fields.size(); ++i) fields[i - 1].size = atoi(t[i].c_str()); }
        else if (t[0] == "TYPE") { for (size_t i = 1; i < t.size() && i - 1 < fields.size(); ++i) fields[i - 1].type = t[i][0]; }
        else if (t[0] == "COUNT") { for (size_t i = 1; i < t.size() && i - 1 < fields.size(); ++i) fields[i - 1].count = atoi(t[i].c_str()); }
        else if (t[0] == "WIDTH" && t.size() > 1) width = atoi(t[1].c_str());
        else if (t[0] == "HEIGHT" && t.size() > 1) height = atoi(t[1].c_str());
        else if (t[0] == "POINTS" && t.size() > 1) points = atoll(t[1].c_str());
        else if (t[0] == "DATA" && t.size() > 1) { data = t[1]; break; }
    }
    if (data.empty() || fields.empty()) { fclose(f); err = "not a PCD file (no FIELDS / DATA): " + path; return false; }
    if (points < 0) points = (long long)width * height;
    int rec = 0, ix = -1, iy = -1, iz = -1, ic = -1;
    for (size_t i = 0; i < fields.size(); ++i) {
        fields[i].offset = rec;
        rec += fields[i].size * fields[i].count;
        if (fields[i].name == "x") ix = (int)i;
        else if (fields[i].name == "y") iy = (int)i;
        else if (fields[i].name == "z") iz = (int)i;
        else if (fields[i].name == "rgba" || fields[i].name == "rgb") ic = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0 || fields[ix].size != 4 || fields[iy].size != 4 || fields[iz].size != 4 || fields[ix].type != 'F') {
        fclose(f); err = "PCD needs 4-byte float fields x y z: " + path; return false;
    }
    pts.resize((size_t)points);
    if (data == "binary") {
        vector<unsigned char> buf((size_t)points * rec);
        const size_t got = fread(buf.data(), 1, buf.size(), f);
        fclose(f);
        if (got != buf.size()) { err = "truncated PCD body: " + path; pts.clear(); return false; }
        for (long long i = 0; i < points; ++i) {
            const unsigned char *p = buf.data() + (size_t)i * rec;
            PointXYZRGBA16 &q = pts[(size_t)i];
            memcpy(&q.x, p + fields[ix].offset, 4); memcpy(&q.y, p + fields[iy].offset, 4); memcpy(&q.z, p + fields[iz].offset, 4);
            q.rgba = 0;
            if (ic >= 0 && fields[ic].size == 4) memcpy(&q.rgba, p + fields[ic].offset, 4);
        }
        return true;
    }
    if (data == "ascii") {
        for (long long i = 0; i < points; ++i) {
            if (!fgets(line, sizeof line, f)) { fclose(f); err = "truncated PCD body: " + path; pts.clear(); return false; }
            vector<string> t = split(line);
            PointXYZRGBA16 &q = pts[(size_t)i];
            q.x = q.y = q.z = 0.0f; q.rgba = 0;
            size_t col = 0;
            for (size_t k = 0; k < fields.size(); ++k) {
                if (col < t.size()) {
                    if ((int)k == ix) q.x = strtof(t[col].c_str(), nullptr);
                    else if ((int)k == iy) q.y = strtof(t[col].c_str(), nullptr);
                    else if ((int)k == iz) q.z = strtof(t[col].c_str(), nullptr);
                    else if ((int)k == ic) {
                        if (fields[k].type == 'F') { const float v = strtof(t[col].c_str(), nullptr); memcpy(&q.rgba, &v, 4); }   // PCL's packed-float rgb
                        else q.rgba = (uint32_t)strtoul(t[col].c_str(), nullptr, 10);
                    }
                }
                col += fields[k].count;
            }
        }
        fclose(f);
        return true;
    }
    fclose(f);
    err = "unsupported PCD DATA mode '" + data + "' (binary_compressed is not read): " + path;
    pts.clear();
    return false;
}

bool write_pcd_ascii(const string &path, const PointXYZRGBA16 *pts, size_t n, int width, int height, string &err)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { err = "cannot create " + path; return false; }
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
               "WIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA ascii\n", width, height, n);
    for (size_t i = 0; i < n; ++i) fprintf(f, "%.9g %.9g %.9g %u\n", pts[i].x, pts[i].y, pts[i].z, pts[i].rgba);   // %.9g round-trips a float
    fclose(f);
    return true;
}

bool write_pcd_binary(const string &path, const PointXYZRGBA16 *pts, size_t n, int width, int height, string &err)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
               "WIDTH %d\nHEIGHT %d\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", width, height, n);
    const bool ok = fwrite(pts, sizeof(PointXYZRGBA16), n, f) == n;
    fclose(f);
    if (!ok) err = "short write: " + path;
    return ok;
}

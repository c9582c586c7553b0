// host_selftest <png> <parameters.yaml> -- prints what the host-side readers see (used by tests/test_host_frontend.py)
#include <cstdio>
#include <iostream>

#include "ParameterReader.h"
#include "png16.h"
#include "pcd_io.h"
#include <cstring>
#include <string>
#include <vector>

// host_selftest pcd <in.pcd> <out.pcd>: reads, prints "pcd n width height checksum", writes the records back
static int pcd_leg(const char *in, const char *out)
{
    std::vector<PointXYZRGBA16> pts;
    int w = 0, h = 0;
    std::string err;
    if (!read_pcd(in, pts, w, h, err)) { std::cerr << err << std::endl; return 1; }
    unsigned long long sum = 0;
    for (size_t i = 0; i < pts.size(); ++i) {
        uint32_t b[4];
        memcpy(b, &pts[i], 16);
        sum += (unsigned long long)(b[0] ^ (b[1] * 3u) ^ (b[2] * 5u) ^ (b[3] * 7u)) * (i % 9973 + 1);
    }
    printf("pcd %zu %d %d %llu\n", pts.size(), w, h, sum);
    if (!write_pcd_binary(out, pts.data(), pts.size(), w, h, err)) { std::cerr << err << std::endl; return 1; }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && std::string(argv[1]) == "pcd") return pcd_leg(argv[2], argv[3]);
    if (argc < 3) return 2;
    int w = 0, h = 0;
    std::vector<uint16_t> px;
    std::string err;
    if (!read_png_gray16(argv[1], w, h, px, err)) { std::cerr << err << std::endl; return 1; }
    unsigned long long sum = 0, nz = 0;
    for (size_t i = 0; i < px.size(); ++i) { sum += (unsigned long long)px[i] * (i % 9973 + 1); nz += px[i] != 0; }
    printf("png %d %d %llu %llu\n", w, h, sum, nz);
    ParameterReader r(argv[2]);
    printf("para data_source=%s z_filter=%s max_planes=%s missing=%s fx=%.3f factor=%.1f iters=%d\n", r.GetPara("data_source").c_str(),
           r.GetPara("z_filter").c_str(), r.GetPara("max_planes").c_str(), r.GetPara("no_such_key").c_str(), camera_fx,
           camera_factor, r.GetInt("icp_iterations", 20));
    return 0;
}

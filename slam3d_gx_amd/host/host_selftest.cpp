// host_selftest <png> <parameters.yaml> -- prints what the host-side readers see (used by tests/test_host_frontend.py)
#include <cstdio>
#include <iostream>

#include "ParameterReader.h"
#include "png16.h"

int main(int argc, char **argv)
{
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

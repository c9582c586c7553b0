// pcd_io.h -- PCD v0.7 reader/writer for the reference's on-disk clouds.
// The reference loads <data_source>/pcd/<i>.pcd with pcl::io::loadPCDFile (src/GraphicEnd.cpp:279-280); the files are
// written by convert2PCD with pcl::io::savePCDFileBinary (src/convert2PCD.cpp:75-79): FIELDS x y z rgba, SIZE 4 4 4 4,
// TYPE F F F U, DATA binary, 16-byte records (data/exp1/pcd/1.pcd header).  Self-contained: no PCL.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct PointXYZRGBA16 { float x, y, z; uint32_t rgba; };   // the binary record == the device record of slam3d_voxel_grid

// binary or ascii; needs fields x, y, z (4-byte floats); rgb/rgba optional (0 when absent); other fields are skipped
bool read_pcd(const std::string &path, std::vector<PointXYZRGBA16> &pts, int &width, int &height, std::string &err);
// header as PCL writes it for PointXYZRGBA (same lines as the reference's fixtures), DATA binary
bool write_pcd_binary(const std::string &path, const PointXYZRGBA16 *pts, size_t n, int width, int height, std::string &err);
// DATA ascii, what pcl::io::savePCDFile(name, cloud) writes by default (src/saveOutput.cpp:101)
bool write_pcd_ascii(const std::string &path, const PointXYZRGBA16 *pts, size_t n, int width, int height, std::string &err);

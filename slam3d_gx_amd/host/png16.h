// png16.h -- minimal reader for the dataset's depth images: 16-bit (or 8-bit) grayscale, non-interlaced PNG
// (what cv::imread(path, -1) yields at src/GraphicEnd.cpp:276).  zlib only; no libpng headers in this image.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// returns false (and a message in err) on anything it does not understand
bool read_png_gray16(const std::string &path, int &width, int &height, std::vector<uint16_t> &pixels, std::string &err);

#include "png16.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static unsigned char paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (unsigned char)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

bool read_png_gray16(const std::string &path, int &width, int &height, std::vector<uint16_t> &pixels, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) { err = "not a PNG: " + path; return false; }
    size_t pos = 8;
    int bit_depth = 0, color_type = -1, interlace = 0;
    std::vector<unsigned char> idat;
    width = height = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char *type = (const char *)&file[pos + 4];
        if (pos + 12 + len > file.size()) { err = "truncated PNG"; return false; }
        const unsigned char *data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            width = (int)be32(data); height = (int)be32(data + 4);
            bit_depth = data[8]; color_type = data[9]; interlace = data[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (color_type != 0 || (bit_depth != 16 && bit_depth != 8) || interlace != 0 || width <= 0 || height <= 0) {
        err = "unsupported PNG (need non-interlaced 8/16-bit grayscale): " + path;
        return false;
    }
    const int bpp = bit_depth / 8;
    const size_t stride = (size_t)width * bpp;
    std::vector<unsigned char> raw((stride + 1) * (size_t)height);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) {
        err = "zlib inflate failed: " + path;
        return false;
    }
    pixels.assign((size_t)width * height, 0);
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (int y = 0; y < height; ++y) {
        const unsigned char *row = &raw[(stride + 1) * (size_t)y];
        const int ft = row[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = row[1 + i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: err = "bad PNG filter"; return false;
            }
            cur[i] = (unsigned char)v;
        }
        for (int x = 0; x < width; ++x)
            pixels[(size_t)y * width + x] = bpp == 2 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : cur[x];
        prev.swap(cur);
    }
    return true;
}

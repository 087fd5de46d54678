// png_reader.hpp — PNG decoding with zlib only (no libpng / OpenCV in this environment): the two cv::imread calls of
// img_loader/ImageLoader.h:130-188.  Non-interlaced, bit depth 8 or 16, colour types 0 (gray), 2 (RGB), 4 (gray+alpha),
// 6 (RGBA); filter types 0-4.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace psgsdf_host {

struct PngImage { int width = 0, height = 0, channels = 0, bit_depth = 0; std::vector<uint16_t> px; };   // samples widened to 16 bit, row-major, interleaved

inline bool read_png(const std::string& path, PngImage& img) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) return false;
    size_t p = 8; int ctype = -1, interlace = 0;
    std::vector<unsigned char> idat;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3]; };
    while (p + 8 <= file.size()) {
        uint32_t len = be32(p); const char* type = (const char*)&file[p + 4]; size_t data = p + 8;
        if (data + len + 4 > file.size()) return false;
        if (!strncmp(type, "IHDR", 4)) { img.width = be32(data); img.height = be32(data + 4); img.bit_depth = file[data + 8]; ctype = file[data + 9]; interlace = file[data + 12]; }
        else if (!strncmp(type, "IDAT", 4)) idat.insert(idat.end(), file.begin() + data, file.begin() + data + len);
        else if (!strncmp(type, "IEND", 4)) break;
        p = data + len + 4;
    }
    if (interlace || (img.bit_depth != 8 && img.bit_depth != 16)) return false;
    img.channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!img.channels) return false;
    const size_t bpp = (size_t)img.channels * img.bit_depth / 8, stride = bpp * img.width;
    std::vector<unsigned char> raw((stride + 1) * img.height);
    uLongf out_len = raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != raw.size()) return false;
    std::vector<unsigned char> cur(stride), prev(stride, 0);
    img.px.resize((size_t)img.width * img.height * img.channels);
    for (int y = 0; y < img.height; ++y) {
        const unsigned char* row = &raw[(stride + 1) * y]; int ft = row[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = row[1 + i], v;
            switch (ft) {
                case 0: v = x; break; case 1: v = x + a; break; case 2: v = x + b; break; case 3: v = x + ((a + b) >> 1); break;
                case 4: { int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c); int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); v = x + pr; break; }
                default: return false;
            }
            cur[i] = (unsigned char)v;
        }
        uint16_t* dst = &img.px[(size_t)y * img.width * img.channels];
        for (int i = 0; i < img.width * img.channels; ++i) dst[i] = img.bit_depth == 8 ? cur[i] : (uint16_t)((cur[2 * i] << 8) | cur[2 * i + 1]);
        prev.swap(cur);
    }
    return true;
}

}  // namespace psgsdf_host

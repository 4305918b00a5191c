// dataset.cpp -- see myslam/dataset.h.
#include "myslam/dataset.h"

#include <zlib.h>

#include <cstdlib>
#include <iomanip>
#include <sstream>

namespace myslam {

std::vector<AssociateEntry> ReadAssociateFile(const std::string& path) {
    std::vector<AssociateEntry> out;
    std::ifstream fin(path);
    std::string line;
    while (std::getline(fin, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        AssociateEntry e;
        std::string depthT;
        if (!(ss >> e.rgbTimeText >> e.rgbFile >> depthT >> e.depthFile)) break;
        e.rgbTime = std::atof(e.rgbTimeText.c_str());
        e.depthTime = std::atof(depthT.c_str());
        out.push_back(e);
    }
    return out;
}

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static inline int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

bool DecodePng(const std::string& path, DecodedImage& out) {
    out = DecodedImage();
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const uint8_t* type = &buf[pos + 4];
        if (pos + 12 + (size_t)len > buf.size()) return false;
        const uint8_t* data = &buf[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || interlace != 0 || ch == 0 || !((depth == 8) || (depth == 16 && ctype == 0))) return false;
    const size_t bpp = (size_t)ch * depth / 8, stride = bpp * w;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
    out.data.assign(stride * h, 0);
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = &raw[(stride + 1) * y];
        uint8_t* cur = &out.data[stride * y];
        const uint8_t* up = y ? &out.data[stride * (y - 1)] : nullptr;
        const int ft = src[0];
        ++src;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (ft) {
                case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
                default: out = DecodedImage(); return false;
            }
            cur[x] = (uint8_t)v;
        }
    }
    if (depth == 16) {                               // big-endian samples -> host order
        uint16_t* p16 = (uint16_t*)out.data.data();
        for (size_t i = 0; i < (size_t)w * h; ++i) { const uint8_t* q = &out.data[2 * i]; p16[i] = (uint16_t)((q[0] << 8) | q[1]); }
    }
    out.width = w; out.height = h; out.channels = ch; out.bitDepth = depth;
    return true;
}

bool ReadColorBGR(const std::string& path, DecodedImage& out) {
    DecodedImage img;
    if (!DecodePng(path, img) || img.bitDepth != 8) return false;
    out = DecodedImage();
    out.width = img.width; out.height = img.height; out.channels = 3; out.bitDepth = 8;
    out.data.resize((size_t)3 * img.width * img.height);
    const size_t n = (size_t)img.width * img.height;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = &img.data[i * img.channels];
        if (img.channels == 1) { out.data[3 * i] = out.data[3 * i + 1] = out.data[3 * i + 2] = p[0]; }
        else { out.data[3 * i] = p[2]; out.data[3 * i + 1] = p[1]; out.data[3 * i + 2] = p[0]; }
    }
    return true;
}

bool ReadDepth16(const std::string& path, DecodedImage& out) {
    if (!DecodePng(path, out)) return false;
    if (out.channels != 1 || out.bitDepth != 16) { out = DecodedImage(); return false; }
    return true;
}

void WritePoseLine(std::ostream& os, const std::string& stamp, const SE3& Twc) {
    double q[4];
    Twc.quaternion(q);
    const Vector3d& t = Twc.translation();
    os << stamp << ' ' << t[0] << ' ' << t[1] << ' ' << t[2] << ' ' << q[0] << ' ' << q[1] << ' ' << q[2] << ' ' << q[3] << std::endl;
}

}  // namespace myslam

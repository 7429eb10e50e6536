// image_io.cpp -- the asset decoders / writers on either side of the hot path:
//   Radiance .hdr (RGBE)  -> linear float RGBA   == stbi_loadf(path, .., 4)  as Texture2D uses it for the skybox
//   PNG                   -> RGBA8               == stbi_load (path, .., 4)  as Texture2D uses it for the blue noise
//   writers: PNG (stored deflate), binary PPM, PFM
// Reference call sites: source/engine/resource/texture_2d.cpp:22-44 (stbi_is_hdr / stbi_loadf / stbi_load, error
// "Could not load image {path}").  stb_image (Conan stb cci.20210910) is not vendored in the reference tree, so this
// restates the published formats: RGBE per Ward's Radiance spec with stb's conversion rule
// (rgb * 2^(e - 136), e = 0 -> 0, alpha = 1), PNG per RFC 2083 (zlib inflate from the system libz).
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.h"

namespace vrt {
namespace {

bool read_file(const char* path, std::vector<uint8_t>& out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        long n = ftell(f);
        if (n > 0) { out.resize((size_t)n); rewind(f); ok = fread(out.data(), 1, out.size(), f) == out.size(); }
    }
    fclose(f);
    return ok;
}

// ---- Radiance RGBE ------------------------------------------------------------------------------------

bool is_hdr(const std::vector<uint8_t>& b)
{
    return (b.size() >= 11 && memcmp(b.data(), "#?RADIANCE\n", 11) == 0) || (b.size() >= 7 && memcmp(b.data(), "#?RGBE\n", 7) == 0);
}

inline void rgbe_to_float(const uint8_t* p, float* out)
{
    if (p[3] != 0) {
        float f = ldexpf(1.0f, (int)p[3] - (128 + 8));
        out[0] = p[0] * f; out[1] = p[1] * f; out[2] = p[2] * f;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
    out[3] = 1.0f;
}

bool decode_hdr(const std::vector<uint8_t>& b, uint32_t& W, uint32_t& H, std::vector<float>& px)
{
    size_t o = 0;
    auto line = [&](std::string& s) {
        s.clear();
        while (o < b.size() && b[o] != '\n') s.push_back((char)b[o++]);
        if (o < b.size()) o++;
        return true;
    };
    std::string s;
    line(s);
    if (s != "#?RADIANCE" && s != "#?RGBE") return false;
    bool fmt = false;
    for (;;) {
        if (o >= b.size()) return false;
        line(s);
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) return false;
    line(s);
    int h = 0, w = 0;
    if (sscanf(s.c_str(), "-Y %d +X %d", &h, &w) != 2 || h <= 0 || w <= 0 || h > 65536 || w > 65536) return false;
    W = (uint32_t)w; H = (uint32_t)h;
    px.assign((size_t)w * h * 4, 0.0f);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto flat_from = [&](size_t first_row, size_t first_px, const uint8_t* head) {
        // non-RLE data: plain RGBE quadruples; `head` (4 bytes already consumed) is the first of them
        size_t i = first_row * (size_t)w + first_px;
        if (head) { rgbe_to_float(head, &px[i * 4]); i++; }
        for (; i < (size_t)w * h; i++) {
            if (o + 4 > b.size()) return false;
            rgbe_to_float(&b[o], &px[i * 4]); o += 4;
        }
        return true;
    };
    if (w < 8 || w >= 32768) return flat_from(0, 0, nullptr);
    for (int y = 0; y < h; y++) {
        if (o + 4 > b.size()) return false;
        uint8_t c1 = b[o], c2 = b[o + 1], lh = b[o + 2], ll = b[o + 3];
        if (c1 != 2 || c2 != 2 || (lh & 0x80)) {
            if (y != 0) return false;                     // stb only accepts a flat file from its very first pixel
            uint8_t head[4] = {c1, c2, lh, ll};
            o += 4;
            return flat_from(0, 0, head);
        }
        o += 4;
        if ((((int)lh << 8) | ll) != w) return false;
        for (int k = 0; k < 4; k++) {
            int i = 0;
            while (i < w) {
                if (o >= b.size()) return false;
                int count = b[o++];
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || count > w - i || o >= b.size()) return false;
                    uint8_t v = b[o++];
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                } else {
                    if (count == 0 || count > w - i || o + (size_t)count > b.size()) return false;
                    for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = b[o++];
                }
            }
        }
        for (int x = 0; x < w; x++) rgbe_to_float(&scan[(size_t)x * 4], &px[((size_t)y * w + x) * 4]);
    }
    return true;
}

// ---- PNG ---------------------------------------------------------------------------------------------

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool decode_png(const std::vector<uint8_t>& b, uint32_t& W, uint32_t& H, std::vector<uint8_t>& px)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (b.size() < 8 || memcmp(b.data(), sig, 8) != 0) return false;
    size_t o = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false;
    while (o + 12 <= b.size()) {
        uint32_t len = be32(&b[o]);
        if (o + 12 + (size_t)len > b.size()) return false;
        const uint8_t* type = &b[o + 4];
        const uint8_t* data = &b[o + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return false;
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || w > 65536 || h > 65536 || interlace != 0) return false;   // Adam7 not supported
    int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!chans) return false;
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && (depth == 16 || plte.empty())) return false;
    size_t bpp_bits = (size_t)chans * depth, stride = ((size_t)w * bpp_bits + 7) / 8, fbpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    // unfilter in place
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = &raw[(stride + 1) * y];
        int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= fbpp ? cur[i - fbpp] : 0, up = prev[i], c = i >= fbpp ? prev[i - fbpp] : 0;
            int v = cur[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += up; break;
                case 3: v += (a + up) >> 1; break;
                case 4: v += paeth(a, up, c); break;
                default: return false;
            }
            cur[i] = (uint8_t)v;
        }
        memcpy(prev.data(), cur, stride);
    }
    W = w; H = h;
    px.assign((size_t)w * h * 4, 255);
    auto sample = [&](const uint8_t* cur, size_t idx) -> uint32_t {      // idx-th sample of the row, raw value
        if (depth == 8) return cur[idx];
        if (depth == 16) return (uint32_t)cur[idx * 2] << 8 | cur[idx * 2 + 1];
        size_t bit = idx * depth;
        return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    };
    auto to8 = [&](uint32_t v) -> uint8_t {                              // stb: 16-bit keeps the high byte; 1/2/4-bit gray scale up
        if (depth == 16) return (uint8_t)(v >> 8);
        if (depth == 8) return (uint8_t)v;
        return (uint8_t)(v * (depth == 1 ? 255u : depth == 2 ? 85u : 17u));
    };
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* cur = &raw[(stride + 1) * y + 1];
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* d = &px[((size_t)y * w + x) * 4];
            if (ctype == 3) {
                uint32_t i = sample(cur, x);
                if ((size_t)i * 3 + 2 >= plte.size()) return false;
                d[0] = plte[i * 3]; d[1] = plte[i * 3 + 1]; d[2] = plte[i * 3 + 2];
                d[3] = i < trns.size() ? trns[i] : 255;
            } else if (ctype == 0 || ctype == 4) {
                uint32_t g = sample(cur, (size_t)x * chans);
                d[0] = d[1] = d[2] = to8(g);
                if (ctype == 4) d[3] = to8(sample(cur, (size_t)x * 2 + 1));
                else if (trns.size() >= 2 && g == ((uint32_t)trns[0] << 8 | trns[1])) d[3] = 0;
            } else {
                uint32_t r = sample(cur, (size_t)x * chans), g = sample(cur, (size_t)x * chans + 1), bl = sample(cur, (size_t)x * chans + 2);
                d[0] = to8(r); d[1] = to8(g); d[2] = to8(bl);
                if (ctype == 6) d[3] = to8(sample(cur, (size_t)x * 4 + 3));
                else if (trns.size() >= 6 && r == ((uint32_t)trns[0] << 8 | trns[1]) && g == ((uint32_t)trns[2] << 8 | trns[3]) &&
                         bl == ((uint32_t)trns[4] << 8 | trns[5])) d[3] = 0;
            }
        }
    }
    return true;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, (uint32_t)crc32(0, &out[start], (uInt)(out.size() - start)));
}

} // namespace

int image_load(const char* path, LoadedImage& img, std::string& err)
{
    std::vector<uint8_t> b;
    if (!read_file(path, b)) { err = std::string("Could not load image ") + path; return VRT_ERR_IO; }
    img.is_hdr = is_hdr(b);
    bool ok = img.is_hdr ? decode_hdr(b, img.w, img.h, img.f32) : decode_png(b, img.w, img.h, img.u8);
    if (!ok) { err = std::string("Could not load image ") + path; return VRT_ERR_PARSE; }
    return VRT_OK;
}

int image_write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h, std::string& err)
{
    std::vector<uint8_t> raw; raw.reserve(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba8 + (size_t)y * w * 4, rgba8 + (size_t)(y + 1) * w * 4); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "png: deflate failed"; return VRT_ERR_IO; }
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr; put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr); png_chunk(out, "IDAT", comp); png_chunk(out, "IEND", {});
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { if (f) fclose(f); err = std::string("cannot write ") + path; return VRT_ERR_IO; }
    fclose(f);
    return VRT_OK;
}

int image_write_ppm(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h, std::string& err)
{
    FILE* f = fopen(path, "wb");
    if (!f) { err = std::string("cannot write ") + path; return VRT_ERR_IO; }
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) fwrite(rgba8 + i * 4, 1, 3, f);
    fclose(f);
    return VRT_OK;
}

int image_write_pfm(const char* path, const float* rgb, uint32_t w, uint32_t h, uint32_t stride_floats, std::string& err)
{
    FILE* f = fopen(path, "wb");
    if (!f) { err = std::string("cannot write ") + path; return VRT_ERR_IO; }
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);                    // little-endian, rows bottom-to-top
    for (uint32_t y = h; y-- > 0;)
        for (uint32_t x = 0; x < w; x++) fwrite(rgb + ((size_t)y * w + x) * stride_floats, sizeof(float), 3, f);
    fclose(f);
    return VRT_OK;
}

} // namespace vrt

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "image_io.h"
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
static void spit(const char* p, const std::vector<uint8_t>& b) { std::ofstream f(p, std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); }
int main(int argc, char** argv)
{
    // usage: img_fuzz ITERATIONS SCRATCH_DIR
    const int iters = argc > 1 ? std::atoi(argv[1]) : 1000;
    const std::string dir = argc > 2 ? argv[2] : "/tmp";
    std::mt19937 rng(7);
    std::string e;
    // a PNG from the library's own writer and a flat + an RLE Radiance file
    std::vector<uint8_t> px(37 * 23 * 4); for (auto& v : px) v = (uint8_t)rng();
    vrt::image_write_png((dir + "/a.png").c_str(), px.data(), 37, 23, e);
    {   std::string h = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 9 +X 40\n"; std::vector<uint8_t> b(h.begin(), h.end());
        for (int y = 0; y < 9; y++) { b.push_back(2); b.push_back(2); b.push_back(0); b.push_back(40);
            for (int c = 0; c < 4; c++) { b.push_back(128 + 20); b.push_back((uint8_t)(y * 7 + c)); b.push_back(20); for (int k = 0; k < 20; k++) b.push_back((uint8_t)rng()); } }
        spit((dir + "/a.hdr").c_str(), b);
        std::string h2 = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 5 +X 6\n"; std::vector<uint8_t> b2(h2.begin(), h2.end());
        for (int k = 0; k < 5 * 6 * 4; k++) b2.push_back((uint8_t)(rng() | 1)); spit((dir + "/b.hdr").c_str(), b2); }
    long ok = 0, err = 0;
    for (const std::string& fs : {dir + "/a.png", dir + "/a.hdr", dir + "/b.hdr"}) {
        const char* f = fs.c_str();
        std::vector<uint8_t> base = slurp(f);
        const char* ext = strrchr(f, '.');
        std::string tmp = dir + "/m" + ext;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> b = base;
            if (it > 0) {
                int kind = rng() % 4;
                if (kind == 0) b.resize(rng() % (b.size() + 1));
                else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 8); k++) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8));
                else if (kind == 2) for (int k = 0; k < 4; k++) b[rng() % b.size()] = (uint8_t)rng();
                else { size_t p = rng() % b.size(); uint32_t v = (rng() % 3 == 0) ? 0xFFFFFFFFu : (uint32_t)rng(); for (int k = 0; k < 4 && p + k < b.size(); k++) b[p + k] = (uint8_t)(v >> (8 * k)); }
            }
            spit(tmp.c_str(), b);
            vrt::LoadedImage img;
            int rc = vrt::image_load(tmp.c_str(), img, e);
            (rc == 0 ? ok : err)++;
        }
    }
    std::printf("images: %ld decoded, %ld rejected\n", ok, err);
    return 0;
}

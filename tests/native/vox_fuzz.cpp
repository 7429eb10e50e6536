// ASan/UBSan harness (CPU only): the product's .vox reader and image decoders on the fixtures and on random mutations of them.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "vox_reader.h"
#include "image_io.h"
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv)
{
    // usage: vox_fuzz ITERATIONS file.vox...
    const int iters = argc > 1 ? std::atoi(argv[1]) : 1000;
    std::mt19937 rng(123);
    long ok = 0, err = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> base = slurp(argv[a]);
        if (base.empty()) continue;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> b = base;
            if (it > 0) {
                int kind = rng() % 4;
                if (kind == 0) b.resize(rng() % (b.size() + 1));                                    // truncate
                else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 8); k++) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8));   // bit flips
                else if (kind == 2) for (int k = 0; k < 4; k++) b[rng() % b.size()] = (uint8_t)rng();  // random bytes
                else { size_t p = rng() % b.size(); uint32_t v = (rng() % 3 == 0) ? 0xFFFFFFFFu : (uint32_t)rng(); for (int k = 0; k < 4 && p + k < b.size(); k++) b[p + k] = (uint8_t)(v >> (8 * k)); }  // wild 32-bit field
            }
            vrt::FlatScene fs; std::string e;
            int rc = vrt::vox_flatten(b.data(), b.size(), fs, e);
            (rc == 0 ? ok : err)++;
        }
    }
    std::printf("vox: %ld parsed, %ld rejected\n", ok, err);
    return 0;
}

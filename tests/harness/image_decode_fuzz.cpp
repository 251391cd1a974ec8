// Test harness (not part of the product): decodes every blob of a corpus file with pth_decode_image, the way the scene
// loader does (a size query, then a fill into a buffer of exactly that size).  Built with the address and
// undefined-behaviour sanitizers by tests/test_image_decode_sanitized.py; any finding aborts the process.
// Corpus format: repeated { uint32 little-endian length, bytes }.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ptamd_host.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> all;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) all.insert(all.end(), buf, buf + got);
    std::fclose(f);
    size_t at = 0, images = 0, errors = 0, disagreements = 0;
    while (at + 4 <= all.size()) {
        uint32_t n;
        std::memcpy(&n, &all[at], 4);
        at += 4;
        if (n > all.size() - at) return 2;
        // a copy of exactly n bytes, so that a read past the end of the file is a heap overflow the sanitizer sees
        std::vector<uint8_t> file(all.begin() + at, all.begin() + at + n);
        at += n;
        for (int want = 0; want <= 4; want += (want == 0 ? 1 : (want == 1 ? 2 : 1))) {  // 0, 1, 3, 4
            int w = 0, h = 0, c = 0;
            if (pth_decode_image(nullptr, file.data(), file.size(), want, &w, &h, &c, nullptr, 0) != 0) {
                ++errors;
                continue;
            }
            if (w <= 0 || h <= 0 || c <= 0 || c > 4) return 3;
            std::vector<uint8_t> pixels(size_t(w) * size_t(h) * size_t(c));
            int w2 = 0, h2 = 0, c2 = 0;
            if (pth_decode_image(nullptr, file.data(), file.size(), want, &w2, &h2, &c2, pixels.data(), pixels.size()) != 0) {
                ++errors;
                continue;
            }
            if (w2 != w || h2 != h || c2 != c) ++disagreements;
            // one byte too few must be refused, not overrun
            if (pixels.size() > 1 && pth_decode_image(nullptr, file.data(), file.size(), want, &w2, &h2, &c2, pixels.data(), pixels.size() - 1) == 0) return 4;
            ++images;
        }
    }
    std::printf("images %zu errors %zu disagreements %zu\n", images, errors, disagreements);
    return disagreements ? 5 : 0;
}

// Image decoding for the scene loader's C route: baseline / extended-sequential / progressive JPEG (ITU-T T.81,
// Huffman coding, 8 bits per sample, 1 or 3 components) and PNG (ISO/IEC 15948: all colour types and bit depths,
// Adam7 interlace, zlib "deflate" streams of RFC 1950 / 1951).  Written from those specifications for this library;
// the reference decodes its textures with the stb_image it vendors (image.cpp:58-103, gltf_model.cpp:96-152), which
// is not linked here.  Differences from that decoder are rounding only: the inverse DCT is done in floating point
// and the chroma of subsampled JPEGs is interpolated with the usual (3, 1) / 4 triangle weights per direction, which
// keeps every channel within 3 / 255 of stb_image's output on the helmet scene's textures (tests/test_image_decode.py).
//
// pth_decode_image has the signature of pth_image_decoder and can be handed to pth_scene_load as is.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptamd_host.h"

namespace {

struct DecodeError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void Fail(const char *what) { throw DecodeError(what); }

struct Pixels {
    int w = 0, h = 0, c = 0;
    std::vector<uint8_t> px;  // rows top to bottom, c interleaved channels
};

// ------------------------------------------------------------------------------------------------ JPEG
constexpr uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t counts[17] = {};   // codes of each length 1..16
    uint8_t symbols[256] = {};
    int32_t max_code[18] = {}; // largest code of each length, -1 when there is none
    int32_t val_offset[17] = {};
    uint16_t fast[512] = {};   // 9 leading bits -> (length << 8) | symbol, 0: longer code

    void Build() {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            val_offset[len] = k - code;
            if (counts[len]) {
                for (int i = 0; i < counts[len]; ++i, ++code, ++k) {
                    if (len <= 9) {
                        const int first = code << (9 - len), n = 1 << (9 - len);
                        if (first + n > 512) Fail("JPEG: Huffman table assigns more codes than its lengths allow");
                        for (int f = 0; f < n; ++f) fast[first + f] = uint16_t((len << 8) | symbols[k]);
                    }
                }
                max_code[len] = code - 1;
                if (code > (1 << len)) Fail("JPEG: Huffman table assigns more codes than its lengths allow");
            } else {
                max_code[len] = -1;
            }
            code <<= 1;
        }
        max_code[17] = 0x7FFFFFFF;
        present = true;
    }
};

// Entropy-coded segment: bytes with 0xFF00 stuffing; a marker ends the data (zeros are supplied from there on).
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int cnt = 0;
    bool at_marker = false;

    void Fill() {
        while (cnt <= 24) {
            uint32_t b = 0;
            if (!at_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    const uint8_t next = (p + 1 < end) ? p[1] : 0xD9;
                    if (next == 0) {
                        p += 2;
                    } else {
                        at_marker = true;  // p stays on the 0xFF of the marker
                        b = 0;
                    }
                } else {
                    ++p;
                }
            }
            acc |= b << (24 - cnt);
            cnt += 8;
        }
    }
    int Bits(int n) {
        if (n == 0) return 0;
        if (cnt < n) Fill();
        const int v = int(acc >> (32 - n));
        acc <<= n;
        cnt -= n;
        return v;
    }
    int Bit() { return Bits(1); }
    int Symbol(const HuffTable &t) {
        if (cnt < 16) Fill();
        const uint16_t f = t.fast[acc >> 23];
        if (f) {
            const int len = f >> 8;
            acc <<= len;
            cnt -= len;
            return f & 0xFF;
        }
        int code = int(acc >> 22);  // the ten leading bits
        int len = 10;
        while (len <= 16 && code > t.max_code[len]) {
            ++len;
            code = int(acc >> (32 - len));
        }
        if (len > 16) Fail("JPEG: bits that are no Huffman code");
        acc <<= len;
        cnt -= len;
        const int index = code + t.val_offset[len];
        if (index < 0 || index > 255) Fail("JPEG: bits that are no Huffman code");
        return t.symbols[index];
    }
    void Reset() {
        acc = 0;
        cnt = 0;
        at_marker = false;
    }
};

inline int Extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;          // tables of the current scan
    int blocks_w = 0, blocks_h = 0;    // allocated (whole MCUs)
    int own_w = 0, own_h = 0;          // blocks that hold samples (what a scan of this component alone covers)
    int pred = 0;
    std::vector<int16_t> coef;   // blocks_w * blocks_h * 64, natural order inside a block
    std::vector<uint8_t> plane;  // blocks_w * 8 by blocks_h * 8 samples
};

struct Jpeg {
    const uint8_t *p, *end;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, have_frame = false;
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    uint16_t quant[4][64] = {};
    bool quant_present[4] = {};
    HuffTable dc[4], ac[4];
    Component comp[3];

    int Byte() {
        if (p >= end) Fail("JPEG: file ends inside a segment");
        return *p++;
    }
    int Word() {
        const int hi = Byte();
        return (hi << 8) | Byte();
    }

    void ReadFrameHeader(int marker) {
        const int len = Word();
        if (Byte() != 8) Fail("JPEG: only 8 bits per sample are supported");
        height = Word();
        width = Word();
        ncomp = Byte();
        if (len != 8 + 3 * ncomp) Fail("JPEG: bad frame header length");
        if (width <= 0 || height <= 0) Fail("JPEG: image has no size");
        if (width > 32768 || height > 32768) Fail("JPEG: image larger than 32768 pixels a side");
        if (ncomp != 1 && ncomp != 3) Fail("JPEG: only grey and three-component images are supported");
        progressive = marker == 0xC2;
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.id = Byte();
            const int hv = Byte();
            c.h = hv >> 4;
            c.v = hv & 15;
            c.tq = Byte();
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) Fail("JPEG: sampling factors other than 1 and 2 are not supported");
            if (c.tq > 3) Fail("JPEG: bad quantisation table number");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a lone component is never interleaved
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.blocks_w = mcus_x * c.h;
            c.blocks_h = mcus_y * c.v;
            const int sw = (width * c.h + hmax - 1) / hmax, sh = (height * c.v + vmax - 1) / vmax;
            c.own_w = (sw + 7) / 8;
            c.own_h = (sh + 7) / 8;
        }
        have_frame = true;
    }

    void ReadTables(int marker) {
        int len = Word() - 2;
        if (len < 0 || p + len > end) Fail("JPEG: segment runs past the end of the file");
        const uint8_t *seg_end = p + len;
        if (marker == 0xDB) {
            while (p < seg_end) {
                const int pq = Byte();
                const int t = pq & 15, wide = pq >> 4;
                if (t > 3 || wide > 1) Fail("JPEG: bad quantisation table");
                for (int k = 0; k < 64; ++k) quant[t][kZigzag[k]] = uint16_t(wide ? Word() : Byte());
                quant_present[t] = true;
            }
        } else if (marker == 0xC4) {
            while (p < seg_end) {
                const int tc = Byte();
                const int t = tc & 15, cls = tc >> 4;
                if (t > 3 || cls > 1) Fail("JPEG: bad Huffman table number");
                HuffTable &h = cls ? ac[t] : dc[t];
                h = HuffTable();
                int total = 0;
                for (int l = 1; l <= 16; ++l) total += (h.counts[l] = uint8_t(Byte()));
                if (total > 256) Fail("JPEG: Huffman table with more than 256 symbols");
                for (int k = 0; k < total; ++k) h.symbols[k] = uint8_t(Byte());
                h.Build();
            }
        } else if (marker == 0xDD) {
            restart_interval = Word();
        }
        if (p != seg_end) Fail("JPEG: table segment has the wrong length");
    }

    // ---- one block of each kind of scan (T.81 F.2.2, G.1.2)
    void BlockSequential(BitReader &br, Component &c, int16_t *b) {
        const int s = br.Symbol(dc[c.td]);
        if (s > 11) Fail("JPEG: bad DC difference size");
        c.pred += Extend(br.Bits(s), s);
        b[0] = int16_t(c.pred);
        const HuffTable &t = ac[c.ta];
        for (int k = 1; k < 64;) {
            const int rs = br.Symbol(t);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r != 15) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) Fail("JPEG: coefficient index beyond the block");
                b[kZigzag[k]] = int16_t(Extend(br.Bits(sz), sz));
                ++k;
            }
        }
    }
    void BlockDcFirst(BitReader &br, Component &c, int16_t *b, int al) {
        const int s = br.Symbol(dc[c.td]);
        if (s > 11) Fail("JPEG: bad DC difference size");
        c.pred += Extend(br.Bits(s), s);
        b[0] = int16_t(c.pred * (1 << al));
    }
    static void BlockDcRefine(BitReader &br, int16_t *b, int al) {
        if (br.Bit()) b[0] = int16_t(b[0] | (1 << al));
    }
    void BlockAcFirst(BitReader &br, const Component &c, int16_t *b, int ss, int se, int al, int &eobrun) {
        if (eobrun > 0) {
            --eobrun;
            return;
        }
        const HuffTable &t = ac[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = br.Symbol(t);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r < 15) {  // end of band for this block and the next (1 << r) + extra - 1 blocks
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += br.Bits(r);
                    break;
                }
                k += 16;
            } else {
                k += r;
                if (k > se) Fail("JPEG: coefficient index beyond the band");
                b[kZigzag[k]] = int16_t(Extend(br.Bits(sz), sz) * (1 << al));
                ++k;
            }
        }
    }
    static void RefineNonZero(BitReader &br, int16_t &coef, int bit) {
        if (br.Bit() && (coef & bit) == 0) coef = int16_t(coef >= 0 ? coef + bit : coef - bit);
    }
    void BlockAcRefine(BitReader &br, const Component &c, int16_t *b, int ss, int se, int al, int &eobrun) {
        const int bit = 1 << al;
        int k = ss;
        if (eobrun == 0) {
            const HuffTable &t = ac[c.ta];
            while (k <= se) {
                const int rs = br.Symbol(t);
                int r = rs >> 4;
                const int sz = rs & 15;
                int value = 0;
                if (sz == 0) {
                    if (r < 15) {
                        eobrun = 1 << r;
                        if (r) eobrun += br.Bits(r);
                        break;
                    }
                } else {
                    if (sz != 1) Fail("JPEG: refinement scan with a coefficient size other than 1");
                    value = br.Bit() ? bit : -bit;
                }
                // pass r coefficients that are still zero; those already non-zero on the way take a correction bit each
                while (k <= se) {
                    int16_t &coef = b[kZigzag[k]];
                    if (coef != 0) {
                        RefineNonZero(br, coef, bit);
                    } else {
                        if (r == 0) {
                            if (value) coef = int16_t(value);
                            ++k;
                            break;
                        }
                        --r;
                    }
                    ++k;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t &coef = b[kZigzag[k]];
                if (coef != 0) RefineNonZero(br, coef, bit);
            }
            --eobrun;
        }
    }

    void ReadScan() {
        if (!have_frame) Fail("JPEG: scan before the frame header");
        const int len = Word();
        const int ns = Byte();
        if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) Fail("JPEG: bad scan header");
        Component *sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = Byte(), tt = Byte();
            sc[i] = nullptr;
            for (int k = 0; k < ncomp; ++k)
                if (comp[k].id == id) sc[i] = &comp[k];
            if (!sc[i]) Fail("JPEG: scan names a component the frame does not have");
            for (int j = 0; j < i; ++j)
                if (sc[j] == sc[i]) Fail("JPEG: scan names a component twice");
            sc[i]->td = tt >> 4;
            sc[i]->ta = tt & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) Fail("JPEG: bad Huffman table number");
        }
        int ss = Byte(), se = Byte();
        const int a = Byte();
        int ah = a >> 4, al = a & 15;
        if (!progressive) {
            ss = 0;
            se = 63;
            ah = al = 0;
        } else {
            if (ss > se || se > 63 || al > 13 || ah > 13) Fail("JPEG: bad spectral selection");
            if (ss == 0 && se != 0) Fail("JPEG: progressive scan mixes DC and AC coefficients");
            if (ss > 0 && ns != 1) Fail("JPEG: AC scan over several components");
            if (ah != 0 && ah != al + 1) Fail("JPEG: bad successive approximation");
        }
        for (int i = 0; i < ns; ++i) {
            const bool need_dc = ss == 0 && ah == 0, need_ac = se > 0;
            if (need_dc && !dc[sc[i]->td].present) Fail("JPEG: scan uses a Huffman table that was never defined");
            if (need_ac && !ac[sc[i]->ta].present) Fail("JPEG: scan uses a Huffman table that was never defined");
            if (sc[i]->coef.empty()) sc[i]->coef.assign(size_t(sc[i]->blocks_w) * sc[i]->blocks_h * 64, 0);
            sc[i]->pred = 0;
        }
        BitReader br{p, end};
        int eobrun = 0;
        auto block = [&](Component &c, int bx, int by) {
            int16_t *b = c.coef.data() + (size_t(by) * c.blocks_w + bx) * 64;
            if (!progressive) BlockSequential(br, c, b);
            else if (ss == 0) ah == 0 ? BlockDcFirst(br, c, b, al) : BlockDcRefine(br, b, al);
            else ah == 0 ? BlockAcFirst(br, c, b, ss, se, al, eobrun) : BlockAcRefine(br, c, b, ss, se, al, eobrun);
        };
        const bool interleaved = ns > 1;
        const int units_x = interleaved ? mcus_x : sc[0]->own_w, units_y = interleaved ? mcus_y : sc[0]->own_h;
        int until_restart = restart_interval;
        for (int uy = 0; uy < units_y; ++uy) {
            for (int ux = 0; ux < units_x; ++ux) {
                if (restart_interval && until_restart == 0) {
                    // the entropy-coded bytes stop at the restart marker; everything starts afresh behind it
                    if (!br.at_marker) br.Fill();
                    const uint8_t *q = br.p;
                    while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) Fail("JPEG: restart marker missing");
                        ++q;
                    }
                    if (q + 1 >= end) Fail("JPEG: restart marker missing");
                    br.p = q + 2;
                    br.Reset();
                    for (int i = 0; i < ns; ++i) sc[i]->pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                if (interleaved) {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h) block(*sc[i], ux * sc[i]->h + h, uy * sc[i]->v + v);
                } else {
                    block(*sc[0], ux, uy);
                }
                --until_restart;
            }
        }
        // continue behind the entropy-coded data: at the marker the reader stopped at, or at the next one
        // (the reader never takes a byte from behind a marker, so the next marker is at br.p or after it)
        const uint8_t *q = br.p;
        while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
        p = q + 1 < end ? q : end;
    }

    // ---- coefficients -> samples
    void InverseDct() {
        static float basis[8][8];  // basis[x][u] = c(u) / 2 * cos((2x + 1) u pi / 16)
        static bool ready = false;
        if (!ready) {
            for (int x = 0; x < 8; ++x)
                for (int u = 0; u < 8; ++u) basis[x][u] = float((u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
            ready = true;
        }
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            if (c.coef.empty()) Fail("JPEG: a component has no scan");
            if (!quant_present[c.tq]) Fail("JPEG: a component uses a quantisation table that was never defined");
            const uint16_t *q = quant[c.tq];
            const int stride = c.blocks_w * 8;
            c.plane.assign(size_t(stride) * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    const int16_t *b = c.coef.data() + (size_t(by) * c.blocks_w + bx) * 64;
                    float f[64], t[64];
                    for (int k = 0; k < 64; ++k) f[k] = float(int(b[k]) * int(q[k]));
                    for (int y = 0; y < 8; ++y)      // rows: t[y][x] = sum_u f[y][u] basis[x][u]
                        for (int x = 0; x < 8; ++x) {
                            float s = 0.0f;
                            for (int u = 0; u < 8; ++u) s += f[y * 8 + u] * basis[x][u];
                            t[y * 8 + x] = s;
                        }
                    uint8_t *out = c.plane.data() + size_t(by) * 8 * stride + bx * 8;
                    for (int x = 0; x < 8; ++x)      // columns
                        for (int y = 0; y < 8; ++y) {
                            float s = 0.0f;
                            for (int v = 0; v < 8; ++v) s += t[v * 8 + x] * basis[y][v];
                            const int level = int(std::floor(s + 128.5f));
                            out[size_t(y) * stride + x] = uint8_t(level < 0 ? 0 : (level > 255 ? 255 : level));
                        }
                }
            std::vector<int16_t>().swap(c.coef);
        }
    }

    // A component's plane at full resolution: chroma of subsampled images is interpolated with weights (3, 1) / 4
    // between the two nearest samples in each subsampled direction (sample centres of a 2:1 plane lie at +- 1/4 pixel).
    std::vector<uint8_t> FullPlane(const Component &c) const {
        const int fx = hmax / c.h, fy = vmax / c.v;
        const int stride = c.blocks_w * 8;
        const int sw = (width * c.h + hmax - 1) / hmax, sh = (height * c.v + vmax - 1) / vmax;  // samples that mean something
        std::vector<uint8_t> out(size_t(width) * height);
        std::vector<int> row(static_cast<size_t>(sw), 0);
        for (int y = 0; y < height; ++y) {
            // vertical step: row[x] = 4 * sample (fy == 1) or 3 * near + far (fy == 2)
            int y0 = y, y1 = y;
            if (fy == 2) {
                y0 = y >> 1;
                y1 = (y & 1) ? y0 + 1 : y0 - 1;
                y1 = y1 < 0 ? 0 : (y1 >= sh ? sh - 1 : y1);
            }
            const uint8_t *near = c.plane.data() + size_t(y0) * stride, *far = c.plane.data() + size_t(y1) * stride;
            for (int x = 0; x < sw; ++x) row[size_t(x)] = 3 * near[x] + far[x];
            uint8_t *o = out.data() + size_t(y) * width;
            if (fx == 1) {
                for (int x = 0; x < width; ++x) o[x] = uint8_t((row[size_t(x)] + 2) >> 2);
            } else {
                for (int x = 0; x < width; ++x) {
                    const int x0 = x >> 1;
                    int x1 = (x & 1) ? x0 + 1 : x0 - 1;
                    x1 = x1 < 0 ? 0 : (x1 >= sw ? sw - 1 : x1);
                    o[x] = uint8_t((3 * row[size_t(x0)] + row[size_t(x1)] + 8) >> 4);
                }
            }
        }
        return out;
    }

    Pixels Decode(bool header_only) {
        if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) Fail("JPEG: no start-of-image marker");
        p += 2;
        bool done = false;
        while (!done) {
            int m = Byte();
            if (m != 0xFF) Fail("JPEG: marker expected");
            while ((m = Byte()) == 0xFF) {
            }
            switch (m) {
                case 0xC0: case 0xC1: case 0xC2:
                    if (have_frame) Fail("JPEG: two frame headers");
                    ReadFrameHeader(m);
                    if (header_only) {
                        Pixels r;
                        r.w = width;
                        r.h = height;
                        r.c = ncomp;
                        return r;
                    }
                    break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    Fail("JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
                case 0xC4: case 0xDB: case 0xDD: ReadTables(m); break;
                case 0xDA: ReadScan(); break;
                case 0xD9: done = true; break;
                case 0x01: case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7: break;  // no segment
                default: {
                    const int len = Word();
                    if (len < 2 || p + (len - 2) > end) Fail("JPEG: segment runs past the end of the file");
                    p += len - 2;
                }
            }
            if (!done && p >= end) {
                if (have_frame && !comp[0].coef.empty()) break;  // no end-of-image marker: decode what the scans gave
                Fail("JPEG: file ends before any image data");
            }
        }
        if (!have_frame) Fail("JPEG: no frame header");
        InverseDct();
        Pixels r;
        r.w = width;
        r.h = height;
        r.c = ncomp;
        r.px.resize(size_t(width) * height * ncomp);
        if (ncomp == 1) {
            const std::vector<uint8_t> y = FullPlane(comp[0]);
            std::memcpy(r.px.data(), y.data(), y.size());
        } else {
            const std::vector<uint8_t> y = FullPlane(comp[0]), cb = FullPlane(comp[1]), cr = FullPlane(comp[2]);
            for (size_t i = 0; i < y.size(); ++i) {  // T.871 (JFIF) conversion
                const float Y = float(y[i]), Cb = float(cb[i]) - 128.0f, Cr = float(cr[i]) - 128.0f;
                const float ch[3] = {Y + 1.402f * Cr, Y - 0.344136f * Cb - 0.714136f * Cr, Y + 1.772f * Cb};
                for (int k = 0; k < 3; ++k) {
                    const int v = int(std::floor(ch[k] + 0.5f));
                    r.px[i * 3 + size_t(k)] = uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v));
                }
            }
        }
        return r;
    }
};

// ------------------------------------------------------------------------------------------------ inflate
struct Inflate {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int cnt = 0;
    std::vector<uint8_t> &out;
    size_t limit;

    Inflate(const uint8_t *b, const uint8_t *e, std::vector<uint8_t> &o, size_t lim) : p(b), end(e), out(o), limit(lim) {}

    int Bits(int n) {
        while (cnt < n) {
            if (p >= end) Fail("PNG: compressed data ends early");
            acc |= uint32_t(*p++) << cnt;
            cnt += 8;
        }
        const int v = int(acc & ((1u << n) - 1u));
        acc >>= n;
        cnt -= n;
        return v;
    }
    struct Codes {
        uint16_t count[16] = {}, symbol[320] = {};
        void Build(const uint8_t *lengths, int n) {
            std::memset(count, 0, sizeof(count));
            for (int i = 0; i < n; ++i) count[lengths[i]]++;
            count[0] = 0;
            int left = 1;
            for (int l = 1; l < 16; ++l) {
                left = (left << 1) - count[l];
                if (left < 0) Fail("PNG: over-subscribed Huffman code");
            }
            uint16_t offs[16];
            offs[1] = 0;
            for (int l = 1; l < 15; ++l) offs[l + 1] = uint16_t(offs[l] + count[l]);
            for (int i = 0; i < n; ++i)
                if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
        }
    };
    int Symbol(const Codes &c) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= Bits(1);
            const int n = c.count[l];
            if (code - n < first) return c.symbol[index + (code - first)];
            index += n;
            first += n;
            first <<= 1;
            code <<= 1;
        }
        Fail("PNG: bits that are no Huffman code");
    }
    void Block(const Codes &lit, const Codes &dist) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            const int s = Symbol(lit);
            if (s < 256) {
                if (out.size() >= limit) Fail("PNG: more image data than the header announces");
                out.push_back(uint8_t(s));
            } else if (s == 256) {
                return;
            } else {
                if (s > 285) Fail("PNG: bad length symbol");
                const int len = lbase[s - 257] + Bits(lext[s - 257]);
                const int ds = Symbol(dist);
                if (ds > 29) Fail("PNG: bad distance symbol");
                const size_t d = size_t(dbase[ds]) + size_t(Bits(dext[ds]));
                if (d > out.size()) Fail("PNG: distance reaches before the start of the data");
                if (out.size() + size_t(len) > limit) Fail("PNG: more image data than the header announces");
                for (int k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
            }
        }
    }
    void Run() {
        const int cmf = Bits(8), flg = Bits(8);
        if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) Fail("PNG: bad zlib header");
        int last;
        do {
            last = Bits(1);
            const int type = Bits(2);
            if (type == 0) {
                acc = 0;
                cnt = 0;
                if (end - p < 4) Fail("PNG: compressed data ends early");
                const int len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
                p += 4;
                if ((len ^ 0xFFFF) != nlen || end - p < len) Fail("PNG: bad stored block");
                if (out.size() + size_t(len) > limit) Fail("PNG: more image data than the header announces");
                out.insert(out.end(), p, p + len);
                p += len;
            } else if (type == 1) {
                uint8_t l[320];
                for (int i = 0; i < 144; ++i) l[i] = 8;
                for (int i = 144; i < 256; ++i) l[i] = 9;
                for (int i = 256; i < 280; ++i) l[i] = 7;
                for (int i = 280; i < 288; ++i) l[i] = 8;
                Codes lit, dist;
                lit.Build(l, 288);
                for (int i = 0; i < 30; ++i) l[i] = 5;
                dist.Build(l, 30);
                Block(lit, dist);
            } else if (type == 2) {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = Bits(5) + 257, ndist = Bits(5) + 1, ncode = Bits(4) + 4;
                if (nlen > 286 || ndist > 30) Fail("PNG: bad code counts");
                uint8_t l[320] = {};
                for (int i = 0; i < ncode; ++i) l[order[i]] = uint8_t(Bits(3));
                Codes cl;
                cl.Build(l, 19);
                uint8_t lengths[320] = {};
                for (int i = 0; i < nlen + ndist;) {
                    const int s = Symbol(cl);
                    if (s < 16) {
                        lengths[i++] = uint8_t(s);
                    } else {
                        int rep, val = 0;
                        if (s == 16) {
                            if (i == 0) Fail("PNG: repeat with nothing to repeat");
                            val = lengths[i - 1];
                            rep = 3 + Bits(2);
                        } else if (s == 17) {
                            rep = 3 + Bits(3);
                        } else {
                            rep = 11 + Bits(7);
                        }
                        if (i + rep > nlen + ndist) Fail("PNG: code lengths run past their count");
                        while (rep--) lengths[i++] = uint8_t(val);
                    }
                }
                if (lengths[256] == 0) Fail("PNG: no end-of-block code");
                Codes lit, dist;
                lit.Build(lengths, nlen);
                dist.Build(lengths + nlen, ndist);
                Block(lit, dist);
            } else {
                Fail("PNG: bad block type");
            }
        } while (!last);
    }
};

// ------------------------------------------------------------------------------------------------ PNG
struct Png {
    const uint8_t *p, *end;
    uint32_t width = 0, height = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> palette, alpha;  // PLTE (rgb), tRNS
    bool has_trns = false;
    uint16_t trns_key[3] = {};

    static uint32_t Be32(const uint8_t *q) { return (uint32_t(q[0]) << 24) | (uint32_t(q[1]) << 16) | (uint32_t(q[2]) << 8) | q[3]; }
    int Channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
    int OutChannels() const {
        if (color == 3) return has_trns ? 4 : 3;
        const int c = Channels();
        return (has_trns && (c == 1 || c == 3)) ? c + 1 : c;
    }

    static void Unfilter(uint8_t *row, const uint8_t *prev, size_t n, size_t bpp, int type) {
        for (size_t i = 0; i < n; ++i) {
            const int a = i >= bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            switch (type) {
                case 0: break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: Fail("PNG: bad filter type");
            }
            row[i] = uint8_t(row[i] + pred);
        }
    }

    Pixels Decode(bool header_only) {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        if (end - p < 8 || std::memcmp(p, sig, 8) != 0) Fail("PNG: bad signature");
        p += 8;
        std::vector<uint8_t> idat;
        bool have_header = false, ended = false;
        while (!ended) {
            if (end - p < 12) Fail("PNG: file ends inside a chunk");
            const uint32_t len = Be32(p);
            const uint8_t *type = p + 4, *data = p + 8;
            if (len > 0x7FFFFFFFu || size_t(end - data) < size_t(len) + 4) Fail("PNG: chunk runs past the end of the file");
            p = data + len + 4;  // the checksum is not verified
            auto is = [&](const char *t) { return std::memcmp(type, t, 4) == 0; };
            if (is("IHDR")) {
                if (len != 13) Fail("PNG: bad header chunk");
                width = Be32(data);
                height = Be32(data + 4);
                depth = data[8];
                color = data[9];
                interlace = data[12];
                if (width == 0 || height == 0 || width > 32768 || height > 32768) Fail("PNG: image size out of range");
                if (data[10] != 0 || data[11] != 0 || interlace > 1) Fail("PNG: unknown compression, filter or interlace method");
                const bool ok = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                                (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                                ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16));
                if (!ok) Fail("PNG: bad colour type / bit depth");
                have_header = true;
            } else if (!have_header) {
                Fail("PNG: first chunk is not the header");
            } else if (is("PLTE")) {
                if (len % 3 != 0 || len > 768) Fail("PNG: bad palette");
                palette.assign(data, data + len);
            } else if (is("tRNS")) {
                has_trns = true;
                if (color == 3) alpha.assign(data, data + len);
                else if (color == 0 && len >= 2) trns_key[0] = uint16_t((data[0] << 8) | data[1]);
                else if (color == 2 && len >= 6)
                    for (int k = 0; k < 3; ++k) trns_key[k] = uint16_t((data[2 * k] << 8) | data[2 * k + 1]);
                else has_trns = false;
            } else if (is("IDAT")) {
                if (header_only) {
                    ended = true;
                } else {
                    idat.insert(idat.end(), data, data + len);
                }
            } else if (is("IEND")) {
                ended = true;
            } else if (!(type[0] & 32)) {
                Fail("PNG: unknown critical chunk");
            }
        }
        if (!have_header) Fail("PNG: no header");
        Pixels r;
        r.w = int(width);
        r.h = int(height);
        r.c = OutChannels();
        if (header_only) return r;
        if (color == 3 && palette.empty()) Fail("PNG: palette image without a palette");

        const int ch = Channels();
        const size_t bits_pp = size_t(ch) * size_t(depth), bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
        struct Pass { uint32_t x0, y0, dx, dy; };
        static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        static const Pass whole = {0, 0, 1, 1};
        size_t raw_size = 0;
        for (int k = 0; k < (interlace ? 7 : 1); ++k) {
            const Pass &ps = interlace ? adam7[k] : whole;
            const size_t pw = (width - ps.x0 + ps.dx - 1) / ps.dx, ph = (height - ps.y0 + ps.dy - 1) / ps.dy;
            if (width > ps.x0 && height > ps.y0 && pw && ph) raw_size += ph * (1 + (pw * bits_pp + 7) / 8);
        }
        std::vector<uint8_t> raw;
        raw.reserve(raw_size);
        Inflate(idat.data(), idat.data() + idat.size(), raw, raw_size).Run();
        if (raw.size() != raw_size) Fail("PNG: less image data than the header announces");

        r.px.assign(size_t(width) * height * size_t(r.c), 255);
        size_t at = 0;
        for (int k = 0; k < (interlace ? 7 : 1); ++k) {
            const Pass &ps = interlace ? adam7[k] : whole;
            if (width <= ps.x0 || height <= ps.y0) continue;
            const size_t pw = (width - ps.x0 + ps.dx - 1) / ps.dx, ph = (height - ps.y0 + ps.dy - 1) / ps.dy;
            const size_t row_bytes = (pw * bits_pp + 7) / 8;
            const uint8_t *prev = nullptr;
            for (size_t y = 0; y < ph; ++y) {
                const int filter = raw[at++];
                uint8_t *row = raw.data() + at;
                Unfilter(row, prev, row_bytes, bpp, filter);
                prev = row;
                at += row_bytes;
                for (size_t x = 0; x < pw; ++x) {
                    uint16_t s[4] = {0, 0, 0, 0};  // the pixel's samples at their own bit depth
                    for (int c = 0; c < ch; ++c) {
                        const size_t i = x * size_t(ch) + size_t(c);
                        if (depth == 8) s[c] = row[i];
                        else if (depth == 16) s[c] = uint16_t((row[2 * i] << 8) | row[2 * i + 1]);
                        else {
                            const size_t bit = i * size_t(depth);
                            s[c] = uint16_t((row[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1));
                        }
                    }
                    uint8_t *o = r.px.data() + ((size_t(ps.y0) + y * ps.dy) * width + ps.x0 + x * ps.dx) * size_t(r.c);
                    auto to8 = [&](uint16_t v) { return uint8_t(depth == 16 ? v >> 8 : depth == 8 ? v : v * 255 / ((1 << depth) - 1)); };
                    if (color == 3) {
                        if (size_t(s[0]) * 3 + 2 >= palette.size()) Fail("PNG: palette index out of range");
                        o[0] = palette[s[0] * 3];
                        o[1] = palette[s[0] * 3 + 1];
                        o[2] = palette[s[0] * 3 + 2];
                        if (has_trns) o[3] = s[0] < alpha.size() ? alpha[s[0]] : 255;
                    } else {
                        for (int c = 0; c < ch; ++c) o[c] = to8(s[c]);
                        if (has_trns && ch == 1) o[1] = s[0] == trns_key[0] ? 0 : 255;
                        if (has_trns && ch == 3) o[3] = (s[0] == trns_key[0] && s[1] == trns_key[1] && s[2] == trns_key[2]) ? 0 : 255;
                    }
                }
            }
        }
        return r;
    }
};

Pixels DecodeAny(const uint8_t *bytes, size_t n, bool header_only) {
    if (n >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        Jpeg j{};
        j.p = bytes;
        j.end = bytes + n;
        return j.Decode(header_only);
    }
    if (n >= 8 && bytes[0] == 0x89 && bytes[1] == 'P') {
        Png g{};
        g.p = bytes;
        g.end = bytes + n;
        return g.Decode(header_only);
    }
    Fail("image is neither a JPEG nor a PNG file");
}

// 1 (grey), 2 (grey + alpha), 3 (RGB) or 4 (RGBA) channels -> `want` channels (3 or 4)
void Convert(const Pixels &in, int want, uint8_t *out) {
    const size_t n = size_t(in.w) * in.h;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *s = in.px.data() + i * size_t(in.c);
        uint8_t rgba[4];
        if (in.c <= 2) {
            rgba[0] = rgba[1] = rgba[2] = s[0];
            rgba[3] = in.c == 2 ? s[1] : 255;
        } else {
            rgba[0] = s[0];
            rgba[1] = s[1];
            rgba[2] = s[2];
            rgba[3] = in.c == 4 ? s[3] : 255;
        }
        std::memcpy(out + i * size_t(want), rgba, size_t(want));
    }
}

thread_local std::string g_decode_error;

}  // namespace

extern "C" {

int pth_decode_image(void *, const uint8_t *file_bytes, size_t n_bytes, int want_channels, int *width, int *height, int *channels,
                     uint8_t *pixels, size_t capacity) {
    try {
        if (!file_bytes || !width || !height || !channels) Fail("null argument");
        if (want_channels != 0 && want_channels != 3 && want_channels != 4) Fail("want_channels must be 0, 3 or 4");
        const Pixels r = DecodeAny(file_bytes, n_bytes, pixels == nullptr);
        const int out_c = want_channels ? want_channels : r.c;
        *width = r.w;
        *height = r.h;
        *channels = out_c;
        if (!pixels) return 0;
        const size_t need = size_t(r.w) * r.h * size_t(out_c);
        if (capacity < need) Fail("pixel buffer too small");
        if (out_c == r.c) std::memcpy(pixels, r.px.data(), need);
        else Convert(r, out_c, pixels);
        return 0;
    } catch (const std::exception &e) {
        g_decode_error = e.what();
        return 1;
    }
}

const char *pth_decode_image_error(void) { return g_decode_error.c_str(); }

}  // extern "C"

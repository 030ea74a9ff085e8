// image_decode.cc — PNG and JPEG decoding to RGBA8 for glTF textures.
//
// Stands in for stbi_load_from_memory(..., 4) in the reference (src/cpu_framebuffer.cc:36-83).  PNG is lossless, so
// the result is the reference's: 16-bit samples keep their high byte, gray is replicated, tRNS becomes alpha.  JPEG
// decoding is defined only up to the inverse DCT and chroma upsampling a decoder chooses, so those texels can differ
// from stb's by a few units out of 255.  Interlaced PNG and progressive JPEG are read; arithmetic-coded, lossless and
// 12-bit JPEG and CMYK are rejected with a clear message (stb rejects the first three too).
#include "image_decode.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace app {
namespace {

[[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("Unable to load image from memory: ") + what); }

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
uint32_t be16(const uint8_t* p) { return (uint32_t(p[0]) << 8) | p[1]; }

// ------------------------------------------------------------------------------------------------ PNG
DecodedImage decode_png(const uint8_t* data, size_t size) {
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool end = false;
    while (!end && pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        const uint8_t* tag = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (pos + 12ull + len > size) bad("truncated PNG chunk");
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) bad("bad IHDR");
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(tag, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            end = true;
        }
        pos += 12ull + len;
    }
    if (!w || !h || w > (1u << 24) || h > (1u << 24) || static_cast<uint64_t>(w) * h > (1ull << 28)) bad("bad PNG size");
    if (interlace > 1) bad("bad PNG interlace method");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: bad("bad PNG colour type");
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) bad("bad PNG bit depth");
    if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) bad("bad PNG bit depth for its colour type");
    if (ctype == 3 && (depth == 16 || palette.empty())) bad("bad PNG palette");
    const size_t bits_per_pixel = static_cast<size_t>(channels) * depth;
    const size_t bpp = std::max<size_t>(1, bits_per_pixel / 8);  // filter distance
    // The picture is one pass over all pixels, or the seven Adam7 passes (each a reduced picture of every dx-th column
    // and dy-th row starting at (x0, y0)), stored one after the other in the data stream.
    struct Pass {
        uint32_t x0, y0, dx, dy;
    };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass* passes = interlace ? adam7 : whole;
    const int n_passes = interlace ? 7 : 1;
    auto pass_size = [&](const Pass& ps, uint32_t& pw, uint32_t& ph) -> size_t {
        pw = ps.x0 < w ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0;
        ph = ps.y0 < h ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        return (pw && ph) ? ((bits_per_pixel * pw + 7) / 8 + 1) * ph : 0;
    };
    size_t raw_size = 0;
    for (int k = 0; k < n_passes; ++k) {
        uint32_t pw, ph;
        raw_size += pass_size(passes[k], pw, ph);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf out_len = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &out_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || out_len != raw.size())
        bad("corrupt PNG data stream");
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = (ctype == 3) ? (trns.empty() ? 3 : 4) : channels;
    img.rgba.resize(static_cast<size_t>(w) * h * 4);
    const int scale_small[5] = {0, 255, 85, 0, 17};  // 1-, 2- and 4-bit gray samples stretched to 0..255
    size_t offset = 0;
    for (int k = 0; k < n_passes; ++k) {
        const Pass& ps = passes[k];
        uint32_t pw, ph;
        const size_t bytes = pass_size(ps, pw, ph);
        if (!bytes) continue;
        uint8_t* base = raw.data() + offset;
        offset += bytes;
        const size_t row_bytes = (bits_per_pixel * pw + 7) / 8;
        // undo the scanline filters in place
        std::vector<uint8_t> zero(row_bytes, 0);
        for (uint32_t y = 0; y < ph; ++y) {
            uint8_t* row = base + y * (row_bytes + 1);
            const int filter = row[0];
            uint8_t* cur = row + 1;
            const uint8_t* up = y ? base + (y - 1) * (row_bytes + 1) + 1 : zero.data();
            for (size_t i = 0; i < row_bytes; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
                int pred = 0;
                switch (filter) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: {
                        const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                        break;
                    }
                    default: bad("bad PNG filter");
                }
                cur[i] = static_cast<uint8_t>(cur[i] + pred);
            }
        }
        for (uint32_t py = 0; py < ph; ++py) {
            const uint8_t* row = base + py * (row_bytes + 1) + 1;
            for (uint32_t px = 0; px < pw; ++px) {
                uint32_t s[4] = {0, 0, 0, 0};  // raw samples of this pixel
                for (int c = 0; c < channels; ++c) {
                    const size_t idx = static_cast<size_t>(px) * channels + c;
                    if (depth == 8) s[c] = row[idx];
                    else if (depth == 16) s[c] = be16(row + idx * 2);
                    else {
                        const size_t bit = idx * depth;
                        s[c] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    }
                }
                const uint32_t x = ps.x0 + px * ps.dx, y = ps.y0 + py * ps.dy;
                uint8_t* o = img.rgba.data() + (static_cast<size_t>(y) * w + x) * 4;
                auto to8 = [&](uint32_t v) -> uint8_t {
                    if (depth == 16) return static_cast<uint8_t>(v >> 8);
                    if (depth < 8) return static_cast<uint8_t>(v * scale_small[depth]);
                    return static_cast<uint8_t>(v);
                };
                if (ctype == 3) {
                    if (s[0] * 3 + 2 >= palette.size()) bad("PNG palette index out of range");
                    o[0] = palette[s[0] * 3]; o[1] = palette[s[0] * 3 + 1]; o[2] = palette[s[0] * 3 + 2];
                    o[3] = s[0] < trns.size() ? trns[s[0]] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    o[0] = o[1] = o[2] = to8(s[0]);
                    o[3] = (ctype == 4) ? to8(s[1]) : 255;
                    if (ctype == 0 && trns.size() >= 2 && s[0] == be16(trns.data())) o[3] = 0;
                } else {
                    o[0] = to8(s[0]); o[1] = to8(s[1]); o[2] = to8(s[2]);
                    o[3] = (ctype == 6) ? to8(s[3]) : 255;
                    if (ctype == 2 && trns.size() >= 6 && s[0] == be16(trns.data()) && s[1] == be16(trns.data() + 2) &&
                        s[2] == be16(trns.data() + 4))
                        o[3] = 0;
                }
            }
        }
    }
    return img;
}

// ------------------------------------------------------------------------------------------------ JPEG
struct Huffman {
    uint8_t size[257];
    uint16_t code[256];
    uint8_t values[256];
    int count = 0;
    int maxcode[18];
    int delta[17];
};
void build_huffman(Huffman& h, const uint8_t counts[16], const uint8_t* values, int n_values) {
    // the 16 counts are file data and can declare up to 16 * 255 symbols: reject the table before anything is filled
    int total = 0;
    for (int len = 0; len < 16; ++len) total += counts[len];
    if (total > 256 || total != n_values) bad("bad JPEG Huffman table");
    int k = 0;
    for (int len = 1; len <= 16; ++len)
        for (int i = 0; i < counts[len - 1] && k < 256; ++i) h.size[k++] = static_cast<uint8_t>(len);
    h.size[k] = 0;
    h.count = k;
    int code = 0;
    k = 0;
    for (int len = 1; len <= 16; ++len) {
        h.delta[len] = k - code;
        while (h.size[k] == len) h.code[k++] = static_cast<uint16_t>(code++);
        if (code > (1 << len)) bad("bad JPEG Huffman table");  // more codes of this length than the length has
        h.maxcode[len] = code;  // first code of this length that is NOT in the table
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    std::memcpy(h.values, values, static_cast<size_t>(n_values));
}

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    void fill() {
        while (bits <= 24) {
            uint32_t b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xff) {
                    const uint8_t n = p < end ? *p : 0;
                    if (n == 0) ++p;                      // stuffed zero
                    else { hit_marker = true; --p; b = 0; }  // a marker: feed zeros from here on
                }
            }
            acc |= b << (24 - bits);
            bits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = static_cast<int>(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    int decode(const Huffman& h) {
        if (bits < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = static_cast<int>(acc >> (32 - len));
            if (code < h.maxcode[len]) {
                acc <<= len;
                bits -= len;
                const int idx = code + h.delta[len];
                if (idx < 0 || idx >= h.count) bad("bad JPEG Huffman code");
                return h.values[idx];
            }
        }
        bad("bad JPEG Huffman code");
    }
    void reset() { acc = 0; bits = 0; hit_marker = false; }
};
int extend_sign(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float in[64], uint8_t* out, size_t stride) {
    static float basis[8][8];
    static bool ready = false;
    if (!ready) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u)
                basis[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
        ready = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            float s = 0;
            for (int u = 0; u < 8; ++u) s += basis[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            float s = 0;
            for (int v = 0; v < 8; ++v) s += basis[y][v] * tmp[v * 8 + x];
            const int q = static_cast<int>(std::lround(s + 128.0f));
            out[y * stride + x] = static_cast<uint8_t>(std::clamp(q, 0, 255));
        }
}

// Baseline, extended-sequential and progressive Huffman JPEG (what stb_image, the reference's decoder, reads): every scan
// adds to per-block coefficient arrays — whole blocks for sequential files; DC / AC bands, first passes and refinement
// passes (successive approximation) for progressive ones; interleaved or one component per scan — and the picture is
// reconstructed once at the end.
DecodedImage decode_jpeg(const uint8_t* data, size_t size) {
    struct Component {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
        int bw = 0, bh = 0;        // plane size in samples (whole MCUs)
        int blocks_w = 0, blocks_h = 0;  // blocks that hold picture data (scans of this component alone cover these)
        std::vector<int16_t> coef;  // bw/8 * bh/8 blocks of 64, natural (de-zigzagged) order
        std::vector<uint8_t> plane;
        int pred = 0;
    };
    uint16_t quant[4][64] = {};
    Huffman dc[4], ac[4];
    bool have_dc[4] = {}, have_ac[4] = {};
    std::vector<Component> comps;
    int width = 0, height = 0, restart = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool adobe_rgb = false, progressive = false, any_scan = false;
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (data[pos] != 0xff) { ++pos; continue; }
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { ++pos; continue; }
        pos += 2;
        if (m == 0x00 || m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
        if (m == 0xd9) break;
        if (pos + 2 > size) bad("truncated JPEG");
        const size_t len = be16(data + pos);
        if (len < 2 || pos + len > size) bad("truncated JPEG segment");
        const uint8_t* s = data + pos + 2;
        const size_t n = len - 2;
        if (m == 0xdb) {  // quantisation tables
            size_t i = 0;
            while (i < n) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) bad("bad JPEG quantisation table");
                for (int k = 0; k < 64; ++k) {
                    quant[tq][kZigzag[k]] = pq ? static_cast<uint16_t>(be16(s + i)) : s[i];
                    i += pq ? 2 : 1;
                }
            }
        } else if (m == 0xc4) {  // Huffman tables
            size_t i = 0;
            while (i + 17 <= n) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) bad("bad JPEG Huffman table id");
                int total = 0;
                for (int k = 0; k < 16; ++k) total += s[i + 1 + k];
                if (i + 17 + total > n) bad("bad JPEG Huffman table");
                build_huffman(tc ? ac[th] : dc[th], s + i + 1, s + i + 17, total);
                (tc ? have_ac : have_dc)[th] = true;
                i += 17 + static_cast<size_t>(total);
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // baseline / extended sequential / progressive frame
            if (!comps.empty()) bad("second JPEG frame header");
            if (n < 6 || s[0] != 8) bad("only 8-bit JPEG is supported");
            progressive = (m == 0xc2);
            height = static_cast<int>(be16(s + 1));
            width = static_cast<int>(be16(s + 3));
            const int nc = s[5];
            if ((nc != 1 && nc != 3) || n < 6 + 3u * nc) bad("unsupported JPEG component count");
            if (width <= 0 || height <= 0 || static_cast<uint64_t>(width) * height > (1ull << 28)) bad("bad JPEG size");
            comps.resize(nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = s[6 + 3 * c];
                comps[c].h = s[7 + 3 * c] >> 4;
                comps[c].v = s[7 + 3 * c] & 15;
                comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) bad("bad JPEG frame header");
                hmax = std::max(hmax, comps[c].h);
                vmax = std::max(vmax, comps[c].v);
            }
            if (nc == 1) comps[0].h = comps[0].v = hmax = vmax = 1;  // a single component is never interleaved
            mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
            mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
            for (Component& c : comps) {
                c.bw = mcus_x * c.h * 8;
                c.bh = mcus_y * c.v * 8;
                c.blocks_w = ((width * c.h + hmax - 1) / hmax + 7) / 8;
                c.blocks_h = ((height * c.v + vmax - 1) / vmax + 7) / 8;
                c.coef.assign(static_cast<size_t>(c.bw) * c.bh, 0);
            }
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            bad("lossless, hierarchical and arithmetic-coded JPEG are not supported");
        } else if (m == 0xdd) {
            if (n >= 2) restart = static_cast<int>(be16(s));
        } else if (m == 0xee) {
            if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_rgb = (s[11] == 0);
        } else if (m == 0xda) {  // one scan
            if (comps.empty()) bad("JPEG scan before frame header");
            const int ns = n ? s[0] : 0;
            if (ns < 1 || ns > static_cast<int>(comps.size()) || n < 1 + 2u * ns + 3) bad("bad JPEG scan header");
            Component* in_scan[3] = {nullptr, nullptr, nullptr};
            for (int k = 0; k < ns; ++k) {
                const int id = s[1 + 2 * k];
                Component* c = nullptr;
                for (Component& cc : comps) if (cc.id == id) c = &cc;
                if (!c) bad("bad JPEG scan header");
                c->td = s[2 + 2 * k] >> 4;
                c->ta = s[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3) bad("bad JPEG scan header");
                in_scan[k] = c;
            }
            const int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
            if (progressive) {
                if (ss > se || se > 63 || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) bad("bad JPEG progressive scan");
            } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
                bad("bad JPEG sequential scan");
            }
            for (int k = 0; k < ns; ++k) {
                const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
                if ((need_dc && !have_dc[in_scan[k]->td]) || (need_ac && !have_ac[in_scan[k]->ta])) bad("JPEG scan names a missing Huffman table");
                in_scan[k]->pred = 0;
            }
            any_scan = true;
            BitReader br{data + pos + len, data + size};
            int eobrun = 0;
            auto block_at = [](Component& c, int bx, int by) { return c.coef.data() + (static_cast<size_t>(by) * (c.bw / 8) + bx) * 64; };
            auto decode_block = [&](Component& c, int16_t* q) {
                if (!progressive) {
                    const int t = br.decode(dc[c.td]);
                    if (t > 11) bad("bad JPEG DC size");
                    c.pred += extend_sign(br.get(t), t);
                    q[0] = static_cast<int16_t>(c.pred);
                    for (int k = 1; k < 64;) {
                        const int rs = br.decode(ac[c.ta]);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r != 15) break;  // end of block
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (k > 63) bad("bad JPEG AC run");
                        q[kZigzag[k]] = static_cast<int16_t>(extend_sign(br.get(sz), sz));
                        ++k;
                    }
                } else if (ss == 0) {
                    if (ah == 0) {  // DC, first pass: the value without its low `al` bits
                        const int t = br.decode(dc[c.td]);
                        if (t > 11) bad("bad JPEG DC size");
                        c.pred += extend_sign(br.get(t), t);
                        q[0] = static_cast<int16_t>(c.pred * (1 << al));
                    } else if (br.get(1)) {  // DC, refinement: one more bit
                        q[0] = static_cast<int16_t>(q[0] | (1 << al));
                    }
                } else if (ah == 0) {  // AC band, first pass
                    if (eobrun > 0) { --eobrun; return; }
                    for (int k = ss; k <= se;) {
                        const int rs = br.decode(ac[c.ta]);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r < 15) {  // end of band for this and the next eobrun blocks
                                eobrun = (1 << r) - 1;
                                if (r) eobrun += br.get(r);
                                break;
                            }
                            k += 16;
                            continue;
                        }
                        k += r;
                        if (k > se) bad("bad JPEG AC run");
                        q[kZigzag[k]] = static_cast<int16_t>(extend_sign(br.get(sz), sz) * (1 << al));
                        ++k;
                    }
                } else {  // AC band, refinement: correction bits for coefficients already non-zero, new ones of magnitude 1 << al
                    const int p1 = 1 << al, m1 = -(1 << al);
                    auto refine = [&](int16_t& v) {
                        if (br.get(1) && (v & p1) == 0) v = static_cast<int16_t>(v + (v >= 0 ? p1 : m1));
                    };
                    int k = ss;
                    if (eobrun == 0) {
                        for (; k <= se; ++k) {
                            const int rs = br.decode(ac[c.ta]);
                            int r = rs >> 4;
                            const int sz = rs & 15;
                            int value = 0;
                            if (sz == 0) {
                                if (r < 15) {
                                    eobrun = 1 << r;
                                    if (r) eobrun += br.get(r);
                                    break;
                                }
                            } else {
                                if (sz != 1) bad("bad JPEG refinement code");
                                value = br.get(1) ? p1 : m1;
                            }
                            for (; k <= se; ++k) {  // pass r still-zero coefficients, refining the non-zero ones on the way
                                int16_t& v = q[kZigzag[k]];
                                if (v != 0) {
                                    refine(v);
                                } else {
                                    if (r == 0) {
                                        if (value) v = static_cast<int16_t>(value);
                                        break;
                                    }
                                    --r;
                                }
                            }
                        }
                    }
                    if (eobrun > 0) {
                        for (; k <= se; ++k) {
                            int16_t& v = q[kZigzag[k]];
                            if (v != 0) refine(v);
                        }
                        --eobrun;
                    }
                }
            };
            int until_restart = restart;
            auto maybe_restart = [&]() {
                if (!restart) return;
                if (until_restart == 0) {  // byte-align, skip the RSTn marker, reset the predictors and the band run
                    br.reset();
                    const uint8_t* q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xff && q[1] >= 0xd0 && q[1] <= 0xd7)) ++q;
                    br.p = (q + 2 <= br.end) ? q + 2 : br.end;
                    for (Component& c : comps) c.pred = 0;
                    eobrun = 0;
                    until_restart = restart;
                }
                --until_restart;
            };
            if (ns == 1) {  // one component: its blocks row by row, only those that hold picture data
                Component& c = *in_scan[0];
                for (int by = 0; by < c.blocks_h; ++by)
                    for (int bx = 0; bx < c.blocks_w; ++bx) {
                        maybe_restart();
                        decode_block(c, block_at(c, bx, by));
                    }
            } else {
                for (int my = 0; my < mcus_y; ++my)
                    for (int mx = 0; mx < mcus_x; ++mx) {
                        maybe_restart();
                        for (int k = 0; k < ns; ++k) {
                            Component& c = *in_scan[k];
                            for (int by = 0; by < c.v; ++by)
                                for (int bx = 0; bx < c.h; ++bx) decode_block(c, block_at(c, mx * c.h + bx, my * c.v + by));
                        }
                    }
            }
            pos = static_cast<size_t>(br.p - data);  // at the marker that ended the scan (or wherever the data ran out)
            continue;
        }
        pos += len;
    }
    if (!any_scan) bad("JPEG without image data");
    // coefficients -> samples
    float block[64];
    for (Component& c : comps) {
        c.plane.assign(static_cast<size_t>(c.bw) * c.bh, 0);
        for (int by = 0; by < c.bh / 8; ++by)
            for (int bx = 0; bx < c.bw / 8; ++bx) {
                const int16_t* q = c.coef.data() + (static_cast<size_t>(by) * (c.bw / 8) + bx) * 64;
                for (int k = 0; k < 64; ++k) block[k] = static_cast<float>(q[k] * static_cast<int>(quant[c.tq][k]));
                idct8x8(block, c.plane.data() + static_cast<size_t>(by * 8) * c.bw + bx * 8, static_cast<size_t>(c.bw));
            }
    }
    // planes -> RGBA: chroma by bilinear interpolation between sample centres
    DecodedImage img;
    img.width = width;
    img.height = height;
    img.channels = static_cast<int>(comps.size());
    img.rgba.resize(static_cast<size_t>(width) * height * 4);
    auto sample = [&](const Component& c, int x, int y) -> float {
        if (c.h == hmax && c.v == vmax) return c.plane[static_cast<size_t>(y) * c.bw + x];
        const float fx = (x + 0.5f) * c.h / hmax - 0.5f, fy = (y + 0.5f) * c.v / vmax - 0.5f;
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
        const int x0 = static_cast<int>(std::floor(fx)), y0 = static_cast<int>(std::floor(fy));
        const float tx = fx - x0, ty = fy - y0;
        auto at = [&](int xx, int yy) {
            xx = std::clamp(xx, 0, cw - 1);
            yy = std::clamp(yy, 0, ch - 1);
            return static_cast<float>(c.plane[static_cast<size_t>(yy) * c.bw + xx]);
        };
        const float top = at(x0, y0) * (1 - tx) + at(x0 + 1, y0) * tx, bot = at(x0, y0 + 1) * (1 - tx) + at(x0 + 1, y0 + 1) * tx;
        return top * (1 - ty) + bot * ty;
    };
    auto to8 = [](float v) { return static_cast<uint8_t>(std::clamp(static_cast<int>(std::lround(v)), 0, 255)); };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            uint8_t* o = img.rgba.data() + (static_cast<size_t>(y) * width + x) * 4;
            if (comps.size() == 1) {
                o[0] = o[1] = o[2] = to8(sample(comps[0], x, y));
            } else if (adobe_rgb) {
                o[0] = to8(sample(comps[0], x, y)); o[1] = to8(sample(comps[1], x, y)); o[2] = to8(sample(comps[2], x, y));
            } else {
                const float Y = sample(comps[0], x, y), cb = sample(comps[1], x, y) - 128.0f, cr = sample(comps[2], x, y) - 128.0f;
                o[0] = to8(Y + 1.402f * cr);
                o[1] = to8(Y - 0.344136f * cb - 0.714136f * cr);
                o[2] = to8(Y + 1.772f * cb);
            }
            o[3] = 255;
        }
    return img;
}

}  // namespace

DecodedImage decode_image(const uint8_t* data, size_t size) {
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (size >= 8 && !std::memcmp(data, png_sig, 8)) return decode_png(data, size);
    if (size >= 3 && data[0] == 0xff && data[1] == 0xd8) return decode_jpeg(data, size);
    bad("neither PNG nor JPEG");
}

}  // namespace app

// jpeg_decode.cpp -- JPEG (ITU T.81) decoder for map_* textures.
//
// The reference decodes textures with the `image` crate (reference src/texture.rs:18, `image = "0.25.9"` in Cargo.toml),
// which is not vendored.  T.81 fixes the entropy decoding (coefficients are exact) but not the arithmetic after it, so a
// decoder has to pick an inverse DCT, a chroma upsampler and a colour conversion.  This one uses the Independent JPEG
// Group's defaults, restated from their published description: the 13-bit fixed-point Loeffler-Ligtenberg-Moschytz
// inverse DCT ("slow integer"), triangle-filter ("fancy") upsampling for 2:1 horizontal / 2:1 vertical / 2x2 chroma,
// replication for other integral ratios, and the 16-bit fixed-point YCbCr->RGB tables.  tests/test_obj_loader.py pins it
// bit-for-bit against Pillow's libjpeg on generated and real files; whether the `image` crate's decoder rounds every sample
// the same way is NOT pinned (no Rust toolchain here): a JPEG texture may differ from the reference's by an LSB.
//
// Supported: baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman, 8-bit, 1 or 3 components, any
// integral sampling ratio, restart intervals, 8/16-bit quantisation tables, JFIF / Adobe colour-transform markers.
// Not supported (reported, texture skipped): arithmetic coding, lossless, 12-bit, 4-component (CMYK / YCCK) files.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace mipt_jpeg {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t look_len[256];          // 8-bit lookahead: code length (0 = longer than 8 bits)
    uint8_t look_sym[256];
    int32_t maxcode[18];            // per length; -1 if none
    int32_t valoff[18];
    uint8_t vals[256];
    bool build(const uint8_t counts[16], const uint8_t *symbols, int n_symbols) {
        memcpy(vals, symbols, (size_t)n_symbols);
        memset(look_len, 0, sizeof look_len);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valoff[len] = k - code;
            const int n = counts[len - 1];
            if (code + n > (1 << len)) return false;
            for (int i = 0; i < n; i++, k++, code++) {
                if (len <= 8) {
                    const int first = code << (8 - len);
                    for (int f = 0; f < (1 << (8 - len)); f++) { look_len[first + f] = (uint8_t)len; look_sym[first + f] = vals[k]; }
                }
            }
            maxcode[len] = n ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return k == n_symbols;
    }
};

struct Bits {                       // entropy-coded segment reader: removes FF00 stuffing, stops (feeding zeros) at a marker
    const uint8_t *p = nullptr, *end = nullptr;
    uint64_t acc = 0;
    int n = 0;
    bool at_marker = false;
    void fill() {
        while (n <= 56) {
            int b = 0;
            if (!at_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    const int b2 = (p + 1 < end) ? p[1] : 0xD9;
                    if (b2 == 0) p += 2;
                    else if (b2 == 0xFF) { p++; continue; }          // fill byte
                    else { at_marker = true; b = 0; }
                } else {
                    p++;
                }
            }
            acc |= (uint64_t)b << (56 - n);
            n += 8;
        }
    }
    inline int peek8() { if (n < 8) fill(); return (int)(acc >> 56); }
    inline void skip(int k) { acc <<= k; n -= k; }
    inline int get(int k) {
        if (k == 0) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (64 - k));
        acc <<= k; n -= k;
        return v;
    }
    inline int bit() { return get(1); }
    void restart() {                // byte-align, step over the RSTn marker
        acc = 0; n = 0;
        while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
            if (p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF) { at_marker = true; return; }   // some other marker: leave it
            p++;
        }
        if (p + 1 < end) p += 2;
        at_marker = false;
    }
};

inline int decode_sym(Bits &b, const Huff &h) {
    const int look = b.peek8();
    const int len = h.look_len[look];
    if (len) { b.skip(len); return h.look_sym[look]; }
    if (b.n < 16) b.fill();
    int code = (int)(b.acc >> 55);  // 9 bits
    int l = 9;
    while (l <= 16 && code > h.maxcode[l]) { l++; code = (int)(b.acc >> (64 - l)); }
    if (l > 16) { b.skip(16); return 0; }                 // corrupt code: libjpeg substitutes 0
    b.skip(l);
    return h.vals[(code + h.valoff[l]) & 255];
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v; }

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                 // current scan's tables
    int blocks_w = 0, blocks_h = 0;     // padded to whole MCUs
    int ds_w = 0, ds_h = 0;             // downsampled_width / _height: the real samples
    int pred = 0;
    std::vector<int16_t> coef;
    std::vector<uint8_t> plane;         // blocks_h*8 rows of blocks_w*8 samples
};

// 8x8 inverse DCT, "slow integer" (LLM with 13-bit constants, 2 extra bits kept between the passes).
void idct_islow(const int16_t *in, const uint16_t *q, uint8_t *out, int stride) {
    const int CB = 13, P1 = 2;
    const int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                  F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
    int64_t ws[64];
    for (int c = 0; c < 8; c++) {
        const int16_t *i = in + c;
        const uint16_t *qq = q + c;
        int64_t z2 = (int64_t)i[16] * qq[16], z3 = (int64_t)i[48] * qq[48];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = (int64_t)i[0] * qq[0]; z3 = (int64_t)i[32] * qq[32];
        int64_t tmp0 = (z2 + z3) * ((int64_t)1 << CB), tmp1 = (z2 - z3) * ((int64_t)1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (int64_t)i[56] * qq[56]; tmp1 = (int64_t)i[40] * qq[40]; tmp2 = (int64_t)i[24] * qq[24]; tmp3 = (int64_t)i[8] * qq[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c + 0] = descale(tmp10 + tmp3, CB - P1);  ws[c + 56] = descale(tmp10 - tmp3, CB - P1);
        ws[c + 8] = descale(tmp11 + tmp2, CB - P1);  ws[c + 48] = descale(tmp11 - tmp2, CB - P1);
        ws[c + 16] = descale(tmp12 + tmp1, CB - P1); ws[c + 40] = descale(tmp12 - tmp1, CB - P1);
        ws[c + 24] = descale(tmp13 + tmp0, CB - P1); ws[c + 32] = descale(tmp13 - tmp0, CB - P1);
    }
    auto limit = [](int64_t x) -> uint8_t {   // the 1024-entry range table, centred: clamp(x + 128) for sane data
        const int v = (int)(x & 1023);
        return (uint8_t)(v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896);
    };
    for (int r = 0; r < 8; r++) {
        const int64_t *w = ws + r * 8;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        int64_t tmp0 = (w[0] + w[4]) * ((int64_t)1 << CB), tmp1 = (w[0] - w[4]) * ((int64_t)1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t *o = out + (size_t)r * stride;
        const int S = CB + P1 + 3;
        o[0] = limit(descale(tmp10 + tmp3, S)); o[7] = limit(descale(tmp10 - tmp3, S));
        o[1] = limit(descale(tmp11 + tmp2, S)); o[6] = limit(descale(tmp11 - tmp2, S));
        o[2] = limit(descale(tmp12 + tmp1, S)); o[5] = limit(descale(tmp12 - tmp1, S));
        o[3] = limit(descale(tmp13 + tmp0, S)); o[4] = limit(descale(tmp13 - tmp0, S));
    }
}

struct Decoder {
    std::vector<uint8_t> file;
    std::string err;
    int width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    int restart_interval = 0;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;

    bool fail(const std::string &m) { err = m; return false; }

    bool parse_sof(const uint8_t *s, int len) {
        if (len < 6) return fail("short SOF");
        if (s[0] != 8) return fail("only 8-bit JPEG is supported");
        height = s[1] << 8 | s[2]; width = s[3] << 8 | s[4];
        const int nc = s[5];
        if (width == 0 || height == 0) return fail("empty image (DNL height is not supported)");
        if ((uint64_t)width * height > (1ull << 27)) return fail("image larger than 2^27 pixels");
        if (nc != 1 && nc != 3) return fail("only 1- or 3-component JPEG is supported (CMYK / YCCK is not)");
        if (len < 6 + nc * 3) return fail("short SOF");
        comps.resize((size_t)nc);
        for (int i = 0; i < nc; i++) {
            Comp &c = comps[(size_t)i];
            c.id = s[6 + i * 3]; c.h = s[7 + i * 3] >> 4; c.v = s[7 + i * 3] & 15; c.tq = s[8 + i * 3] & 3;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("bad sampling factor");
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }      // a single component is never subsampled
        for (Comp &c : comps)
            if (hmax % c.h || vmax % c.v) return fail("fractional chroma sampling ratios are not supported");
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax); mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (Comp &c : comps) {
            c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
            c.ds_w = (width * c.h + hmax - 1) / hmax; c.ds_h = (height * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        return true;
    }

    bool parse_dqt(const uint8_t *s, int len) {
        while (len > 0) {
            const int pq = s[0] >> 4, t = s[0] & 15;
            if (t > 3 || pq > 1) return fail("bad DQT");
            const int need = 1 + 64 * (pq + 1);
            if (len < need) return fail("short DQT");
            for (int i = 0; i < 64; i++) qt[t][kZigzag[i]] = pq ? (uint16_t)(s[1 + 2 * i] << 8 | s[2 + 2 * i]) : s[1 + i];
            qt_present[t] = true;
            s += need; len -= need;
        }
        return true;
    }

    bool parse_dht(const uint8_t *s, int len) {
        while (len > 0) {
            if (len < 17) return fail("short DHT");
            const int tc = s[0] >> 4, th = s[0] & 15;
            if (tc > 1 || th > 3) return fail("bad DHT");
            int n = 0;
            for (int i = 0; i < 16; i++) n += s[1 + i];
            if (n > 256 || len < 17 + n) return fail("bad DHT");
            if (!(tc ? ac[th] : dc[th]).build(s + 1, s + 17, n)) return fail("bad Huffman table");
            s += 17 + n; len -= 17 + n;
        }
        return true;
    }

    // one block of one scan; returns false on a structural error
    void block_baseline(Bits &b, Comp &c, int16_t *blk) {
        const int t = decode_sym(b, dc[c.td]) & 15;          // categories above 11 only occur in corrupt data
        if (t) c.pred += extend(b.get(t), t);
        blk[0] = (int16_t)c.pred;
        const Huff &h = ac[c.ta];
        for (int k = 1; k < 64; k++) {
            const int rs = decode_sym(b, h), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 15;
            } else {
                k += r;
                if (k > 63) break;
                blk[kZigzag[k]] = (int16_t)extend(b.get(s), s);
            }
        }
    }
    void block_dc_first(Bits &b, Comp &c, int16_t *blk, int al) {
        const int t = decode_sym(b, dc[c.td]) & 15;
        if (t) c.pred += extend(b.get(t), t);
        blk[0] = (int16_t)((int64_t)c.pred * (1 << al));
    }
    void block_ac_first(Bits &b, Comp &c, int16_t *blk, int ss, int se, int al, int &eobrun) {
        if (eobrun > 0) { eobrun--; return; }
        const Huff &h = ac[c.ta];
        for (int k = ss; k <= se; k++) {
            const int rs = decode_sym(b, h), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += b.get(r);
                    break;
                }
                k += 15;
            } else {
                k += r;
                if (k > 63) break;
                blk[kZigzag[k]] = (int16_t)(extend(b.get(s), s) * (1 << al));
            }
        }
    }
    void block_ac_refine(Bits &b, Comp &c, int16_t *blk, int ss, int se, int al, int &eobrun) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            const Huff &h = ac[c.ta];
            for (; k <= se; k++) {
                const int rs = decode_sym(b, h);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = b.bit() ? p1 : m1;                    // the new coefficient's value (size is always 1)
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += b.get(r);
                    break;
                }
                // pass already-nonzero coefficients (each takes a correction bit) and skip r still-zero ones
                do {
                    int16_t *cp = blk + kZigzag[k];
                    if (*cp != 0) {
                        if (b.bit() && (*cp & p1) == 0) *cp = (int16_t)(*cp + (*cp >= 0 ? p1 : m1));
                    } else {
                        if (--r < 0) break;
                    }
                    k++;
                } while (k <= se);
                if (s && k <= 63) blk[kZigzag[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t *cp = blk + kZigzag[k];
                if (*cp != 0 && b.bit() && (*cp & p1) == 0) *cp = (int16_t)(*cp + (*cp >= 0 ? p1 : m1));
            }
            eobrun--;
        }
    }

    bool decode_scan(const uint8_t *hdr, int len, const uint8_t *data, const uint8_t *end, const uint8_t **next) {
        if (comps.empty()) return fail("SOS before SOF");
        if (len < 1) return fail("short SOS");
        const int ns = hdr[0];
        if (ns < 1 || ns > (int)comps.size() || len < 1 + 2 * ns + 3) return fail("bad SOS");
        Comp *sc[4];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (Comp &c : comps)
                if (c.id == hdr[1 + 2 * i]) sc[i] = &c;
            if (!sc[i]) return fail("SOS names an unknown component");
            sc[i]->td = hdr[2 + 2 * i] >> 4; sc[i]->ta = hdr[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        const int ss = hdr[1 + 2 * ns], se = hdr[2 + 2 * ns], ah = hdr[3 + 2 * ns] >> 4, al = hdr[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss != 0 && ns != 1) || al > 13) return fail("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            return fail("bad sequential scan parameters");
        }
        for (int i = 0; i < ns; i++) {
            const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
            if (need_dc && !dc[sc[i]->td].present) return fail("scan uses an undefined DC table");
            if (need_ac && !ac[sc[i]->ta].present) return fail("scan uses an undefined AC table");
        }
        Bits b;
        b.p = data; b.end = end;
        int eobrun = 0;
        for (Comp &c : comps) c.pred = 0;
        auto do_block = [&](Comp &c, int bx, int by) {
            int16_t *blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            if (!progressive) block_baseline(b, c, blk);
            else if (ss == 0) { if (ah == 0) block_dc_first(b, c, blk, al); else if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << al)); }
            else if (ah == 0) block_ac_first(b, c, blk, ss, se, al, eobrun);
            else block_ac_refine(b, c, blk, ss, se, al, eobrun);
        };
        int todo = restart_interval;
        auto maybe_restart = [&]() {
            if (restart_interval && --todo == 0) {
                b.restart();
                for (Comp &c : comps) c.pred = 0;
                eobrun = 0;
                todo = restart_interval;
                return true;
            }
            return false;
        };
        if (ns == 1) {                                       // non-interleaved: the component's own block grid
            Comp &c = *sc[0];
            const int bw = (c.ds_w + 7) / 8, bh = (c.ds_h + 7) / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    do_block(c, bx, by);
                    if (!(by == bh - 1 && bx == bw - 1)) maybe_restart();
                }
        } else {
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    for (int i = 0; i < ns; i++)
                        for (int v = 0; v < sc[i]->v; v++)
                            for (int h = 0; h < sc[i]->h; h++) do_block(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v);
                    if (!(my == mcus_y - 1 && mx == mcus_x - 1)) maybe_restart();
                }
        }
        // step to the next marker (the reader never runs past one)
        const uint8_t *p = b.p;
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        *next = p;
        return true;
    }

    // chroma upsampling of one component to hmax x vmax resolution: out has out_w = ds_w * (hmax / h) columns per row
    void upsample(const Comp &c, std::vector<uint8_t> &out, int &out_w) {
        const int hr = hmax / c.h, vr = vmax / c.v, sw = c.blocks_w * 8;
        out_w = c.ds_w * hr;
        const int out_h = c.ds_h * vr;
        out.assign((size_t)out_w * out_h, 0);
        auto row = [&](int r) { if (r < 0) r = 0; if (r >= c.ds_h) r = c.ds_h - 1; return c.plane.data() + (size_t)r * sw; };
        const int n = c.ds_w;
        if (hr == 1 && vr == 1) {
            for (int y = 0; y < out_h; y++) memcpy(&out[(size_t)y * out_w], row(y), (size_t)out_w);
        } else if (hr == 2 && vr == 1 && n > 2) {            // triangle filter along the row
            for (int y = 0; y < out_h; y++) {
                const uint8_t *in = row(y);
                uint8_t *o = &out[(size_t)y * out_w];
                o[0] = in[0]; o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < n - 1; i++) {
                    o[2 * i] = (uint8_t)((in[i] * 3 + in[i - 1] + 1) >> 2);
                    o[2 * i + 1] = (uint8_t)((in[i] * 3 + in[i + 1] + 2) >> 2);
                }
                o[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2); o[2 * n - 1] = in[n - 1];
            }
        } else if (hr == 2 && vr == 2 && n > 2) {            // triangle filter in both directions (9/16, 3/16, 3/16, 1/16)
            for (int r = 0; r < c.ds_h; r++)
                for (int v = 0; v < 2; v++) {
                    const uint8_t *in0 = row(r), *in1 = row(v == 0 ? r - 1 : r + 1);
                    uint8_t *o = &out[(size_t)(2 * r + v) * out_w];
                    int last, cur = in0[0] * 3 + in1[0], nxt = in0[1] * 3 + in1[1];
                    o[0] = (uint8_t)((cur * 4 + 8) >> 4); o[1] = (uint8_t)((cur * 3 + nxt + 7) >> 4);
                    last = cur; cur = nxt;
                    for (int i = 1; i < n - 1; i++) {
                        nxt = in0[i + 1] * 3 + in1[i + 1];
                        o[2 * i] = (uint8_t)((cur * 3 + last + 8) >> 4);
                        o[2 * i + 1] = (uint8_t)((cur * 3 + nxt + 7) >> 4);
                        last = cur; cur = nxt;
                    }
                    o[2 * n - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); o[2 * n - 1] = (uint8_t)((cur * 4 + 7) >> 4);
                }
        } else if (hr == 1 && vr == 2) {                     // triangle filter down the column
            for (int r = 0; r < c.ds_h; r++)
                for (int v = 0; v < 2; v++) {
                    const uint8_t *in0 = row(r), *in1 = row(v == 0 ? r - 1 : r + 1);
                    uint8_t *o = &out[(size_t)(2 * r + v) * out_w];
                    for (int i = 0; i < n; i++) o[i] = (uint8_t)((in0[i] * 3 + in1[i] + (v == 0 ? 1 : 2)) >> 2);
                }
        } else {                                             // replication
            for (int y = 0; y < out_h; y++) {
                const uint8_t *in = row(y / vr);
                uint8_t *o = &out[(size_t)y * out_w];
                for (int x = 0; x < out_w; x++) o[x] = in[x / hr];
            }
        }
    }

    bool run(uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgba) {
        const uint8_t *p = file.data(), *end = p + file.size();
        if (file.size() < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG file (no SOI)");
        p += 2;
        bool eoi = false, any_scan = false;
        while (!eoi && p + 1 < end) {
            if (p[0] != 0xFF) { p++; continue; }
            const int m = p[1];
            if (m == 0xFF) { p++; continue; }
            p += 2;
            if (m == 0xD9) { eoi = true; break; }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) continue;   // parameterless
            if (p + 2 > end) break;
            const int len = (p[0] << 8 | p[1]) - 2;
            if (len < 0 || p + 2 + len > end) return fail("truncated marker segment");
            const uint8_t *s = p + 2;
            p += 2 + len;
            switch (m) {
            case 0xC0: case 0xC1: case 0xC2:
                if (!comps.empty()) return fail("more than one frame");
                progressive = (m == 0xC2);
                if (!parse_sof(s, len)) return false;
                break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
            case 0xC4: if (!parse_dht(s, len)) return false; break;
            case 0xDB: if (!parse_dqt(s, len)) return false; break;
            case 0xDD: if (len < 2) return fail("short DRI"); restart_interval = s[0] << 8 | s[1]; break;
            case 0xE0: if (len >= 5 && memcmp(s, "JFIF\0", 5) == 0) saw_jfif = true; break;
            case 0xEE: if (len >= 12 && memcmp(s, "Adobe", 5) == 0) { saw_adobe = true; adobe_transform = s[11]; } break;
            case 0xDA: {
                const uint8_t *next = nullptr;
                if (!decode_scan(s, len, p, end, &next)) return false;
                p = next;
                any_scan = true;
                break;
            }
            default: break;                                   // APPn, COM, DNL ...: skipped
            }
        }
        if (comps.empty() || !any_scan) return fail("no image data");
        // dequantise + inverse DCT
        for (Comp &c : comps) {
            if (!qt_present[c.tq]) return fail("component uses an undefined quantisation table");
            const int sw = c.blocks_w * 8;
            c.plane.assign((size_t)sw * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++)
                    idct_islow(&c.coef[((size_t)by * c.blocks_w + bx) * 64], qt[c.tq], &c.plane[(size_t)by * 8 * sw + (size_t)bx * 8], sw);
            std::vector<int16_t>().swap(c.coef);
        }
        rgba->assign((size_t)width * height * 4, 255);
        if (comps.size() == 1) {
            const Comp &c = comps[0];
            const int sw = c.blocks_w * 8;
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const uint8_t g = c.plane[(size_t)y * sw + x];
                    uint8_t *d = &(*rgba)[((size_t)y * width + x) * 4];
                    d[0] = d[1] = d[2] = g;
                }
        } else {
            std::vector<uint8_t> up[3];
            int uw[3];
            for (int i = 0; i < 3; i++) upsample(comps[(size_t)i], up[i], uw[i]);
            // colour space: JFIF => YCbCr; Adobe transform 0 => RGB; ids 'R','G','B' without either => RGB; otherwise YCbCr
            bool ycc = true;
            if (saw_jfif) ycc = true;
            else if (saw_adobe) ycc = adobe_transform != 0;
            else if (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
            int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
            for (int i = 0; i < 256; i++) {
                const int x = i - 128;
                cr_r[i] = (91881 * x + 32768) >> 16;          // FIX(1.40200)
                cb_b[i] = (116130 * x + 32768) >> 16;         // FIX(1.77200)
                cr_g[i] = -46802 * x;                         // FIX(0.71414)
                cb_g[i] = -22554 * x + 32768;                 // FIX(0.34414)
            }
            auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const int a = up[0][(size_t)y * uw[0] + x], bb = up[1][(size_t)y * uw[1] + x], cc = up[2][(size_t)y * uw[2] + x];
                    uint8_t *d = &(*rgba)[((size_t)y * width + x) * 4];
                    if (ycc) {
                        d[0] = clamp(a + cr_r[cc]);
                        d[1] = clamp(a + ((cb_g[bb] + cr_g[cc]) >> 16));
                        d[2] = clamp(a + cb_b[bb]);
                    } else {
                        d[0] = (uint8_t)a; d[1] = (uint8_t)bb; d[2] = (uint8_t)cc;
                    }
                }
        }
        *w_out = (uint32_t)width; *h_out = (uint32_t)height;
        return true;
    }
};

} // namespace

// top-down RGBA8 (alpha 255), the rows in file order
bool decode(const std::string &path, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgba, std::string *err) {
    Decoder d;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot open"; return false; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) { fclose(f); if (err) *err = "empty file"; return false; }
    d.file.resize((size_t)n);
    const size_t got = fread(d.file.data(), 1, (size_t)n, f);
    fclose(f);
    if (got != (size_t)n) { if (err) *err = "short read"; return false; }
    try {
        if (!d.run(w_out, h_out, rgba)) { if (err) *err = d.err; return false; }
    } catch (const std::exception &e) {            // allocation failure on a hostile header
        if (err) *err = std::string("decoder exception: ") + e.what();
        return false;
    }
    return true;
}

} // namespace mipt_jpeg

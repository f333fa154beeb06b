// png_decode.cpp -- minimal PNG reader for the OBJ/MTL loader's map_* textures.
// The reference decodes textures with the `image` crate (image::open(path).flipv().to_rgba8(), src/texture.rs:18),
// which is not vendored; this restates the published PNG/zlib formats (RFC 2083, RFC 1950/1951): non-interlaced,
// bit depth 8 (and 16, reduced to the high byte), colour types 0, 2, 3, 4, 6, all five scanline filters, tRNS for
// palettes.  Output is RGBA8, top row first (the caller applies flipv).  No reference fixture pins the decoder:
// "parity unpinned" for texel bytes; tests round-trip against Python's zlib.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mipt_png {

namespace {

struct BitReader {
    const uint8_t *p; size_t n, pos = 0; uint32_t bitbuf = 0; int bitcnt = 0; bool err = false;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len) {}
    uint32_t bits(int k) {
        while (bitcnt < k) {
            if (pos >= n) { err = true; return 0; }
            bitbuf |= (uint32_t)p[pos++] << bitcnt; bitcnt += 8;
        }
        uint32_t v = bitbuf & ((1u << k) - 1u);
        bitbuf >>= k; bitcnt -= k;
        return k ? v : 0;
    }
    void align() { bitbuf = 0; bitcnt = 0; }
};

struct Huff {                       // canonical Huffman decoding table (count/symbol form, RFC 1951 3.2.2)
    uint16_t count[16]; uint16_t symbol[320];
    bool build(const uint8_t *len, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.bits(1);
            if (br.err) return -1;
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out) {
    if (n < 6) return false;
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) return false;   // zlib header
    BitReader br(src + 2, n - 2);
    static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
    static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
    for (;;) {
        uint32_t final = br.bits(1), type = br.bits(2);
        if (br.err) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            uint32_t len = br.p[br.pos] | br.p[br.pos + 1] << 8, nlen = br.p[br.pos + 2] | br.p[br.pos + 3] << 8;
            br.pos += 4;
            if ((len ^ 0xffff) != nlen || br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lens[i] = 8;
                for (; i < 256; i++) lens[i] = 9;
                for (; i < 280; i++) lens[i] = 7;
                for (; i < 288; i++) lens[i] = 8;
                hl.build(lens, 288);
                for (i = 0; i < 30; i++) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huff hc;
                if (br.err || nlen > 286 || ndist > 30 || !hc.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = hc.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t val = 0;
                        if (sym == 16) { if (idx == 0) return false; val = lens[idx - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = val;
                    }
                }
                if (!hl.build(lens, nlen) || !hd.build(lens + nlen, ndist)) return false;
            }
            for (;;) {
                int sym = hl.decode(br);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    size_t len = lbase[sym] + br.bits(lext[sym]);
                    int ds = hd.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t dist = dbase[ds] + br.bits(dext[ds]);
                    if (br.err || dist > out.size()) return false;
                    size_t from = out.size() - dist;
                    for (size_t i = 0; i < len; i++) out.push_back(out[from + i]);
                }
            }
        } else return false;
        if (final) break;
    }
    return !br.err;
}

inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }
inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace

// Returns false (with *err) on anything unsupported or malformed.
bool decode(const std::string &path, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgba, std::string *err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { *err = "cannot open"; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (buf.size() < 8 + 25 || memcmp(buf.data(), sig, 8)) { *err = "not a PNG file"; return false; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const uint8_t *type = &buf[pos + 4], *data = &buf[pos + 8];
        if (pos + 12 + (size_t)len > buf.size()) { *err = "truncated chunk"; return false; }
        if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 65536 || h > 65536) { *err = "bad dimensions"; return false; }
    if (interlace) { *err = "interlaced PNG is not supported"; return false; }
    if (depth != 8 && depth != 16) { *err = "only bit depths 8 and 16 are supported"; return false; }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (ctype == 3 && depth != 8)) { *err = "unsupported colour type"; return false; }
    std::vector<uint8_t> raw;
    if (!inflate(idat.data(), idat.size(), raw)) { *err = "zlib stream is corrupt"; return false; }
    const size_t bpp = (size_t)channels * depth / 8, stride = (size_t)w * bpp;
    if (raw.size() < (stride + 1) * h) { *err = "image data too short"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgba->assign((size_t)w * h * 4, 255);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = &raw[(stride + 1) * y];
        const int ft = row[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = row[1 + i];
            int v;
            switch (ft) {
            case 0: v = x; break;
            case 1: v = x + a; break;
            case 2: v = x + b; break;
            case 3: v = x + ((a + b) >> 1); break;
            case 4: v = x + paeth(a, b, c); break;
            default: *err = "bad filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *s = &cur[(size_t)x * bpp];
            uint8_t *d = &(*rgba)[((size_t)y * w + x) * 4];
            const size_t st = depth / 8;                         // 16-bit samples: keep the high byte
            switch (ctype) {
            case 0: d[0] = d[1] = d[2] = s[0]; break;
            case 2: d[0] = s[0]; d[1] = s[st]; d[2] = s[2 * st]; break;
            case 3: {
                const size_t k = s[0];
                if (k * 3 + 2 >= plte.size()) { *err = "palette index out of range"; return false; }
                d[0] = plte[k * 3]; d[1] = plte[k * 3 + 1]; d[2] = plte[k * 3 + 2];
                if (k < trns.size()) d[3] = trns[k];
                break;
            }
            case 4: d[0] = d[1] = d[2] = s[0]; d[3] = s[st]; break;
            case 6: d[0] = s[0]; d[1] = s[st]; d[2] = s[2 * st]; d[3] = s[3 * st]; break;
            }
        }
        prev.swap(cur);
    }
    *w_out = w; *h_out = h;
    return true;
}

} // namespace mipt_png

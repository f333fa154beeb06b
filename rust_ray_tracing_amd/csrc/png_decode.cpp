// png_decode.cpp -- PNG reader for the OBJ/MTL loader's map_* textures.
// The reference decodes textures with the `image` crate (image::open(path).flipv().to_rgba8(), src/texture.rs:18),
// which is not vendored; this restates the published PNG/zlib formats (RFC 2083, RFC 1950/1951): every colour type and
// bit depth of the standard (grey 1/2/4/8/16, RGB 8/16, palette 1/2/4/8, grey+alpha and RGBA 8/16), all five scanline
// filters, Adam7 interlacing, tRNS for palettes and as a colour key.  Output is RGBA8, top row first (the caller applies
// flipv): low-depth grey is scaled to the full range and 16-bit samples are rounded, (x + 128) / 257, which is how the
// `image` crate narrows u16 to u8 -- from its published behaviour, not checked against it here ("parity unpinned" for
// 16-bit files; 8-bit and lower are lossless and checked against Pillow's decoder and Python's zlib in the tests).
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

// `limit`: the decoder knows how many bytes the image needs (IHDR); output beyond it is a corrupt or hostile stream
bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t limit) {
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
            if (out.size() + len > limit) return false;
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
                if (sym < 256) { if (out.size() >= limit) return false; out.push_back((uint8_t)sym); }
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
                    if (out.size() + len > limit) return false;
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
    if (interlace > 1) { *err = "unknown interlace method"; return false; }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!channels || !depth_ok) { *err = "unsupported colour type / bit depth"; return false; }
    const size_t bits = (size_t)channels * depth, bpp = bits >= 8 ? bits / 8 : 1;       // filter unit: whole bytes, at least 1
    if ((uint64_t)w * h > (1ull << 27)) { *err = "image larger than 2^27 pixels"; return false; }   // same budget as the JPEG path
    // exact size of the filtered image data (sum over the Adam7 passes): inflate may not produce more, and must produce this
    static const uint8_t px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1}, pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t need = 0;
    for (int pass = 0; pass < (interlace ? 7 : 1); pass++) {
        const uint32_t x0 = interlace ? px0[pass] : 0, y0 = interlace ? py0[pass] : 0, dx = interlace ? pdx[pass] : 1, dy = interlace ? pdy[pass] : 1;
        if (x0 >= w || y0 >= h) continue;
        const uint32_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        need += (((size_t)pw * bits + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw;
    raw.reserve(need);
    if (!inflate(idat.data(), idat.size(), raw, need)) { *err = "zlib stream is corrupt or longer than the image"; return false; }
    if (raw.size() < need) { *err = "image data too short"; return false; }
    // tRNS colour key for grey / RGB images: a pixel equal to it (at the file's bit depth) becomes transparent
    bool has_key = false;
    uint32_t key[3] = {0, 0, 0};
    if (ctype == 0 && trns.size() >= 2) { has_key = true; key[0] = (uint32_t)trns[0] << 8 | trns[1]; }
    if (ctype == 2 && trns.size() >= 6) { has_key = true; for (int k = 0; k < 3; k++) key[k] = (uint32_t)trns[2 * k] << 8 | trns[2 * k + 1]; }
    // sample -> 8 bits as the `image` crate's to_rgba8 does: 1/2/4-bit grey scaled to the full range, 16 bit rounded ((x + 128) / 257)
    auto to8 = [&](uint32_t v) -> uint8_t {
        return depth == 16 ? (uint8_t)((v + 128u) / 257u) : depth == 8 ? (uint8_t)v : (uint8_t)(v * (255u / ((1u << depth) - 1u)));
    };
    rgba->assign((size_t)w * h * 4, 255);
    // Adam7: seven reduced images, each filtered on its own; a non-interlaced file is the single pass (0, 0, 1, 1)
    size_t rp = 0;
    std::vector<uint8_t> prev, cur;
    for (int pass = 0; pass < (interlace ? 7 : 1); pass++) {
        const uint32_t x0 = interlace ? px0[pass] : 0, y0 = interlace ? py0[pass] : 0, dx = interlace ? pdx[pass] : 1, dy = interlace ? pdy[pass] : 1;
        if (x0 >= w || y0 >= h) continue;
        const uint32_t pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
        const size_t stride = ((size_t)pw * bits + 7) / 8;
        if (raw.size() < rp + (stride + 1) * ph) { *err = "image data too short"; return false; }
        prev.assign(stride, 0); cur.assign(stride, 0);
        for (uint32_t py = 0; py < ph; py++) {
            const uint8_t *row = &raw[rp];
            rp += stride + 1;
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
            auto sample = [&](size_t k) -> uint32_t {            // k-th sample of the row, at the file's bit depth
                if (depth == 16) return (uint32_t)cur[2 * k] << 8 | cur[2 * k + 1];
                if (depth == 8) return cur[k];
                const size_t bit = k * (size_t)depth;
                return (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
            };
            for (uint32_t px = 0; px < pw; px++) {
                uint8_t *d = &(*rgba)[((size_t)(y0 + py * dy) * w + (x0 + px * dx)) * 4];
                switch (ctype) {
                case 0: { const uint32_t g = sample(px); d[0] = d[1] = d[2] = to8(g); if (has_key && g == key[0]) d[3] = 0; break; }
                case 2: {
                    const uint32_t r = sample((size_t)px * 3), g = sample((size_t)px * 3 + 1), b = sample((size_t)px * 3 + 2);
                    d[0] = to8(r); d[1] = to8(g); d[2] = to8(b);
                    if (has_key && r == key[0] && g == key[1] && b == key[2]) d[3] = 0;
                    break;
                }
                case 3: {
                    const size_t k = sample(px);
                    if (k * 3 + 2 >= plte.size()) { *err = "palette index out of range"; return false; }
                    d[0] = plte[k * 3]; d[1] = plte[k * 3 + 1]; d[2] = plte[k * 3 + 2];
                    if (k < trns.size()) d[3] = trns[k];
                    break;
                }
                case 4: d[0] = d[1] = d[2] = to8(sample((size_t)px * 2)); d[3] = to8(sample((size_t)px * 2 + 1)); break;
                case 6: for (int k = 0; k < 4; k++) d[k] = to8(sample((size_t)px * 4 + k)); break;
                }
            }
            prev.swap(cur);
        }
    }
    *w_out = w; *h_out = h;
    return true;
}


// ---- writer: the image output of Renderer::render (reference src/renderer.rs:66-83, image::save_buffer) ---------------
// RGBA, 8 or 16 bits per sample (16-bit samples are taken in host order and stored big-endian), filter type 0, zlib
// "stored" blocks -- a valid PNG without a compressor; rows top first.
bool write_rgba(const std::string &path, uint32_t w, uint32_t h, int bits, const void *rgba, std::string *err) {
    if (w == 0 || h == 0 || (bits != 8 && bits != 16) || !rgba) { *err = "bad PNG write arguments"; return false; }
    struct CrcTable {                                   // function-local static of class type: initialised once, thread-safely (C++11)
        uint32_t t[256];
        CrcTable() { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; t[n] = c; } }
    };
    static const CrcTable crc_tab;
    const uint32_t *crc_table = crc_tab.t;
    auto crc = [&](uint32_t c, const uint8_t *p, size_t n) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 255u] ^ (c >> 8); return c; };
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { *err = "cannot create '" + path + "'"; return false; }
    auto put32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    auto chunk = [&](const char *tag, const std::vector<uint8_t> &data) {
        uint8_t hd[8];
        put32(hd, (uint32_t)data.size()); memcpy(hd + 4, tag, 4);
        uint32_t c = crc(0xffffffffu, hd + 4, 4);
        c = crc(c, data.data(), data.size()) ^ 0xffffffffu;
        uint8_t tail[4];
        put32(tail, c);
        return fwrite(hd, 1, 8, f) == 8 && (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) && fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    bool ok = fwrite(sig, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr(13);
    put32(&ihdr[0], w); put32(&ihdr[4], h); ihdr[8] = (uint8_t)bits; ihdr[9] = 6; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    ok = ok && chunk("IHDR", ihdr);
    // raw scanlines (filter byte 0 + samples), then a zlib stream of stored blocks
    const size_t row = (size_t)w * 4 * (size_t)(bits / 8), total = (row + 1) * h;
    std::vector<uint8_t> raw(total);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *d = &raw[(row + 1) * y];
        d[0] = 0;
        if (bits == 8) {
            memcpy(d + 1, (const uint8_t *)rgba + row * y, row);
        } else {
            const uint16_t *s = (const uint16_t *)rgba + (size_t)w * 4 * y;
            for (size_t i = 0; i < (size_t)w * 4; i++) { d[1 + 2 * i] = (uint8_t)(s[i] >> 8); d[2 + 2 * i] = (uint8_t)s[i]; }
        }
    }
    std::vector<uint8_t> z;
    z.reserve(total + total / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < total || pos == 0; ) {
        const size_t n = total - pos < 65535 ? total - pos : 65535;
        z.push_back(pos + n >= total ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        for (size_t i = 0; i < n; i++) { a += raw[pos + i]; if (a >= 65521u) a -= 65521u; b += a; if (b >= 65521u) b -= 65521u; }
        pos += n;
        if (n == 0) break;
    }
    uint8_t ad[4];
    put32(ad, b << 16 | a);
    z.insert(z.end(), ad, ad + 4);
    ok = ok && chunk("IDAT", z) && chunk("IEND", {});
    ok = (fclose(f) == 0) && ok;
    if (!ok) *err = "short write to '" + path + "'";
    return ok;
}

} // namespace mipt_png

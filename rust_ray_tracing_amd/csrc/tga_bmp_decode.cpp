// tga_bmp_decode.cpp -- Truevision TGA and Windows BMP readers for map_* textures.
//
// The reference opens textures with image::open (reference src/texture.rs:18), which also understands these two lossless
// formats that OBJ exports commonly carry.  Restated from the published file-format descriptions; output is RGBA8 with the
// first row on top (the caller applies flipv).  tests/test_obj_loader.py checks both against Pillow's readers; 5- and 6-bit
// fields are expanded with rounding, round(v * 255 / max) (Pillow truncates; the test states its own expectation there).
//   TGA: image types 1/2/3 and their run-length forms 9/10/11; 8-bit grey, 16-bit grey+alpha, 15/16/24/32-bit colour,
//        colour maps with 15/16/24/32-bit entries; either vertical and horizontal origin.
//   BMP: CORE / INFO / V4 / V5 headers; 1/4/8-bit palettes, 16/24/32-bit BI_RGB, 16/32-bit BI_BITFIELDS (with an alpha
//        mask when the header has one); bottom-up and top-down.  Run-length BMPs are reported as unsupported.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mipt_img {

namespace {

bool read_file(const std::string &path, std::vector<uint8_t> *buf, std::string *err) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { *err = "cannot open"; return false; }
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf->insert(buf->end(), tmp, tmp + got);
    fclose(f);
    return true;
}
inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
inline uint32_t le32(const uint8_t *p) { return le16(p) | le16(p + 2) << 16; }
inline uint8_t five_to_eight(uint32_t v) { return (uint8_t)((v * 527u + 23u) >> 6); }   // round(v * 255 / 31), as for BMP bit fields

// one TGA pixel / colour-map entry of `bits` bits -> RGBA
inline void tga_colour(const uint8_t *s, int bits, uint8_t *d) {
    if (bits == 15 || bits == 16) {
        const uint32_t v = le16(s);
        d[0] = five_to_eight((v >> 10) & 31); d[1] = five_to_eight((v >> 5) & 31); d[2] = five_to_eight(v & 31); d[3] = 255;
    } else {
        d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; d[3] = bits == 32 ? s[3] : 255;
    }
}

} // namespace

bool decode_tga(const std::string &path, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgba, std::string *err) {
    std::vector<uint8_t> b;
    if (!read_file(path, &b, err)) return false;
    if (b.size() < 18) { *err = "not a TGA file"; return false; }
    const int id_len = b[0], cmap_type = b[1], type = b[2], cmap_bits = b[7], bpp = b[16], desc = b[17];
    const uint32_t cmap_first = le16(&b[3]), cmap_len = le16(&b[5]), w = le16(&b[12]), h = le16(&b[14]);
    const bool rle = type >= 9;
    const int base = rle ? type - 8 : type;
    if (w == 0 || h == 0) { *err = "empty TGA image"; return false; }
    if (base < 1 || base > 3 || cmap_type > 1) { *err = "unsupported TGA image type"; return false; }
    if (base == 1 && !(cmap_type == 1 && (bpp == 8 || bpp == 16))) { *err = "bad colour-mapped TGA"; return false; }
    if (base == 2 && !(bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32)) { *err = "unsupported TGA pixel depth"; return false; }
    if (base == 3 && !(bpp == 8 || bpp == 16)) { *err = "unsupported TGA grey depth"; return false; }
    const size_t cmap_entry = cmap_type ? (size_t)(cmap_bits + 7) / 8 : 0;
    if (cmap_type && !(cmap_bits == 15 || cmap_bits == 16 || cmap_bits == 24 || cmap_bits == 32)) { *err = "unsupported TGA colour-map depth"; return false; }
    size_t pos = 18 + (size_t)id_len;
    const size_t cmap_pos = pos;
    pos += cmap_entry * cmap_len;
    if (pos > b.size()) { *err = "truncated TGA"; return false; }
    const size_t px = (size_t)(bpp + 7) / 8, n = (size_t)w * h;
    std::vector<uint8_t> raw(n * px);
    if (!rle) {
        if (b.size() - pos < raw.size()) { *err = "truncated TGA pixel data"; return false; }
        memcpy(raw.data(), &b[pos], raw.size());
    } else {
        size_t o = 0;
        while (o < n) {
            if (pos >= b.size()) { *err = "truncated TGA run-length data"; return false; }
            const int hd = b[pos++];
            size_t cnt = (size_t)(hd & 127) + 1;
            if (cnt > n - o) cnt = n - o;
            if (hd & 128) {
                if (b.size() - pos < px) { *err = "truncated TGA run-length data"; return false; }
                for (size_t i = 0; i < cnt; i++) memcpy(&raw[(o + i) * px], &b[pos], px);
                pos += px;
            } else {
                if (b.size() - pos < cnt * px) { *err = "truncated TGA run-length data"; return false; }
                memcpy(&raw[o * px], &b[pos], cnt * px);
                pos += cnt * px;
            }
            o += cnt;
        }
    }
    rgba->assign(n * 4, 255);
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *s = &raw[((size_t)y * w + x) * px];
            const uint32_t oy = top_down ? y : h - 1 - y, ox = right_left ? w - 1 - x : x;
            uint8_t *d = &(*rgba)[((size_t)oy * w + ox) * 4];
            if (base == 3) {
                d[0] = d[1] = d[2] = s[0]; d[3] = bpp == 16 ? s[1] : 255;
            } else if (base == 2) {
                tga_colour(s, bpp, d);
            } else {
                const uint32_t idx = (bpp == 16 ? le16(s) : s[0]);
                if (idx < cmap_first || idx - cmap_first >= cmap_len) { *err = "TGA colour index out of range"; return false; }
                tga_colour(&b[cmap_pos + (size_t)(idx - cmap_first) * cmap_entry], cmap_bits, d);
            }
        }
    *w_out = w; *h_out = h;
    return true;
}

bool decode_bmp(const std::string &path, uint32_t *w_out, uint32_t *h_out, std::vector<uint8_t> *rgba, std::string *err) {
    std::vector<uint8_t> b;
    if (!read_file(path, &b, err)) return false;
    if (b.size() < 26 || b[0] != 'B' || b[1] != 'M') { *err = "not a BMP file"; return false; }
    const uint32_t data_off = le32(&b[10]), hdr = le32(&b[14]);
    if (!(hdr == 12 || hdr == 40 || hdr == 52 || hdr == 56 || hdr == 108 || hdr == 124) || b.size() < 14 + (size_t)hdr) { *err = "unsupported BMP header"; return false; }
    int64_t w, hh;
    uint32_t bpp, comp = 0, used = 0;
    if (hdr == 12) { w = le16(&b[18]); hh = le16(&b[20]); bpp = le16(&b[24]); }
    else { w = (int32_t)le32(&b[18]); hh = (int32_t)le32(&b[22]); bpp = le16(&b[28]); comp = le32(&b[30]); used = le32(&b[46]); }
    const bool top_down = hh < 0;
    const int64_t h = top_down ? -hh : hh;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) { *err = "bad BMP dimensions"; return false; }
    if (comp == 1 || comp == 2) { *err = "run-length BMP is not supported"; return false; }
    if (!(comp == 0 || comp == 3 || comp == 6)) { *err = "unsupported BMP compression"; return false; }
    if (!(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 16 || bpp == 24 || bpp == 32)) { *err = "unsupported BMP bit count"; return false; }
    // channel masks for 16/32-bit pixels
    uint32_t mask[4] = {0, 0, 0, 0};
    if (bpp == 16) { mask[0] = 0x7c00; mask[1] = 0x03e0; mask[2] = 0x001f; }
    if (bpp == 32) { mask[0] = 0x00ff0000; mask[1] = 0x0000ff00; mask[2] = 0x000000ff; }
    if ((comp == 3 || comp == 6) && (bpp == 16 || bpp == 32)) {
        const size_t mo = 14 + 40;                            // masks follow the 40-byte part (inside V2+ headers, after it for INFO)
        const size_t need = mo + (comp == 6 || hdr >= 56 ? 16 : 12);
        if (b.size() < need) { *err = "truncated BMP masks"; return false; }
        for (int k = 0; k < 3; k++) mask[k] = le32(&b[mo + 4 * k]);
        if (comp == 6 || hdr >= 56) mask[3] = le32(&b[mo + 12]);
    }
    // palette
    std::vector<uint8_t> pal;
    if (bpp <= 8) {
        const size_t entry = hdr == 12 ? 3 : 4, n = used ? used : (size_t)1 << bpp, po = 14 + (size_t)hdr;
        if (n > 256 || b.size() < po + n * entry) { *err = "truncated BMP palette"; return false; }
        pal.resize(n * 3);
        for (size_t i = 0; i < n; i++) { pal[i * 3] = b[po + i * entry + 2]; pal[i * 3 + 1] = b[po + i * entry + 1]; pal[i * 3 + 2] = b[po + i * entry]; }
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if (data_off > b.size() || b.size() - data_off < stride * (size_t)h) { *err = "truncated BMP pixel data"; return false; }
    auto channel = [](uint32_t v, uint32_t m) -> uint8_t {      // masked field scaled to 8 bits
        if (!m) return 0;
        int sh = 0;
        while (!((m >> sh) & 1u)) sh++;
        const uint32_t f = (v & m) >> sh, mx = m >> sh;
        return (uint8_t)((f * 255u + mx / 2u) / mx);
    };
    rgba->assign((size_t)w * h * 4, 255);
    for (int64_t y = 0; y < h; y++) {
        const uint8_t *row = &b[data_off + stride * (size_t)y];
        const int64_t oy = top_down ? y : h - 1 - y;
        for (int64_t x = 0; x < w; x++) {
            uint8_t *d = &(*rgba)[((size_t)oy * w + x) * 4];
            if (bpp <= 8) {
                const uint32_t idx = bpp == 8 ? row[x] : bpp == 4 ? (row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15u : (row[x >> 3] >> (7 - (x & 7))) & 1u;
                if ((size_t)idx * 3 + 2 >= pal.size()) { *err = "BMP palette index out of range"; return false; }
                d[0] = pal[idx * 3]; d[1] = pal[idx * 3 + 1]; d[2] = pal[idx * 3 + 2];
            } else if (bpp == 24) {
                d[0] = row[x * 3 + 2]; d[1] = row[x * 3 + 1]; d[2] = row[x * 3];
            } else {
                const uint32_t v = bpp == 16 ? le16(&row[x * 2]) : le32(&row[x * 4]);
                d[0] = channel(v, mask[0]); d[1] = channel(v, mask[1]); d[2] = channel(v, mask[2]);
                if (mask[3]) d[3] = channel(v, mask[3]);
            }
        }
    }
    *w_out = (uint32_t)w; *h_out = (uint32_t)h;
    return true;
}

} // namespace mipt_img

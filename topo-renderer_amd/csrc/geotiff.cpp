// geotiff.cpp -- see geotiff.hpp.
#include "geotiff.hpp"

#include <zlib.h>

#include <cstring>

#include "../../include/topo_hip.h"

namespace topo {

namespace {

struct Reader {
    const uint8_t* d;
    size_t n;
    bool be;
    bool ok(uint64_t off, uint64_t len) const { return off <= n && len <= n - off; }
    uint16_t u16(uint64_t o) const { return be ? (uint16_t)((d[o] << 8) | d[o + 1]) : (uint16_t)(d[o] | (d[o + 1] << 8)); }
    uint32_t u32(uint64_t o) const {
        return be ? ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]
                  : ((uint32_t)d[o + 3] << 24) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 1] << 8) | d[o];
    }
    double f64(uint64_t o) const {
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | d[be ? o + i : o + 7 - i];
        double r;
        memcpy(&r, &v, 8);
        return r;
    }
};

struct Entry {
    uint16_t tag = 0, type = 0;
    uint32_t count = 0;
    uint64_t value_off = 0;      // where the values are
};

size_t type_size(uint16_t t) {
    switch (t) {
        case 1: case 2: case 6: case 7: return 1;
        case 3: case 8: return 2;
        case 4: case 9: case 11: return 4;
        case 5: case 10: case 12: return 8;
        default: return 0;
    }
}

// unsigned integer values of a BYTE / SHORT / LONG entry
bool uints(const Reader& r, const Entry& e, std::vector<uint64_t>& out) {
    out.clear();
    for (uint32_t i = 0; i < e.count; ++i) {
        if (e.type == 1) out.push_back(r.d[e.value_off + i]);
        else if (e.type == 3) out.push_back(r.u16(e.value_off + 2ull * i));
        else if (e.type == 4) out.push_back(r.u32(e.value_off + 4ull * i));
        else return false;
    }
    return true;
}

bool doubles(const Reader& r, const Entry& e, std::vector<double>& out) {
    out.clear();
    if (e.type != 12) return false;
    for (uint32_t i = 0; i < e.count; ++i) out.push_back(r.f64(e.value_off + 8ull * i));
    return true;
}

}  // namespace

int tiff_parse(const uint8_t* data, size_t n, TiffInfo& info, std::string& err) {
    info = TiffInfo{};
    if (!data || n < 8) { err = "not a TIFF: shorter than its header"; return TOPO_ERR_INVALID; }
    Reader r{data, n, false};
    if (data[0] == 'I' && data[1] == 'I') r.be = false;
    else if (data[0] == 'M' && data[1] == 'M') r.be = true;
    else { err = "not a TIFF: bad byte-order mark"; return TOPO_ERR_INVALID; }
    const uint16_t magic = r.u16(2);
    if (magic == 43) { err = "BigTIFF is not supported"; return TOPO_ERR_UNSUPPORTED; }
    if (magic != 42) { err = "not a TIFF: bad magic"; return TOPO_ERR_INVALID; }
    info.big_endian = r.be;
    const uint64_t ifd = r.u32(4);
    if (!r.ok(ifd, 2)) { err = "IFD offset outside the file"; return TOPO_ERR_INVALID; }
    const uint16_t n_entries = r.u16(ifd);
    if (!r.ok(ifd + 2, 12ull * n_entries)) { err = "IFD runs past the end of the file"; return TOPO_ERR_INVALID; }
    uint32_t bits = 1, spp = 1, sample_format = 1, planar = 1, rows_per_strip = 0xFFFFFFFFu, tile_w = 0, tile_h = 0;
    std::vector<uint64_t> strip_off, strip_cnt, tile_off, tile_cnt, tmp;
    for (uint16_t i = 0; i < n_entries; ++i) {
        const uint64_t o = ifd + 2 + 12ull * i;
        Entry e;
        e.tag = r.u16(o); e.type = r.u16(o + 2); e.count = r.u32(o + 4);
        const size_t ts = type_size(e.type);
        if (ts == 0) continue;                                   // unknown field type: skipped, as readers do
        const uint64_t total = (uint64_t)ts * e.count;
        e.value_off = total <= 4 ? o + 8 : r.u32(o + 8);
        if (!r.ok(e.value_off, total)) { err = "tag " + std::to_string(e.tag) + " points outside the file"; return TOPO_ERR_INVALID; }
        auto one = [&](uint32_t& dst) { if (uints(r, e, tmp) && !tmp.empty()) dst = (uint32_t)tmp[0]; };
        switch (e.tag) {
            case 256: one(info.width); break;
            case 257: one(info.height); break;
            case 258: one(bits); break;
            case 259: one(info.compression); break;
            case 273: uints(r, e, strip_off); break;
            case 277: one(spp); break;
            case 278: one(rows_per_strip); break;
            case 279: uints(r, e, strip_cnt); break;
            case 284: one(planar); break;
            case 317: one(info.predictor); break;
            case 322: one(tile_w); break;
            case 323: one(tile_h); break;
            case 324: uints(r, e, tile_off); break;
            case 325: uints(r, e, tile_cnt); break;
            case 339: one(sample_format); break;
            case 33550: info.has_pixel_scale = doubles(r, e, info.pixel_scale); break;
            case 33922: info.has_tie_points = doubles(r, e, info.tie_points); break;
            case 34264: info.has_model_transformation = doubles(r, e, info.model_transformation); break;
            default: break;
        }
    }
    if (info.width == 0 || info.height == 0) { err = "ImageWidth / ImageLength missing"; return TOPO_ERR_INVALID; }
    if (info.width > 32768 || info.height > 32768) { err = "image larger than 32768 x 32768"; return TOPO_ERR_UNSUPPORTED; }
    if (bits != 32 || sample_format != 3 || spp != 1) { err = "only one 32-bit IEEE float sample per pixel is supported (DecodingResult::F32)"; return TOPO_ERR_UNSUPPORTED; }
    if (planar != 1 && spp != 1) { err = "planar configuration 2 is not supported"; return TOPO_ERR_UNSUPPORTED; }
    if (info.compression != 1 && info.compression != 5 && info.compression != 8 && info.compression != 32946 && info.compression != 32773) {
        err = "compression " + std::to_string(info.compression) + " is not supported";
        return TOPO_ERR_UNSUPPORTED;
    }
    if (info.predictor < 1 || info.predictor > 3) { err = "unknown predictor"; return TOPO_ERR_UNSUPPORTED; }
    if (!tile_off.empty()) {
        if (tile_w == 0 || tile_h == 0 || tile_w > 32768 || tile_h > 32768) { err = "bad tile size"; return TOPO_ERR_INVALID; }
        const uint32_t tx = (info.width + tile_w - 1) / tile_w, ty = (info.height + tile_h - 1) / tile_h;
        if ((uint64_t)tx * ty > (1u << 20)) { err = "more than 2^20 tiles"; return TOPO_ERR_UNSUPPORTED; }
        if (tile_off.size() != (size_t)tx * ty || tile_cnt.size() != tile_off.size()) { err = "tile offset/byte-count tables have the wrong length"; return TOPO_ERR_INVALID; }
        for (uint32_t j = 0; j < ty; ++j)
            for (uint32_t i = 0; i < tx; ++i) {
                const size_t k = (size_t)j * tx + i;
                info.segments.push_back(TiffSegment{tile_off[k], tile_cnt[k], i * tile_w, j * tile_h, tile_w, tile_h});
            }
    } else {
        if (strip_off.empty() || strip_cnt.size() != strip_off.size()) { err = "strip offset/byte-count tables missing or of different length"; return TOPO_ERR_INVALID; }
        if (rows_per_strip == 0) { err = "RowsPerStrip is 0"; return TOPO_ERR_INVALID; }
        const uint32_t rps = rows_per_strip > info.height ? info.height : rows_per_strip;
        const uint32_t ns = (info.height + rps - 1) / rps;
        if (strip_off.size() != ns) { err = "strip tables have the wrong length"; return TOPO_ERR_INVALID; }
        for (uint32_t k = 0; k < ns; ++k) {
            const uint32_t y0 = k * rps, hh = y0 + rps > info.height ? info.height - y0 : rps;
            info.segments.push_back(TiffSegment{strip_off[k], strip_cnt[k], 0, y0, info.width, hh});
        }
    }
    uint64_t staged = 0;
    for (const TiffSegment& s : info.segments) {
        if (!r.ok(s.offset, s.bytes)) { err = "a strip/tile lies outside the file"; return TOPO_ERR_INVALID; }
        staged += (uint64_t)s.w * s.h * 4;
    }
    // bounds on what a file can make the decoder allocate: a million segments, 8 GiB of decompressed samples
    if (info.segments.size() > (1u << 20) || staged > (1ull << 33)) { err = "more strips/tiles or padded samples than the decoder accepts"; return TOPO_ERR_UNSUPPORTED; }
    return TOPO_OK;
}

namespace {

// TIFF LZW (TIFF 6.0 section 13): MSB-first codes of 9..12 bits, ClearCode 256, EOI 257, the code width grows one code early.
bool lzw_decode(const uint8_t* in, size_t n, uint8_t* out, size_t want) {
    struct Ent { uint16_t prev; uint16_t len; uint8_t first, last; };
    std::vector<Ent> tab(4096);
    for (int i = 0; i < 256; ++i) tab[i] = Ent{0xFFFF, 1, (uint8_t)i, (uint8_t)i};
    uint32_t next = 258, width = 9, bitbuf = 0, nbits = 0;
    size_t ip = 0, op = 0;
    int prev = -1;
    auto emit = [&](int code) -> bool {          // writes the string of `code` at op
        const uint16_t len = tab[code].len;
        if (op + len > want) return false;
        size_t p = op + len;
        for (int c = code; c != 0xFFFF; c = tab[c].prev) out[--p] = tab[c].last;
        op += len;
        return true;
    };
    while (op < want) {
        while (nbits < width) {
            if (ip >= n) return false;
            bitbuf = (bitbuf << 8) | in[ip++];
            nbits += 8;
        }
        const int code = (int)((bitbuf >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; prev = -1; continue; }
        if (prev < 0) {
            if (code >= 256) return false;
            if (!emit(code)) return false;
        } else if ((uint32_t)code < next) {
            if (!emit(code)) return false;
            if (next < 4096) { tab[next] = Ent{(uint16_t)prev, (uint16_t)(tab[prev].len + 1), tab[prev].first, tab[code].first}; ++next; }
        } else if ((uint32_t)code == next && next < 4096) {
            tab[next] = Ent{(uint16_t)prev, (uint16_t)(tab[prev].len + 1), tab[prev].first, tab[prev].first};
            ++next;
            if (!emit(code)) return false;
        } else {
            return false;
        }
        prev = code;
        if (next + 1 >= (1u << width) && width < 12) ++width;      // "one code early"
    }
    return op == want;
}

bool packbits_decode(const uint8_t* in, size_t n, uint8_t* out, size_t want) {
    size_t ip = 0, op = 0;
    while (op < want && ip < n) {
        const int8_t c = (int8_t)in[ip++];
        if (c >= 0) {
            const size_t k = (size_t)c + 1;
            if (ip + k > n || op + k > want) return false;
            memcpy(out + op, in + ip, k);
            ip += k; op += k;
        } else if (c != -128) {
            const size_t k = (size_t)(1 - c);
            if (ip >= n || op + k > want) return false;
            memset(out + op, in[ip++], k);
            op += k;
        }
    }
    return op == want;
}

}  // namespace

int tiff_segment_bytes(const uint8_t* data, size_t n, const TiffInfo& info, const TiffSegment& seg, uint8_t* out, std::string& err) {
    const size_t want = (size_t)seg.w * seg.h * 4;
    if (seg.offset > n || seg.bytes > n - seg.offset) { err = "segment outside the file"; return TOPO_ERR_INVALID; }
    const uint8_t* src = data + seg.offset;
    bool ok = false;
    switch (info.compression) {
        case 1:
            ok = seg.bytes >= want;
            if (ok) memcpy(out, src, want);
            break;
        case 8: case 32946: {
            uLongf got = (uLongf)want;
            const int rc = uncompress(out, &got, src, (uLong)seg.bytes);
            ok = (rc == Z_OK || rc == Z_BUF_ERROR) && got == want;      // a padded last strip may hold more than `want`
            if (rc == Z_BUF_ERROR && got == want) ok = true;
            break;
        }
        case 5: ok = lzw_decode(src, seg.bytes, out, want); break;
        case 32773: ok = packbits_decode(src, seg.bytes, out, want); break;
        default: break;
    }
    if (!ok) { err = "a strip/tile does not decompress to its nominal size"; return TOPO_ERR_INVALID; }
    return TOPO_OK;
}

}  // namespace topo

// Image files for Detector::load_image / detect(std::string) (reference: yolo_v2_class.cpp:111-149, which decodes through
// the vendored stb_image with three channels forced).  A dependency-free reader for what the robot's callers hand over:
//   * PNG  : 1/2/4/8/16-bit grey, grey+alpha, RGB, RGBA, palette; interlaced or not (inflate written out below)
//   * JPEG : baseline / extended sequential Huffman, 8-bit, grey or YCbCr with 1x1 / 2x1 / 1x2 / 2x2 luma sampling,
//            restart intervals; progressive files are refused with a message
//   * PNM  : binary P5 / P6
// Output is always 8-bit RGB, interleaved (grey replicated, alpha dropped, 16-bit samples reduced to their high byte):
// what stbi_load(..., 3) hands the reference.  File I/O only -- nothing here is on the inference path.
// Parity: PNG and PNM are exact by definition of the formats; JPEG decoders differ in their IDCT and chroma upsampling by
// one or two LSB (tests/test_eval_host.py compares with Pillow's decode at that tolerance): parity unpinned against
// stb_image itself.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

typedef unsigned char u8;

[[noreturn]] void fail(const std::string &m) { throw std::runtime_error("load_image: " + m); }

// ---------------------------------------------------------------------------------------------------------------
// inflate (RFC 1951) inside a zlib wrapper (RFC 1950)
// ---------------------------------------------------------------------------------------------------------------
struct BitReader {
    const u8 *p, *end;
    unsigned long long acc = 0;
    int n = 0;
    BitReader(const u8 *b, const u8 *e) : p(b), end(e) {}
    unsigned bits(int k)
    {
        while (n < k) {
            if (p >= end) fail("truncated deflate stream");
            acc |= (unsigned long long)(*p++) << n;
            n += 8;
        }
        const unsigned v = (unsigned)(acc & ((1ull << k) - 1));
        acc >>= k; n -= k;
        return v;
    }
    void align() { acc >>= (n & 7); n -= (n & 7); }
};

struct Huff {                       // canonical code: symbols sorted by (length, value)
    unsigned short count[16], symbol[320];
    void build(const u8 *len, int nsym)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < nsym; ++i) ++count[len[i]];
        count[0] = 0;
        unsigned short off[16];
        off[1] = 0;
        for (int l = 1; l < 15; ++l) off[l + 1] = off[l] + count[l];
        for (int i = 0; i < nsym; ++i) if (len[i]) symbol[off[len[i]]++] = (unsigned short)i;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= (int)br.bits(1);
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        fail("bad Huffman code in deflate stream");
    }
};

std::vector<u8> inflate_zlib(const std::vector<u8> &z, size_t expect)
{
    if (z.size() < 6 || (z[0] & 15) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 32)) fail("not a zlib stream");
    BitReader br(z.data() + 2, z.data() + z.size());
    std::vector<u8> out;
    out.reserve(expect);
    static const unsigned short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const u8 lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const unsigned short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const u8 dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const unsigned last = br.bits(1), type = br.bits(2);
        if (type == 0) {
            br.align();
            const unsigned len = br.bits(16), nlen = br.bits(16);
            if ((len ^ nlen) != 0xffffu) fail("bad stored block");
            for (unsigned i = 0; i < len; ++i) out.push_back((u8)br.bits(8));
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            u8 len[320];
            if (type == 1) {
                for (int i = 0; i < 288; ++i) len[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                hl.build(len, 288);
                for (int i = 0; i < 30; ++i) len[i] = 5;
                hd.build(len, 30);
            } else {
                const int nl = (int)br.bits(5) + 257, nd = (int)br.bits(5) + 1, nc = (int)br.bits(4) + 4;
                static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                u8 cl[19] = {0};
                for (int i = 0; i < nc; ++i) cl[order[i]] = (u8)br.bits(3);
                Huff hc;
                hc.build(cl, 19);
                int i = 0;
                while (i < nl + nd) {
                    const int sym = hc.decode(br);
                    if (sym < 16) len[i++] = (u8)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!i) fail("bad code lengths"); val = len[i - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (i + rep > nl + nd) fail("bad code lengths");
                        while (rep--) len[i++] = (u8)val;
                    }
                }
                hl.build(len, nl);
                hd.build(len + nl, nd);
            }
            for (;;) {
                const int sym = hl.decode(br);
                if (sym < 256) out.push_back((u8)sym);
                else if (sym == 256) break;
                else {
                    if (sym > 285) fail("bad length symbol");
                    const unsigned l = lbase[sym - 257] + br.bits(lext[sym - 257]);
                    const int ds = hd.decode(br);
                    if (ds > 29) fail("bad distance symbol");
                    const size_t d = dbase[ds] + br.bits(dext[ds]);
                    if (d > out.size()) fail("distance beyond the window");
                    const size_t from = out.size() - d;
                    for (unsigned i = 0; i < l; ++i) { const u8 b = out[from + i]; out.push_back(b); }
                }
            }
        } else fail("bad deflate block type");
        if (last) break;
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------------------------
unsigned be32(const u8 *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// un-filter `rows` scanlines of `stride` bytes each (a filter byte in front of every one) in place; returns the next input
const u8 *unfilter(const u8 *in, const u8 *end, u8 *dst, int rows, size_t stride, int bpp)
{
    std::vector<u8> zero(stride, 0);
    const u8 *prev = zero.data();
    for (int y = 0; y < rows; ++y) {
        if (in + 1 + stride > end) fail("PNG: image data too short");
        const int ft = *in++;
        u8 *cur = dst + (size_t)y * stride;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = in[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: fail("PNG: bad filter type");
            }
            cur[i] = (u8)v;
        }
        in += stride;
        prev = cur;
    }
    return in;
}

void decode_png(const std::vector<u8> &f, int &w, int &h, std::vector<u8> &rgb)
{
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<u8> idat, plte;
    bool have_hdr = false;
    while (pos + 12 <= f.size()) {
        const unsigned len = be32(&f[pos]);
        const char *tag = (const char *)&f[pos + 4];
        if (pos + 12 + (size_t)len > f.size()) fail("PNG: truncated chunk");
        const u8 *d = &f[pos + 8];
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len < 13) fail("PNG: bad IHDR");
            w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28) || d[10] || d[11] || interlace > 1) fail("PNG: unsupported header");
            have_hdr = true;
        } else if (!std::memcmp(tag, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!std::memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || idat.empty()) fail("PNG: no image data");
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: fail("PNG: bad colour type");
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16))
        fail("PNG: bad bit depth");
    if (ctype == 3 && plte.size() < 3) fail("PNG: palette missing");
    const int bits = ch * depth, bpp = bits >= 8 ? bits / 8 : 1;
    auto row_bytes = [&](int pw) { return ((size_t)pw * bits + 7) / 8; };
    // pass geometry: one pass, or Adam7
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t expect = 0;
    const int npass = interlace ? 7 : 1;
    for (int p = 0; p < npass; ++p) {
        const int pw = interlace ? (w - xs[p] + dx[p] - 1) / dx[p] : w, ph = interlace ? (h - ys[p] + dy[p] - 1) / dy[p] : h;
        if (pw > 0 && ph > 0) expect += (row_bytes(pw) + 1) * ph;
    }
    const std::vector<u8> raw = inflate_zlib(idat, expect);
    rgb.assign((size_t)w * h * 3, 0);
    const u8 *in = raw.data(), *end = raw.data() + raw.size();
    std::vector<u8> buf;
    for (int p = 0; p < npass; ++p) {
        const int pw = interlace ? (w - xs[p] + dx[p] - 1) / dx[p] : w, ph = interlace ? (h - ys[p] + dy[p] - 1) / dy[p] : h;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = row_bytes(pw);
        buf.assign(stride * ph, 0);
        in = unfilter(in, end, buf.data(), ph, stride, bpp);
        for (int y = 0; y < ph; ++y) {
            const u8 *row = &buf[(size_t)y * stride];
            for (int x = 0; x < pw; ++x) {
                int s[4] = {0, 0, 0, 0};
                for (int k = 0; k < ch; ++k) {
                    if (depth == 8) s[k] = row[x * ch + k];
                    else if (depth == 16) s[k] = row[(x * ch + k) * 2];                       // the high byte
                    else {
                        const int bi = x * depth, v = (row[bi >> 3] >> (8 - depth - (bi & 7))) & ((1 << depth) - 1);
                        s[k] = ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
                    }
                }
                const int ox = interlace ? xs[p] + x * dx[p] : x, oy = interlace ? ys[p] + y * dy[p] : y;
                u8 *o = &rgb[((size_t)oy * w + ox) * 3];
                if (ctype == 3) {
                    if ((size_t)s[0] * 3 + 2 >= plte.size()) fail("PNG: palette index out of range");
                    o[0] = plte[s[0] * 3]; o[1] = plte[s[0] * 3 + 1]; o[2] = plte[s[0] * 3 + 2];
                } else if (ch <= 2) o[0] = o[1] = o[2] = (u8)s[0];
                else { o[0] = (u8)s[0]; o[1] = (u8)s[1]; o[2] = (u8)s[2]; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// JPEG (ITU T.81 baseline / extended sequential, Huffman, 8 bit)
// ---------------------------------------------------------------------------------------------------------------
struct JHuff {
    u8 bits[17], vals[256];
    int mincode[17], maxcode[18], valptr[17];
    bool set = false;
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        set = true;
    }
};

struct JComp { int id, h, v, tq, td, ta, pred; int bw, bh; std::vector<u8> plane; };   // plane: bw x bh samples (whole blocks)

struct JBits {
    const u8 *p, *end;
    unsigned acc = 0;
    int n = 0;
    bool marker = false;
    int bit()
    {
        if (!n) {
            int b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xff) {
                    const int b2 = p < end ? *p : 0xd9;
                    if (b2 == 0) ++p;
                    else { marker = true; --p; b = 0; }          // a marker ends the entropy-coded segment: feed zeros
                }
            }
            acc = (unsigned)b; n = 8;
        }
        --n;
        return (acc >> n) & 1;
    }
    int receive(int s) { int v = 0; while (s--) v = (v << 1) | bit(); return v; }
    void reset() { n = 0; marker = false; }
};

int jdecode(JBits &br, const JHuff &h)
{
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    fail("JPEG: bad Huffman code");
}

int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

const u8 zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// separable 8x8 inverse DCT in double (the defining formula, T.81 A.3.3), +128, clamped
void idct_block(const int *coef, u8 *out, int stride)
{
    static double cs[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) cs[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * 0.5;
        init = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; ++y)                     // rows: over u
        for (int x = 0; x < 8; ++x) {
            double s = 0;
            for (int u = 0; u < 8; ++u) s += cs[x][u] * coef[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            double s = 0;
            for (int v = 0; v < 8; ++v) s += cs[y][v] * tmp[v * 8 + x];
            const int r = (int)std::floor(s + 128.5);
            out[y * stride + x] = (u8)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
}

void decode_jpeg(const std::vector<u8> &f, int &w, int &h, std::vector<u8> &rgb)
{
    size_t pos = 2;
    unsigned short qt[4][64];
    bool qset[4] = {false, false, false, false};
    JHuff dc[4], ac[4];
    std::vector<JComp> comp;
    int restart = 0, adobe_transform = -1;
    bool have_sof = false;
    auto need = [&](size_t n) { if (pos + n > f.size()) fail("JPEG: truncated file"); };
    for (;;) {
        need(4);
        if (f[pos] != 0xff) fail("JPEG: marker expected");
        while (pos < f.size() && f[pos] == 0xff) ++pos;
        need(1);
        const int m = f[pos++];
        if (m == 0xd9) fail("JPEG: no scan before the end of the image");
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
        need(2);
        const size_t len = ((size_t)f[pos] << 8) | f[pos + 1];
        if (len < 2) fail("JPEG: bad segment length");
        need(len);
        const u8 *d = &f[pos + 2];
        const size_t n = len - 2;
        if (m == 0xdb) {
            size_t i = 0;
            while (i < n) {
                const int pq = d[i] >> 4, tq = d[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) fail("JPEG: bad quantisation table");
                for (int k = 0; k < 64; ++k) { qt[tq][zigzag[k]] = pq ? (unsigned short)((d[i] << 8) | d[i + 1]) : d[i]; i += pq ? 2 : 1; }
                qset[tq] = true;
            }
        } else if (m == 0xc4) {
            size_t i = 0;
            while (i < n) {
                const int tc = d[i] >> 4, th = d[i] & 15;
                ++i;
                if (tc > 1 || th > 3 || i + 16 > n) fail("JPEG: bad Huffman table");
                JHuff &t = tc ? ac[th] : dc[th];
                int total = 0;
                t.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = d[i + l - 1]; total += t.bits[l]; }
                i += 16;
                if (total > 256 || i + total > n) fail("JPEG: bad Huffman table");
                std::memcpy(t.vals, d + i, total);
                i += total;
                t.build();
            }
        } else if (m == 0xc0 || m == 0xc1) {
            if (n < 6 || d[0] != 8) fail("JPEG: only 8-bit samples are supported");
            h = (d[1] << 8) | d[2]; w = (d[3] << 8) | d[4];
            const int nc = d[5];
            if (w <= 0 || h <= 0 || (nc != 1 && nc != 3) || n < (size_t)6 + 3 * nc) fail("JPEG: unsupported frame header");
            comp.resize(nc);
            for (int c = 0; c < nc; ++c) {
                comp[c].id = d[6 + 3 * c]; comp[c].h = d[7 + 3 * c] >> 4; comp[c].v = d[7 + 3 * c] & 15; comp[c].tq = d[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) fail("JPEG: unsupported sampling factors");
            }
            if (nc == 3 && (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1)) fail("JPEG: unsupported chroma sampling");
            if (nc == 1) comp[0].h = comp[0].v = 1;
            have_sof = true;
        } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
            fail(m == 0xc2 ? "progressive JPEG is not supported (baseline only)" : "JPEG: unsupported coding process");
        } else if (m == 0xdd) {
            if (n < 2) fail("JPEG: bad DRI");
            restart = (d[0] << 8) | d[1];
        } else if (m == 0xee && n >= 12 && !std::memcmp(d, "Adobe", 5)) {
            adobe_transform = d[11];
        } else if (m == 0xda) {
            if (!have_sof) fail("JPEG: scan before the frame header");
            const int ns = d[0];
            if (ns != (int)comp.size() || n < (size_t)1 + 2 * ns + 3) fail("JPEG: only one interleaved scan is supported");
            for (int k = 0; k < ns; ++k) {
                int c = -1;
                for (size_t q = 0; q < comp.size(); ++q) if (comp[q].id == d[1 + 2 * k]) c = (int)q;
                if (c < 0) fail("JPEG: scan names an unknown component");
                comp[c].td = d[2 + 2 * k] >> 4; comp[c].ta = d[2 + 2 * k] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !dc[comp[c].td].set || !ac[comp[c].ta].set || !qset[comp[c].tq]) fail("JPEG: missing table");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    // ---- entropy-coded data: MCUs of hmax x vmax blocks of 8x8 ----
    const int hmax = comp[0].h, vmax = comp[0].v;
    const int mcux = (w + 8 * hmax - 1) / (8 * hmax), mcuy = (h + 8 * vmax - 1) / (8 * vmax);
    for (JComp &c : comp) {
        c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8; c.pred = 0;
        c.plane.assign((size_t)c.bw * c.bh, 0);
    }
    JBits br;
    br.p = f.data() + pos; br.end = f.data() + f.size();
    int coef[64], left = restart;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart && left == 0) {
                br.reset();                                             // byte-align, then the RSTn marker
                while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) ++br.p;
                if (br.p + 1 < br.end) br.p += 2;
                for (JComp &c : comp) c.pred = 0;
                left = restart;
            }
            for (JComp &c : comp)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        std::memset(coef, 0, sizeof coef);
                        const int s = jdecode(br, dc[c.td]);
                        if (s > 11) fail("JPEG: bad DC size");
                        c.pred += extend(br.receive(s), s);
                        coef[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            const int rs = jdecode(br, ac[c.ta]), r = rs >> 4, sz = rs & 15;
                            if (!sz) { if (r == 15) { k += 16; continue; } break; }
                            k += r;
                            if (k > 63) fail("JPEG: coefficient index out of range");
                            coef[zigzag[k]] = extend(br.receive(sz), sz) * qt[c.tq][zigzag[k]];
                            ++k;
                        }
                        idct_block(coef, &c.plane[(size_t)((my * c.v + by) * 8) * c.bw + (mx * c.h + bx) * 8], c.bw);
                    }
            if (restart) --left;
        }
    // ---- chroma to full resolution ("fancy" triangle filter for 2:1, as the common decoders do), colour conversion ----
    rgb.assign((size_t)w * h * 3, 0);
    if (comp.size() == 1) {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) { const u8 v = comp[0].plane[(size_t)y * comp[0].bw + x]; u8 *o = &rgb[((size_t)y * w + x) * 3]; o[0] = o[1] = o[2] = v; }
        return;
    }
    const int cw = comp[1].bw;
    auto up = [&](const JComp &c, std::vector<int> &full) {      // values scaled by 16 (h2v2), 4 (h2v1 / h1v2) or 1
        full.assign((size_t)w * h, 0);
        const int need_w = (w + hmax - 1) / hmax, need_h = (h + vmax - 1) / vmax;      // real chroma samples
        auto at = [&](int x, int y) {
            x = x < 0 ? 0 : x >= need_w ? need_w - 1 : x; y = y < 0 ? 0 : y >= need_h ? need_h - 1 : y;
            return (int)c.plane[(size_t)y * cw + x];
        };
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int v;
                if (hmax == 2 && vmax == 2) {
                    const int cx = x >> 1, cy = y >> 1, nx = (x & 1) ? cx + 1 : cx - 1, ny = (y & 1) ? cy + 1 : cy - 1;
                    const int near_row = 3 * at(cx, cy) + at(cx, ny), far_row = 3 * at(nx, cy) + at(nx, ny);
                    v = (3 * near_row + far_row + ((x & 1) ? 7 : 8)) >> 4;
                } else if (hmax == 2) {
                    const int cx = x >> 1, nx = (x & 1) ? cx + 1 : cx - 1;
                    v = (3 * at(cx, y) + at(nx, y) + ((x & 1) ? 2 : 1)) >> 2;
                } else if (vmax == 2) {
                    const int cy = y >> 1, ny = (y & 1) ? cy + 1 : cy - 1;
                    v = (3 * at(x, cy) + at(x, ny) + ((y & 1) ? 2 : 1)) >> 2;
                } else v = at(x, y);
                full[(size_t)y * w + x] = v;
            }
    };
    std::vector<int> cb, cr;
    up(comp[1], cb);
    up(comp[2], cr);
    const bool ycc = adobe_transform != 0;                      // Adobe transform 0 = the three components ARE R, G, B
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int Y = comp[0].plane[(size_t)y * comp[0].bw + x], B = cb[(size_t)y * w + x], R = cr[(size_t)y * w + x];
            u8 *o = &rgb[((size_t)y * w + x) * 3];
            if (!ycc) { o[0] = (u8)Y; o[1] = (u8)B; o[2] = (u8)R; continue; }
            auto clamp = [](double v) { const int r = (int)std::floor(v + 0.5); return (u8)(r < 0 ? 0 : r > 255 ? 255 : r); };
            o[0] = clamp(Y + 1.402 * (R - 128));
            o[1] = clamp(Y - 0.344136 * (B - 128) - 0.714136 * (R - 128));
            o[2] = clamp(Y + 1.772 * (B - 128));
        }
}

// ---------------------------------------------------------------------------------------------------------------
// binary PNM
// ---------------------------------------------------------------------------------------------------------------
void decode_pnm(const std::vector<u8> &f, int &w, int &h, std::vector<u8> &rgb)
{
    const bool grey = f[1] == '5';
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < f.size()) {
        const u8 ch = f[pos];
        if (ch == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
        if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') { ++pos; continue; }
        if (ch < '0' || ch > '9') fail("bad PNM header");
        int v = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos] - '0'); if (v > (1 << 28)) fail("bad PNM header"); ++pos; }
        vals[got++] = v;
    }
    if (got != 3) fail("bad PNM header");
    ++pos;                                                       // the single whitespace byte behind maxval
    w = vals[0]; h = vals[1];
    if (w <= 0 || h <= 0 || vals[2] != 255 || (long long)w * h > (1ll << 28)) fail("unsupported PNM (8-bit binary P5 / P6 only)");
    const size_t need = (size_t)w * h * (grey ? 1 : 3);
    if (pos + need > f.size()) fail("short PNM");
    rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int k = 0; k < 3; ++k) rgb[i * 3 + k] = f[pos + (grey ? i : i * 3 + k)];
}

}  // namespace

// 8-bit interleaved RGB of an image file; throws std::runtime_error ("file not found" as the reference, cpp:131)
void y2_decode_image_file(const std::string &path, int &w, int &h, std::vector<unsigned char> &rgb)
{
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("file not found");
    std::vector<u8> f;
    u8 buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    std::fclose(fp);
    static const u8 png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() >= 8 && !std::memcmp(f.data(), png_sig, 8)) decode_png(f, w, h, rgb);
    else if (f.size() >= 4 && f[0] == 0xff && f[1] == 0xd8) decode_jpeg(f, w, h, rgb);
    else if (f.size() >= 8 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) decode_pnm(f, w, h, rgb);
    else fail("unknown image format (PNG, baseline JPEG and binary PNM are read)");
}

// C face for bindings and tests: 0 and a malloc'ed RGB buffer (free with y2_free_image_rgb), or -1 and the message in `err`
extern "C" int y2_decode_image_rgb(const char *path, int *w, int *h, unsigned char **rgb, char *err, int errlen)
{
    try {
        std::vector<unsigned char> v;
        int ww = 0, hh = 0;
        y2_decode_image_file(path ? path : "", ww, hh, v);
        unsigned char *p = (unsigned char *)std::malloc(v.size() ? v.size() : 1);
        if (!p) throw std::runtime_error("out of memory");
        std::memcpy(p, v.data(), v.size());
        *w = ww; *h = hh; *rgb = p;
        return 0;
    } catch (const std::exception &e) {
        if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.what());
        return -1;
    }
}

extern "C" void y2_free_image_rgb(unsigned char *rgb) { std::free(rgb); }

// image_io.cpp -- cv::imread(path, 0) / cv::imwrite stand-ins of the host mirror: 8-bit greyscale PGM (P5) and PNG.
// Everything a file says about itself is checked before a byte is allocated for it: dimensions (positive, bounded, equal
// to the expected camera size when the caller has one), chunk lengths and CRCs, the size of the inflated stream.  No
// function here throws on bad input; they return an empty image / false and an error text.
// PNG: colour types 0 / 2 / 4 / 6 at 8 or 16 bits per sample, non-interlaced and Adam7; colour is converted like OpenCV's
// RGB -> grey fixed point, (R*4899 + G*9617 + B*1868 + 8192) >> 14; 16-bit samples keep their high byte.
#include "duke.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <fstream>

namespace duke {

namespace {

constexpr size_t kMaxFileBytes = (size_t)1 << 31;        // 2 GiB: nothing this path reads is larger
constexpr long long kMaxPixels = 1ll << 28;              // 268 Mpixel (16384 x 16384)

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool dims_ok(long long w, long long h, int want_w, int want_h, std::string &err)
{
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535 || w * h > kMaxPixels) { err = "implausible image size"; return false; }
    if (want_w > 0 && (w != want_w || h != want_h)) { err = "image size differs from the configured camera size"; return false; }
    return true;
}

// ---- PGM ---------------------------------------------------------------------------------------------------------------
// buf_n bytes of the file are in buf_p (the whole file, or its head when total_n > buf_n is the file's size)
bool pgm_header(const uint8_t *buf_p, size_t buf_n, int &w, int &h, size_t &data_pos, std::string &err, size_t total_n = 0)
{
    struct { const uint8_t *p; size_t n; size_t size() const { return n; } uint8_t operator[](size_t i) const { return p[i]; } } buf{buf_p, buf_n};
    size_t pos = 2;
    long long vals[3] = {0, 0, 0};
    for (int got = 0; got < 3;) {
        while (pos < buf.size() && (buf[pos] == ' ' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\t')) pos++;
        if (pos < buf.size() && buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n') pos++; continue; }
        long long v = 0;
        int digits = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9' && digits < 9) { v = v * 10 + (buf[pos] - '0'); pos++; digits++; }
        if (digits == 0 || (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9')) { err = "malformed PGM header"; return false; }
        vals[got++] = v;
    }
    if (pos >= buf.size()) { err = "truncated PGM"; return false; }
    pos++;                                               // the single whitespace byte after maxval
    if (vals[2] < 1 || vals[2] > 255) { err = "PGM maxval must be 1..255"; return false; }
    if (!dims_ok(vals[0], vals[1], 0, 0, err)) return false;
    const size_t all = total_n > buf.size() ? total_n : buf.size();
    if (pos > buf.size() || (unsigned long long)(vals[0] * vals[1]) > all - pos) { err = "truncated PGM"; return false; }
    w = (int)vals[0]; h = (int)vals[1]; data_pos = pos;
    return true;
}

// ---- PNG ---------------------------------------------------------------------------------------------------------------
struct PngInfo { int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0, channels = 0; };

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// chunk walk: IHDR first, every chunk's CRC verified, the IDAT payloads listed in file order (not copied)
struct Span { const uint8_t *p; size_t n; };
bool png_chunks(const uint8_t *buf, size_t size, PngInfo &info, std::vector<Span> &idat, std::string &err)
{
    size_t pos = 8;
    bool have_ihdr = false, have_iend = false;
    while (pos + 12 <= size) {
        const uint32_t len = be32(&buf[pos]);
        if (len > 0x7FFFFFFFu || (size_t)len > size - pos - 12) { err = "PNG chunk runs past the end of the file"; return false; }
        const uint8_t *type = &buf[pos + 4], *data = &buf[pos + 8];
        if ((uint32_t)crc32(crc32(0L, type, 4), data, len) != be32(data + len)) { err = "PNG chunk CRC mismatch"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (have_ihdr || len != 13 || pos != 8) { err = "bad IHDR"; return false; }
            const uint32_t w = be32(data), h = be32(data + 4);
            if (w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) { err = "implausible image size"; return false; }
            info.w = (int)w; info.h = (int)h; info.depth = data[8]; info.ctype = data[9]; info.interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { err = "unknown PNG compression / filter method"; return false; }
            have_ihdr = true;
        } else if (!have_ihdr) { err = "PNG does not start with IHDR"; return false; }
        else if (!memcmp(type, "IDAT", 4)) { if (len) idat.push_back(Span{data, (size_t)len}); }
        else if (!memcmp(type, "IEND", 4)) { have_iend = true; break; }
        else if (!(type[0] & 0x20)) { if (memcmp(type, "PLTE", 4)) { err = "unknown critical PNG chunk"; return false; } }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend || idat.empty()) { err = "incomplete PNG"; return false; }
    info.channels = info.ctype == 0 ? 1 : info.ctype == 2 ? 3 : info.ctype == 4 ? 2 : info.ctype == 6 ? 4 : 0;
    if (!info.channels || (info.depth != 8 && info.depth != 16) || info.interlace > 1) { err = "unsupported PNG pixel format"; return false; }
    return true;
}

// the zlib stream spread over the IDAT chunks, inflated a band of scanlines at a time: a 12-Mpixel plane never exists as
// one filtered buffer (that buffer, freshly mapped per file by dozens of loader threads, cost more in page faults than
// the inflate itself).  The stream must hold exactly the bytes IHDR announces: more or fewer = corrupt; bytes after the
// end of the zlib stream are ignored, as libpng does.
struct Inflater {
    z_stream z;
    const std::vector<Span> &in;
    size_t si = 0;
    bool live = false, ended = false;
    explicit Inflater(const std::vector<Span> &spans) : in(spans) { memset(&z, 0, sizeof z); }
    ~Inflater() { if (live) inflateEnd(&z); }
    bool start() { live = inflateInit(&z) == Z_OK; return live; }
    size_t read(uint8_t *out, size_t n)                      // up to n bytes; fewer only at the end of the stream (or on an error)
    {
        size_t done = 0;
        while (done < n && !ended) {
            if (z.avail_in == 0) {
                if (si == in.size()) break;
                z.next_in = const_cast<Bytef *>(in[si].p);
                z.avail_in = (uInt)in[si].n;                 // (a chunk is < 2^31 bytes: png_chunks)
                si++;
            }
            const size_t ask = n - done < ((size_t)1 << 30) ? n - done : ((size_t)1 << 30);
            z.next_out = out + done; z.avail_out = (uInt)ask;
            const int r = inflate(&z, Z_NO_FLUSH);
            done += ask - z.avail_out;
            if (r == Z_STREAM_END) ended = true;
            else if (r != Z_OK && r != Z_BUF_ERROR) break;
            else if (r == Z_BUF_ERROR && z.avail_in != 0 && z.avail_out != 0) break;
        }
        return done;
    }
    bool finished()                                          // true when not one byte more comes out and the stream closed properly
    {
        uint8_t spare;
        return read(&spare, 1) == 0 && ended;
    }
};

// `rows` scanlines of a (sub-)image pw pixels wide, filtered, at `raw`: undo the filters in place (`prev` = the line above,
// unfiltered, or null on the first line) and grey-convert into dst at (x0 + i*dx, y + j*dy).  Returns the last line, or null when
// a scanline names a filter type the format does not have.
const uint8_t *png_rows(uint8_t *raw, const PngInfo &f, int pw, int rows, const uint8_t *prev, uint8_t *dst, int x0, int y, int dx, int dy)
{
    const int bps = f.depth / 8, bpp = f.channels * bps;
    const size_t stride = (size_t)pw * bpp;
    for (int j = 0; j < rows; j++) {
        uint8_t *line = raw + (stride + 1) * j;
        const int ft = line[0];
        uint8_t *cur = line + 1;
        const size_t head = stride < (size_t)bpp ? stride : (size_t)bpp;
        switch (ft) {
        case 1:
            for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            if (prev) for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
            break;
        case 3:
            for (size_t i = 0; i < head; i++) cur[i] = (uint8_t)(cur[i] + ((prev ? prev[i] : 0) >> 1));
            if (prev) for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
            else for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + (cur[i - bpp] >> 1));
            break;
        case 4:
            for (size_t i = 0; i < head; i++) cur[i] = (uint8_t)(cur[i] + paeth(0, prev ? prev[i] : 0, 0));
            if (prev) for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bpp], prev[i], prev[i - bpp]));
            else for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bpp], 0, 0));
            break;
        case 0: break;
        default: return nullptr;                             // filter types are 0 .. 4 (PNG spec 9.2): a corrupt or hostile file
        }
        uint8_t *out = dst + (size_t)(y + j * dy) * f.w + x0;
        if (bpp == 1 && dx == 1) memcpy(out, cur, (size_t)pw);
        else
            for (int i = 0; i < pw; i++) {
                const uint8_t *p = cur + (size_t)i * bpp;
                out[(size_t)i * dx] = f.channels <= 2 ? p[0] : (uint8_t)((p[0] * 4899 + p[bps] * 9617 + p[2 * bps] * 1868 + 8192) >> 14);
            }
        prev = cur;
    }
    return prev;
}

bool decode_png_into(const uint8_t *buf, size_t size, int want_w, int want_h, uint8_t *dst, std::vector<uint8_t> *own, int &w, int &h,
                     std::string &err)
{
    PngInfo f;
    std::vector<Span> idat;
    if (!png_chunks(buf, size, f, idat, err) || !dims_ok(f.w, f.h, want_w, want_h, err)) return false;
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    const size_t bpp = (size_t)f.channels * (f.depth / 8);
    const char *bad = "PNG image data does not inflate to the size IHDR announces";
    if (own) { own->resize((size_t)f.w * f.h); dst = own->data(); }
    Inflater inf(idat);
    if (!inf.start()) { err = "zlib init failed"; return false; }
    const size_t line = (size_t)f.w * bpp + 1;               // the widest scanline of any pass
    constexpr size_t kBand = (size_t)256 << 10;              // filtered bytes in flight: L2-resident
    const int band_rows = (int)(kBand / line > 0 ? kBand / line : 1);
    std::vector<uint8_t> band(line * (size_t)band_rows), above(line);
    const int passes = f.interlace ? 7 : 1;
    for (int p = 0; p < passes; p++) {
        const int x0 = f.interlace ? ax0[p] : 0, y0 = f.interlace ? ay0[p] : 0, dx = f.interlace ? adx[p] : 1, dy = f.interlace ? ady[p] : 1;
        const int pw = (f.w - x0 + dx - 1) / dx, ph = (f.h - y0 + dy - 1) / dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t pline = (size_t)pw * bpp + 1;
        const uint8_t *prev = nullptr;
        for (int j = 0; j < ph; j += band_rows) {
            const int rows = ph - j < band_rows ? ph - j : band_rows;
            if (inf.read(band.data(), pline * rows) != pline * rows) { err = bad; return false; }
            const uint8_t *last = png_rows(band.data(), f, pw, rows, prev, dst, x0, y0 + j * dy, dx, dy);
            if (!last) { err = "invalid PNG filter type"; return false; }
            memcpy(above.data(), last, pline - 1);           // the band buffer is about to be overwritten
            prev = above.data();
        }
    }
    if (!inf.finished()) { err = bad; return false; }
    w = f.w; h = f.h;
    return true;
}

// a file's bytes, not zero-filled first (std::vector::resize would touch every page twice)
struct FileBytes {
    uint8_t *p = nullptr;
    size_t n = 0;
    ~FileBytes() { free(p); }
};

bool slurp(const std::string &path, FileBytes &out, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n >= 0 && (size_t)n <= kMaxFileBytes && fseek(f, 0, SEEK_SET) == 0) {
            out.p = (uint8_t *)malloc((size_t)n + 1);
            out.n = (size_t)n;
            ok = out.p && (n == 0 || fread(out.p, 1, (size_t)n, f) == (size_t)n);
        }
    }
    fclose(f);
    if (!ok) err = "cannot read " + path;
    return ok;
}

// A binary PGM whose pixels go to a caller-owned buffer (the series loader's page-locked slots): the header from the file's first
// 4 KB, the pixels by ONE fread straight into dst -- no staging copy of the whole file (round 4: the staging malloc's page faults and
// the second copy were half of a PGM scan's host time).  false + empty err: not that case, the caller takes the general path.
static bool pgm_read_into(const std::string &path, int want_w, int want_h, uint8_t *dst, int &w, int &h, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    uint8_t head[4096];
    const size_t got = fread(head, 1, sizeof head, f);
    bool ok = false;
    if (got >= 3 && head[0] == 'P' && head[1] == '5') {
        size_t pos = 0;
        long fsize = -1;
        if (fseek(f, 0, SEEK_END) == 0) fsize = ftell(f);
        std::string herr;
        int hw = 0, hh = 0;
        bool hdr = false;
        hdr = fsize >= 0 && pgm_header(head, got, hw, hh, pos, herr, (size_t)fsize) && pos <= got;   // (parses inside the 4 KB head only)
        if (!hdr) { fclose(f); return false; }           // (a header beyond the first 4 KB, or a bad one: the general path decides)
        if (!dims_ok(hw, hh, want_w, want_h, err)) { fclose(f); return false; }
        const size_t npx = (size_t)hw * hh, inhead = got - pos < npx ? got - pos : npx;
        memcpy(dst, head + pos, inhead);
        ok = fseek(f, (long)(pos + inhead), SEEK_SET) == 0 && (npx == inhead || fread(dst + inhead, 1, npx - inhead, f) == npx - inhead);
        if (!ok) err = "truncated PGM";
        w = hw; h = hh;
    }
    fclose(f);
    return ok;
}

bool decode_any(const std::string &path, int want_w, int want_h, uint8_t *dst, std::vector<uint8_t> *own, int &w, int &h, std::string &err)
{
    if (dst && !own) {                                   // (a PGM into a caller-owned buffer: no staging copy)
        std::string e2;
        if (pgm_read_into(path, want_w, want_h, dst, w, h, e2)) return true;
        if (!e2.empty()) { err = e2; return false; }
    }
    FileBytes buf;
    if (!slurp(path, buf, err)) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.n >= 8 && !memcmp(buf.p, sig, 8)) return decode_png_into(buf.p, buf.n, want_w, want_h, dst, own, w, h, err);
    if (buf.n >= 3 && buf.p[0] == 'P' && buf.p[1] == '5') {
        size_t pos = 0;
        if (!pgm_header(buf.p, buf.n, w, h, pos, err) || !dims_ok(w, h, want_w, want_h, err)) return false;
        if (own) { own->assign(buf.p + pos, buf.p + pos + (size_t)w * h); }
        else memcpy(dst, buf.p + pos, (size_t)w * h);
        return true;
    }
    err = "not a PNG or binary PGM: " + path;
    return false;
}

void png_chunk(std::ofstream &f, const char *type, const std::vector<uint8_t> &data)
{
    const uint32_t n = (uint32_t)data.size();
    const uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
    f.write((const char *)len, 4);
    f.write(type, 4);
    if (n) f.write((const char *)data.data(), (std::streamsize)n);
    uLong crc = crc32(0L, (const Bytef *)type, 4);
    if (n) crc = crc32(crc, data.data(), n);
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    f.write((const char *)c, 4);
}

}  // namespace

Image8 imread_gray(const std::string &path)
{
    Image8 img;
    std::string err;
    if (!decode_any(path, 0, 0, nullptr, &img.d, img.w, img.h, err)) return Image8();
    return img;
}

bool imread_gray_into(const std::string &path, int w, int h, uint8_t *dst, std::string &err)
{
    int gw = 0, gh = 0;
    return decode_any(path, w, h, dst, nullptr, gw, gh, err);
}

bool imwrite_pgm(const std::string &path, const Image8 &img)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    f << "P5\n" << img.w << " " << img.h << "\n255\n";
    f.write((const char *)img.d.data(), (std::streamsize)img.d.size());
    return (bool)f;
}

bool imwrite_png(const std::string &path, const Image8 &img, bool adam7)
{
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f || img.w <= 0 || img.h <= 0) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write((const char *)sig, 8);
    std::vector<uint8_t> ihdr(13, 0);
    for (int i = 0; i < 4; i++) { ihdr[i] = (uint8_t)(img.w >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(img.h >> (24 - 8 * i)); }
    ihdr[8] = 8;                                             // depth 8, colour type 0 (grey)
    ihdr[12] = adam7 ? 1 : 0;
    png_chunk(f, "IHDR", ihdr);
    std::vector<uint8_t> raw;
    if (adam7) {                                             // test images only: filter type 0 everywhere
        static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; p++)
            for (int y = ay0[p]; y < img.h; y += ady[p]) {
                if (ax0[p] >= img.w) break;
                raw.push_back(0);
                for (int x = ax0[p]; x < img.w; x += adx[p]) raw.push_back(img.d[(size_t)y * img.w + x]);
            }
    } else {
        raw.resize((size_t)(img.w + 1) * img.h);
        for (int y = 0; y < img.h; y++) {
            raw[(size_t)(img.w + 1) * y] = 0;
            memcpy(&raw[(size_t)(img.w + 1) * y + 1], &img.d[(size_t)img.w * y], img.w);
        }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    comp.resize(clen);
    png_chunk(f, "IDAT", comp);
    png_chunk(f, "IEND", std::vector<uint8_t>());
    return (bool)f;
}

bool exportMat(const std::string &path, const double *m, int rows, int cols)
{
    std::ofstream out(path.c_str());
    if (!out) return false;
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) out << m[r * cols + c] << "\t";
        out << "\n";
    }
    return true;
}

}  // namespace duke

// ImageIO.cpp -- see ImageIO.h.  Header grammar of the reference's read_header
// (ImageIO.cpp:301-330): magic, white space, optional '#' comment lines, width, height,
// maxval, then exactly one newline (optionally preceded by CR).
#include "ImageIO.h"
#include "Error.h"
#include <cstdio>
#include <cstring>
#include <string>

namespace {

struct File {
    FILE *f;
    File(const char *name, const char *mode) : f(fopen(name, mode)) {}
    ~File() { if (f) fclose(f); }
};

int next_int(FILE *f, const char *filename)
{
    int c = getc(f);
    for (;;) {
        while (c == ' ' || c == '\t' || c == '\r' || c == '\n') c = getc(f);
        if (c != '#') break;
        while (c != '\n' && c != EOF) c = getc(f);
    }
    if (c < '0' || c > '9')
        throw CError("ReadImage(%s): malformed header", filename);
    long long v = 0;
    while (c >= '0' && c <= '9') {
        v = v * 10 + (c - '0');
        if (v > 0x7fffffffLL)
            throw CError("ReadImage(%s): header value out of range", filename);
        c = getc(f);
    }
    ungetc(c, f);
    return (int)v;
}

// width x height x bytes per pixel must be positive and small enough for one allocation: a corrupt
// header must not turn into a huge or negative size
void check_dims(int w, int h, int b, size_t elem, const char *filename)
{
    const unsigned long long cap = 1ULL << 34;     // 16 GiB
    if (w <= 0 || h <= 0 || b <= 0 ||
        (unsigned long long)w * (unsigned long long)b * elem > cap ||
        (unsigned long long)w * (unsigned long long)b * elem * (unsigned long long)h > cap)
        throw CError("ReadImage(%s): invalid image size %d x %d x %d", filename, w, h, b);
}

bool ends_with(const char *s, const char *ext)
{
    size_t n = strlen(s), m = strlen(ext);
    return n >= m && strcmp(s + n - m, ext) == 0;
}

}  // namespace

void ReadImage(CByteImage &img, const char *filename)
{
    bool gray = ends_with(filename, ".pgm"), color = ends_with(filename, ".ppm");
    if (!gray && !color)
        throw CError("ReadImage(%s): file type not supported (only .pgm and .ppm)", filename);
    File in(filename, "rb");
    if (!in.f)
        throw CError("ReadImage: could not open %s", filename);
    if (getc(in.f) != 'P' || getc(in.f) != (gray ? '5' : '6'))
        throw CError("ReadImage(%s): wrong magic code", filename);
    int w = next_int(in.f, filename), h = next_int(in.f, filename);
    int maxval = next_int(in.f, filename);
    check_dims(w, h, gray ? 1 : 4, 1, filename);
    if (maxval < 1 || maxval > 255)                  // one byte per sample is all the path handles
        throw CError("ReadImage(%s): maxval %d not supported (1..255)", filename, maxval);
    int c = getc(in.f);
    if (c == '\r') c = getc(in.f);
    if (c != '\n')
        throw CError("ReadImage(%s): newline expected after the header", filename);
    img.ReAllocate(CShape(w, h, gray ? 1 : 4));
    std::vector<uchar> row((size_t)w * 3);
    for (int y = 0; y < h; y++) {
        uchar *dst = img.Row(y);
        if (gray) {
            if (fread(dst, 1, (size_t)w, in.f) != (size_t)w)
                throw CError("ReadImage(%s): file is too short", filename);
        } else {
            if (fread(row.data(), 1, row.size(), in.f) != row.size())
                throw CError("ReadImage(%s): file is too short", filename);
            for (int x = 0; x < w; x++) {
                dst[4 * x + 0] = row[3 * x + 2];
                dst[4 * x + 1] = row[3 * x + 1];
                dst[4 * x + 2] = row[3 * x + 0];
                dst[4 * x + 3] = 255;
            }
        }
    }
}

void WriteImage(const CByteImage &img, const char *filename)
{
    CShape sh = img.Shape();
    if (sh.nBands != 1 && sh.nBands != 4)
        throw CError("WriteImage(%s): unhandled # of bands %d", filename, sh.nBands);
    File out(filename, "wb");
    if (!out.f)
        throw CError("WriteImage: could not open %s", filename);
    fprintf(out.f, "P%d\n%d %d\n%d\n", sh.nBands == 1 ? 5 : 6, sh.width, sh.height, 255);
    std::vector<uchar> row((size_t)sh.width * 3);
    for (int y = 0; y < sh.height; y++) {
        const uchar *src = img.Row(y);
        if (sh.nBands == 1) {
            fwrite(src, 1, (size_t)sh.width, out.f);
        } else {
            for (int x = 0; x < sh.width; x++) {
                row[3 * x + 0] = src[4 * x + 2];
                row[3 * x + 1] = src[4 * x + 1];
                row[3 * x + 2] = src[4 * x + 0];
            }
            fwrite(row.data(), 1, row.size(), out.f);
        }
    }
}

void ReadImage(CFloatImage &img, const char *filename)
{
    if (!ends_with(filename, ".pmf"))
        throw CError("ReadImage(%s): a float image is read from a .pmf file", filename);
    File in(filename, "rb");
    if (!in.f)
        throw CError("ReadImage: could not open %s", filename);
    if (getc(in.f) != 'P' || getc(in.f) != '9')
        throw CError("ReadImage(%s): wrong magic code for PMF file", filename);
    int w = next_int(in.f, filename), h = next_int(in.f, filename), b = next_int(in.f, filename);
    int c = getc(in.f);
    if (c == '\r') c = getc(in.f);
    if (c != '\n')
        throw CError("ReadImage(%s): newline expected after the header", filename);
    check_dims(w, h, b, sizeof(float), filename);
    img.ReAllocate(CShape(w, h, b));
    size_t n = (size_t)w * b * sizeof(float);
    for (int y = 0; y < h; y++)
        if (fread(img.Row(y), 1, n, in.f) != n)
            throw CError("ReadImage(%s): file is too short", filename);
}

void WriteImage(const CFloatImage &img, const char *filename)
{
    if (!ends_with(filename, ".pmf"))
        throw CError("WriteImage(%s): can only write floating point image as pmf", filename);
    CShape sh = img.Shape();
    File out(filename, "wb");
    if (!out.f)
        throw CError("WriteImage: could not open %s", filename);
    fprintf(out.f, "P9\n%d %d\n%d\n", sh.width, sh.height, sh.nBands);
    size_t n = (size_t)sh.width * sh.nBands * sizeof(float);
    for (int y = 0; y < sh.height; y++)
        if (fwrite(img.Row(y), 1, n, out.f) != n)
            throw CError("WriteImage(%s): write failed", filename);
}

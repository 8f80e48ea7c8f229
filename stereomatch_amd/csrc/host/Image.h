// Image.h -- minimal host image container with the reference's memory layout contract
// (Image.h:160-164, Image.cpp:97-98): pixel (x, y, band) at y*rowBytes + (x*nBands + band)*sizeof(T),
// rows rounded up to 8 bytes.  Only what the hot-path driver needs; no sub-images, no
// reference counting (images are owned by value).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

typedef unsigned char uchar;

struct CShape {
    int width = 0, height = 0, nBands = 0;
    CShape() {}
    CShape(int w, int h, int b) : width(w), height(h), nBands(b) {}
    bool operator==(const CShape &o) const { return width == o.width && height == o.height && nBands == o.nBands; }
    bool operator!=(const CShape &o) const { return !(*this == o); }
};

template <class T>
class CImageOf {
public:
    CImageOf() {}
    explicit CImageOf(CShape s) { ReAllocate(s); }
    CImageOf(int w, int h, int b) { ReAllocate(CShape(w, h, b)); }

    void ReAllocate(CShape s)
    {
        m_shape = s;
        m_rowBytes = (sizeof(T) * (size_t)s.nBands * s.width + 7) & ~(size_t)7;
        m_mem.assign(m_rowBytes * (size_t)s.height, 0);
    }
    void ClearPixels() { std::fill(m_mem.begin(), m_mem.end(), 0); }
    CShape Shape() const { return m_shape; }
    size_t RowBytes() const { return m_rowBytes; }
    size_t ImageSize() const { return m_mem.size(); }
    T *Row(int y) { return reinterpret_cast<T *>(m_mem.data() + m_rowBytes * (size_t)y); }
    const T *Row(int y) const { return reinterpret_cast<const T *>(m_mem.data() + m_rowBytes * (size_t)y); }
    T &Pixel(int x, int y, int band) { return Row(y)[(size_t)x * m_shape.nBands + band]; }
    const T &Pixel(int x, int y, int band) const { return Row(y)[(size_t)x * m_shape.nBands + band]; }
    void *Data() { return m_mem.data(); }
    const void *Data() const { return m_mem.data(); }

private:
    CShape m_shape;
    size_t m_rowBytes = 0;
    std::vector<uint8_t> m_mem;
};

typedef CImageOf<uchar> CByteImage;
typedef CImageOf<int> CIntImage;
typedef CImageOf<float> CFloatImage;

// ImageIO.h -- PGM (P5) / PPM (P6) / PMF (P9, multi-band float) files with the reference's band
// conventions.
#pragma once
#include "Image.h"

// .pgm -> 1 band; .ppm -> 4 bands stored B,G,R,A with A = 255 (reference ImageIO.cpp:378-398)
void ReadImage(CByteImage &img, const char *filename);
// 1 band -> "P5\n%d %d\n255\n"; 4 bands -> P6 written R,G,B (reference ImageIO.cpp:428-478)
void WriteImage(const CByteImage &img, const char *filename);
// .pmf: "P9\n%d %d\n%d\n" (width, height, bands) + rows of native floats (reference ImageIO.cpp:352-376,
// :428-441) -- the reference's lossless container for a DSI
void ReadImage(CFloatImage &img, const char *filename);
void WriteImage(const CFloatImage &img, const char *filename);

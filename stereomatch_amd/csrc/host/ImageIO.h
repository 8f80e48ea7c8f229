// ImageIO.h -- PGM (P5) / PPM (P6) files with the reference's band conventions.
#pragma once
#include "Image.h"

// .pgm -> 1 band; .ppm -> 4 bands stored B,G,R,A with A = 255 (reference ImageIO.cpp:378-398)
void ReadImage(CByteImage &img, const char *filename);
// 1 band -> "P5\n%d %d\n255\n"; 4 bands -> P6 written R,G,B (reference ImageIO.cpp:428-478)
void WriteImage(const CByteImage &img, const char *filename);

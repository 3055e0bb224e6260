// matrixIO3D.h -- raw 3-D matrix files of the reference (Include/Util/matrixIO3D.h:21-64): a 12-byte
// header (int32 m, n, p = x, y, z sizes) followed by m*n*p fp32 values, x fastest.
#pragma once
#include "common.h"

// *volume is malloc()ed (the reference's factory free()s it, Src/cSIFT3D.cc:123). Returns 0 on success.
SIFT_LIBRARY_API int ReadMatrixFromDisk(const char *filename, int *m, int *n, int *p, float **volume);
SIFT_LIBRARY_API int WriteMatrixToDisk(const char *filename, int m, int n, int p, const float *volume);

// matrixIO3D.h -- raw 3-D matrix files of the reference (Include/Util/matrixIO3D.h:15-140): a 12-byte
// header (int32 m, n, p = x, y, z sizes) followed by m*n*p values of the element type, x fastest.  Same
// entry points and return convention as the reference (0 = success, non-zero = failure): the header
// helpers are functions, the payload routines templates on the element type; the float forms are also
// exported as ordinary functions (what CreateCSIFT3D(std::string) and the tests call).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "common.h"

SIFT_LIBRARY_API int ReadMatrixSizeFromStream(FILE *file, int *m, int *n, int *p);
SIFT_LIBRARY_API int ReadMatrixSizeFromDisk(const char *filename, int *m, int *n, int *p);
SIFT_LIBRARY_API int WriteMatrixHeaderToStream(FILE *file, int m, int n, int p);

// *volume is malloc()ed (the reference's factory free()s it, Src/cSIFT3D.cc:123)
SIFT_LIBRARY_API int ReadMatrixFromDisk(const char *filename, int *m, int *n, int *p, float **volume);
SIFT_LIBRARY_API int WriteMatrixToDisk(const char *filename, int m, int n, int p, const float *volume);

template <class T>
int ReadMatrixFromStream(FILE *file, int M, int N, int P, T *matrix) {
	const size_t cnt = (size_t)M * (size_t)N * (size_t)P;
	return fread(matrix, sizeof(T), cnt, file) == cnt ? 0 : 1;
}

template <class T>
int WriteMatrixToStream(FILE *file, size_t m, size_t n, size_t p, T *matrix) {
	return fwrite(matrix, sizeof(T), m * n * p, file) == m * n * p ? 0 : 1;
}

template <class T>
int ReadMatrixFromDisk(const char *filename, int *m, int *n, int *p, T **matrix) {
	FILE *f = fopen(filename, "rb");
	if (!f) { printf("Can't open input matrix file: %s.\n", filename); return 1; }
	int rc = ReadMatrixSizeFromStream(f, m, n, p);
	if (rc == 0 && (*m <= 0 || *n <= 0 || *p <= 0)) rc = 1;
	if (rc == 0) {
		*matrix = (T *)malloc(sizeof(T) * (size_t)*m * (size_t)*n * (size_t)*p);
		rc = *matrix ? ReadMatrixFromStream(f, *m, *n, *p, *matrix) : 1;
		if (rc) { free(*matrix); *matrix = nullptr; }
	}
	if (rc) printf("Error reading matrix from disk file: %s.\n", filename);
	fclose(f);
	return rc ? 1 : 0;
}

template <class T>
int WriteMatrixToDisk(const char *filename, int m, int n, int p, T *matrix) {
	FILE *f = fopen(filename, "wb");
	if (!f) { printf("Can't open output file: %s.\n", filename); return 1; }
	const int rc = WriteMatrixHeaderToStream(f, m, n, p) || WriteMatrixToStream(f, (size_t)m, (size_t)n, (size_t)p, matrix);
	if (rc) printf("Error writing the matrix to disk file: %s.\n", filename);
	fclose(f);
	return rc ? 1 : 0;
}

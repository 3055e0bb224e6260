// readNii.h -- NIfTI loader with the reference's signature (Include/Util/readNii.h:6).
#pragma once
#include "common.h"

// Reads a NIfTI-1 / NIfTI-2 volume -- single file (.nii) or .hdr + .img pair named by either file, ANALYZE 7.5 pairs, each optionally
// gzip-compressed -- and returns it as a new float[] (caller
// delete[]s), converting the stored datatype to fp32 WITHOUT applying scl_slope / scl_inter, like
// the reference (Src/Util/readNii.cpp:17-33).  Returns nullptr on failure.
SIFT_LIBRARY_API float *readNiiFile(const char *filename, int &nx, int &ny, int &nz);

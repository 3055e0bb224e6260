// io.cpp -- host-side volume loaders + timer printing of the drop-in shell:
//   readNiiFile          reference Src/Util/readNii.cpp:5-39 (NIfTI-1 single file, optional gzip)
//   Read/WriteMatrix...  reference Include/Util/matrixIO3D.h:21-64, Src/Util/matrixIO3D.cpp:7-29
//   operator<<           reference Src/Util/common.cpp:5-36
// Own minimal implementations (the reference vendors the 11 kLoC layNii/nifti2 reader instead).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../Include/Util/common.h"
#include "../Include/Util/matrixIO3D.h"
#include "../Include/Util/readNii.h"

namespace {

template <typename T>
T bswap(T v) {
	unsigned char *b = reinterpret_cast<unsigned char *>(&v);
	for (size_t i = 0; i < sizeof(T) / 2; i++) std::swap(b[i], b[sizeof(T) - 1 - i]);
	return v;
}

template <typename T>
void convert(const unsigned char *raw, size_t n, bool swap, float *out) {
	for (size_t i = 0; i < n; i++) {
		T v;
		memcpy(&v, raw + i * sizeof(T), sizeof(T));
		if (swap) v = bswap(v);
		out[i] = (float)v;
	}
}

}  // namespace

float *readNiiFile(const char *filename, int &nx, int &ny, int &nz) {
	nx = ny = nz = 0;
	gzFile f = gzopen(filename, "rb");  // transparently reads plain and gzip-compressed files
	if (!f) { fprintf(stderr, "readNiiFile: cannot open %s\n", filename); return nullptr; }
	unsigned char hdr[352];
	if (gzread(f, hdr, 348) != 348) { gzclose(f); fprintf(stderr, "readNiiFile: short header\n"); return nullptr; }
	int32_t sizeof_hdr;
	memcpy(&sizeof_hdr, hdr, 4);
	bool swap = false;
	if (sizeof_hdr != 348) {
		if (bswap(sizeof_hdr) == 348) swap = true;
		else { gzclose(f); fprintf(stderr, "readNiiFile: not a NIfTI-1 file (sizeof_hdr=%d)\n", sizeof_hdr); return nullptr; }
	}
	int16_t dim[8], datatype, bitpix;
	memcpy(dim, hdr + 40, 16);
	memcpy(&datatype, hdr + 70, 2);
	memcpy(&bitpix, hdr + 72, 2);
	float vox_offset;
	memcpy(&vox_offset, hdr + 108, 4);
	if (swap) {
		for (auto &d : dim) d = bswap(d);
		datatype = bswap(datatype); bitpix = bswap(bitpix); vox_offset = bswap(vox_offset);
	}
	if (memcmp(hdr + 344, "n+1", 3) != 0) { gzclose(f); fprintf(stderr, "readNiiFile: only single-file NIfTI-1 (n+1) is supported\n"); return nullptr; }
	// the header is untrusted input: dimensions must be positive and the element size must agree with the datatype code.  A payload
	// offset below the header size is read as 348, like the reference's reader does for single-file images (nifti2_io.cpp:5187-5189
	// "set ioff from vox_offset (but at least sizeof(header))"): lax writers leave vox_offset at 0 or 348; NaN and absurd values are
	// rejected
	size_t esize = 0;
	switch (datatype) {
	case 2: case 256: esize = 1; break;
	case 4: case 512: esize = 2; break;
	case 8: case 16: case 768: esize = 4; break;
	case 64: esize = 8; break;
	default: break;
	}
	const int ndim = dim[0];
	if (ndim < 1 || ndim > 7 || dim[1] <= 0 || (ndim >= 2 && dim[2] <= 0) || (ndim >= 3 && dim[3] <= 0) || esize == 0 ||
	    bitpix != (int16_t)(8 * esize) || !(vox_offset == vox_offset) || vox_offset > 1.0e9f || vox_offset < -1.0e9f) {
		gzclose(f);
		fprintf(stderr, "readNiiFile: bad or unsupported header (dim %d: %d %d %d, datatype %d, bitpix %d, vox_offset %g)\n", ndim,
		        (int)dim[1], (int)dim[2], (int)dim[3], (int)datatype, (int)bitpix, (double)vox_offset);
		nx = ny = nz = 0;
		return nullptr;
	}
	nx = dim[1]; ny = ndim >= 2 ? dim[2] : 1; nz = ndim >= 3 ? dim[3] : 1;
	const size_t n = (size_t)nx * ny * nz, bytes = n * esize;
	if (vox_offset < 348.0f) vox_offset = 348.0f;
	long skip = (long)vox_offset - 348;
	for (unsigned char junk[4096]; skip > 0;) {  // header extensions between the header and the payload
		const int want = (int)std::min<long>(skip, (long)sizeof(junk));
		if (gzread(f, junk, (unsigned)want) != want) { gzclose(f); fprintf(stderr, "readNiiFile: truncated before the payload\n"); nx = ny = nz = 0; return nullptr; }
		skip -= want;
	}
	// the payload is read in bounded pieces into a buffer that grows with the data that actually arrives: a header that claims
	// terabytes on a short file ends as "truncated payload", not as an allocation of the claimed size
	std::vector<unsigned char> raw;
	size_t got = 0;
	try {
		while (got < bytes) {
			const size_t piece = std::min<size_t>(bytes - got, (size_t)64 << 20);
			raw.resize(got + piece);
			int r = gzread(f, raw.data() + got, (unsigned)piece);
			if (r <= 0) break;
			got += (size_t)r;
			if ((size_t)r < piece) {  // short read: EOF unless more follows
				raw.resize(got);
				continue;
			}
		}
	} catch (...) { gzclose(f); fprintf(stderr, "readNiiFile: out of memory\n"); nx = ny = nz = 0; return nullptr; }
	gzclose(f);
	if (got != bytes) { fprintf(stderr, "readNiiFile: truncated payload\n"); nx = ny = nz = 0; return nullptr; }
	float *out = new (std::nothrow) float[n];
	if (!out) { fprintf(stderr, "readNiiFile: out of memory\n"); nx = ny = nz = 0; return nullptr; }
	switch (datatype) {  // NIfTI datatype codes; slope/intercept deliberately ignored (see readNii.h)
	case 2: convert<uint8_t>(raw.data(), n, false, out); break;
	case 4: convert<int16_t>(raw.data(), n, swap, out); break;
	case 8: convert<int32_t>(raw.data(), n, swap, out); break;
	case 16: convert<float>(raw.data(), n, swap, out); break;
	case 64: convert<double>(raw.data(), n, swap, out); break;
	case 256: convert<int8_t>(raw.data(), n, false, out); break;
	case 512: convert<uint16_t>(raw.data(), n, swap, out); break;
	case 768: convert<uint32_t>(raw.data(), n, swap, out); break;
	default:
		fprintf(stderr, "readNiiFile: unsupported datatype %d\n", (int)datatype);
		delete[] out;
		nx = ny = nz = 0;
		return nullptr;
	}
	return out;
}

// ---- raw matrix files (reference Include/Util/matrixIO3D.h:15-140, Src/Util/matrixIO3D.cpp): 0 = success, 1 = failure ----
int ReadMatrixSizeFromStream(FILE *file, int *m, int *n, int *p) {
	int32_t h[3];
	if (!file || fread(h, sizeof(int32_t), 3, file) != 3) return 1;
	*m = h[0]; *n = h[1]; *p = h[2];
	return 0;
}

int ReadMatrixSizeFromDisk(const char *filename, int *m, int *n, int *p) {
	FILE *f = fopen(filename, "rb");
	if (!f) return 1;
	const int rc = ReadMatrixSizeFromStream(f, m, n, p);
	fclose(f);
	return rc;
}

int WriteMatrixHeaderToStream(FILE *file, int m, int n, int p) {
	const int32_t h[3] = {m, n, p};
	return (file && fwrite(h, sizeof(int32_t), 3, file) == 3) ? 0 : 1;
}

int ReadMatrixFromDisk(const char *filename, int *m, int *n, int *p, float **volume) { return ReadMatrixFromDisk<float>(filename, m, n, p, volume); }

int WriteMatrixToDisk(const char *filename, int m, int n, int p, const float *volume) { return WriteMatrixToDisk<const float>(filename, m, n, p, volume); }

std::ostream &operator<<(std::ostream &os, const SIFT_TimerPara &st) {
	os << "3D SIFT timing (s): total " << st.d_TotalTime << " | GSS+DoG " << (st.d_BuildGSS + st.d_BuildDOG) << " | detect "
	   << st.d_Detect << " | orientation " << st.d_AssignOrientation << " | description " << st.d_Extraction << "\n";
	return os;
}

// ---- keypoint coordinate lists (reference Src/cUtil.cc:938-954 write_sift_kp, 1002-1016 read_sift_kp) ----
#include "../Include/cUtil.h"
namespace CPUSIFT {
void write_sift_kp(std::vector<Cvec> &kp, const char *file_name) {
	FILE *f = fopen(file_name, "w");
	if (!f) { fprintf(stderr, "write_sift_kp: cannot open %s\n", file_name); return; }
	for (const Cvec &c : kp) fprintf(f, "%.5lf,%.5lf,%.5lf\n", (double)c.x, (double)c.y, (double)c.z);
	fclose(f);
}
void read_sift_kp(const char *file_name, std::vector<Cvec> &kp) {
	FILE *f = fopen(file_name, "r");
	if (!f) { fprintf(stderr, "read_sift_kp: cannot open %s\n", file_name); return; }
	double x, y, z;
	while (fscanf(f, " %lf , %lf , %lf", &x, &y, &z) == 3) kp.push_back(Cvec((float)x, (float)y, (float)z));
	fclose(f);
}
}  // namespace CPUSIFT

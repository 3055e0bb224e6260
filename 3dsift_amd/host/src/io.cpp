// io.cpp -- host-side volume loaders + timer printing of the drop-in shell:
//   readNiiFile          reference Src/Util/readNii.cpp:5-39 (NIfTI-1 / NIfTI-2, single file or .hdr + .img pair, ANALYZE 7.5 pairs, optional gzip)
//   Read/WriteMatrix...  reference Include/Util/matrixIO3D.h:21-64, Src/Util/matrixIO3D.cpp:7-29
//   operator<<           reference Src/Util/common.cpp:5-36
// Own minimal implementations (the reference vendors the 11 kLoC layNii/nifti2 reader instead).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
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

// What a header says about the payload, whatever its version.  NIfTI-1: 348 bytes, int16 dim[8] at 40, datatype / bitpix at 70 / 72, float
// vox_offset at 108, magic at 344.  NIfTI-2: 540 bytes, magic at 4, datatype / bitpix at 12 / 14, int64 dim[8] at 16, int64 vox_offset at
// 168.  ANALYZE 7.5 (no magic, .hdr + .img only) shares the NIfTI-1 offsets.  Single file: magic "n+1" / "n+2"; pair: "ni1" / "ni2" / none.
struct NiiHeader {
	int version = 1;       // 1: NIfTI-1 / ANALYZE, 2: NIfTI-2
	bool swap = false, onefile = true;
	long long dim[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int datatype = 0, bitpix = 0;
	double vox_offset = 0;
	size_t size = 348;     // bytes of the header proper
};

// reads and decodes the header at the start of f; false + message on anything that is not one
bool read_nii_header(gzFile f, NiiHeader &H, const char **why) {
	unsigned char hdr[540];
	if (gzread(f, hdr, 4) != 4) { *why = "short header"; return false; }
	int32_t sizeof_hdr;
	memcpy(&sizeof_hdr, hdr, 4);
	if (sizeof_hdr == 348 || sizeof_hdr == 540) H.swap = false;
	else if (bswap(sizeof_hdr) == 348 || bswap(sizeof_hdr) == 540) { H.swap = true; sizeof_hdr = bswap(sizeof_hdr); }
	else { *why = "not a NIfTI-1 / NIfTI-2 / ANALYZE header (sizeof_hdr)"; return false; }
	H.version = sizeof_hdr == 540 ? 2 : 1;
	H.size = (size_t)sizeof_hdr;
	if (gzread(f, hdr + 4, (unsigned)(H.size - 4)) != (int)(H.size - 4)) { *why = "short header"; return false; }
	const unsigned char *magic = hdr + (H.version == 2 ? 4 : 344);
	if (H.version == 1) {
		int16_t dim[8], datatype, bitpix;
		float vox_offset;
		memcpy(dim, hdr + 40, 16); memcpy(&datatype, hdr + 70, 2); memcpy(&bitpix, hdr + 72, 2); memcpy(&vox_offset, hdr + 108, 4);
		if (H.swap) { for (auto &d : dim) d = bswap(d); datatype = bswap(datatype); bitpix = bswap(bitpix); vox_offset = bswap(vox_offset); }
		for (int i = 0; i < 8; i++) H.dim[i] = dim[i];
		H.datatype = datatype; H.bitpix = bitpix; H.vox_offset = (double)vox_offset;
		if (memcmp(magic, "n+1", 3) == 0) H.onefile = true;
		else if (memcmp(magic, "ni1", 3) == 0 || magic[0] == 0) H.onefile = false;  // pair; no magic at all: ANALYZE 7.5
		else { *why = "unknown NIfTI-1 magic"; return false; }
	} else {
		int64_t dim[8], vox_offset;
		int16_t datatype, bitpix;
		memcpy(&datatype, hdr + 12, 2); memcpy(&bitpix, hdr + 14, 2); memcpy(dim, hdr + 16, 64); memcpy(&vox_offset, hdr + 168, 8);
		if (H.swap) { for (auto &d : dim) d = bswap(d); datatype = bswap(datatype); bitpix = bswap(bitpix); vox_offset = bswap(vox_offset); }
		for (int i = 0; i < 8; i++) H.dim[i] = dim[i];
		H.datatype = datatype; H.bitpix = bitpix; H.vox_offset = (double)vox_offset;
		if (memcmp(magic, "n+2", 3) == 0) H.onefile = true;
		else if (memcmp(magic, "ni2", 3) == 0) H.onefile = false;
		else { *why = "unknown NIfTI-2 magic"; return false; }
	}
	return true;
}

bool ends_with(const std::string &s, const char *suffix) {
	const size_t n = strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}
bool readable(const std::string &p) {
	FILE *f = fopen(p.c_str(), "rb");
	if (f) fclose(f);
	return f != nullptr;
}
// x.hdr[.gz] <-> x.img[.gz]: the sibling with the same compression if it exists, else with the other
std::string sibling(const std::string &name, const char *from, const char *to) {
	std::string base = name;
	const bool gz = ends_with(base, ".gz");
	if (gz) base.resize(base.size() - 3);
	if (!ends_with(base, from)) return std::string();
	base.resize(base.size() - strlen(from));
	const std::string same = base + to + (gz ? ".gz" : ""), other = base + to + (gz ? "" : ".gz");
	return readable(same) ? same : (readable(other) ? other : same);
}

}  // namespace

float *readNiiFile(const char *filename, int &nx, int &ny, int &nz) {
	nx = ny = nz = 0;
	// a pair may be named by either file (the reference's reader resolves both ways, nifti_findhdrname / nifti_findimgname)
	std::string hdr_name = filename ? filename : "", img_name;
	{
		const std::string h = sibling(hdr_name, ".img", ".hdr");
		if (!h.empty()) { img_name = hdr_name; hdr_name = h; }
	}
	gzFile f = gzopen(hdr_name.c_str(), "rb");  // transparently reads plain and gzip-compressed files
	if (!f) { fprintf(stderr, "readNiiFile: cannot open %s\n", hdr_name.c_str()); return nullptr; }
	NiiHeader H;
	const char *why = "";
	if (!read_nii_header(f, H, &why)) { gzclose(f); fprintf(stderr, "readNiiFile: %s: %s\n", hdr_name.c_str(), why); return nullptr; }
	// the header is untrusted input: dimensions must be positive and the element size must agree with the datatype code.  A payload
	// offset below the header size of a single-file image is read as the header size, like the reference's reader does
	// (nifti2_io.cpp:4917-4923, 5185-5189 "set ioff from vox_offset (but at least sizeof(header))"): lax writers leave vox_offset at 0 or
	// 348; NaN and absurd values are rejected
	size_t esize = 0;
	switch (H.datatype) {
	case 2: case 256: esize = 1; break;
	case 4: case 512: esize = 2; break;
	case 8: case 16: case 768: esize = 4; break;
	case 64: esize = 8; break;
	default: break;
	}
	const long long ndim = H.dim[0];
	const long long kMaxDim = 1 << 20;
	if (ndim < 1 || ndim > 7 || H.dim[1] <= 0 || H.dim[1] > kMaxDim || (ndim >= 2 && (H.dim[2] <= 0 || H.dim[2] > kMaxDim)) ||
	    (ndim >= 3 && (H.dim[3] <= 0 || H.dim[3] > kMaxDim)) || esize == 0 || H.bitpix != (int)(8 * esize) || !(H.vox_offset == H.vox_offset) ||
	    H.vox_offset > 1.0e9 || H.vox_offset < -1.0e9) {
		gzclose(f);
		fprintf(stderr, "readNiiFile: bad or unsupported header (dim %lld: %lld %lld %lld, datatype %d, bitpix %d, vox_offset %g)\n", ndim, H.dim[1],
		        H.dim[2], H.dim[3], H.datatype, H.bitpix, H.vox_offset);
		return nullptr;
	}
	const int dx = (int)H.dim[1], dy = ndim >= 2 ? (int)H.dim[2] : 1, dz = ndim >= 3 ? (int)H.dim[3] : 1;
	const size_t n = (size_t)dx * dy * dz, bytes = n * esize;
	long skip;
	if (H.onefile) {
		double off = H.vox_offset;
		if (off < (double)H.size) off = (double)H.size;
		skip = (long)off - (long)H.size;  // header extensions between the header and the payload
	} else {
		gzclose(f);
		if (img_name.empty()) img_name = sibling(hdr_name, ".hdr", ".img");
		if (img_name.empty()) { fprintf(stderr, "readNiiFile: %s is the header of a two-file image but is not named *.hdr\n", hdr_name.c_str()); return nullptr; }
		f = gzopen(img_name.c_str(), "rb");
		if (!f) { fprintf(stderr, "readNiiFile: cannot open %s\n", img_name.c_str()); return nullptr; }
		skip = H.vox_offset > 0 ? (long)H.vox_offset : 0;
	}
	for (unsigned char junk[4096]; skip > 0;) {
		const int want = (int)std::min<long>(skip, (long)sizeof(junk));
		if (gzread(f, junk, (unsigned)want) != want) { gzclose(f); fprintf(stderr, "readNiiFile: truncated before the payload\n"); return nullptr; }
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
	} catch (...) { gzclose(f); fprintf(stderr, "readNiiFile: out of memory\n"); return nullptr; }
	gzclose(f);
	if (got != bytes) { fprintf(stderr, "readNiiFile: truncated payload\n"); return nullptr; }
	float *out = new (std::nothrow) float[n];
	if (!out) { fprintf(stderr, "readNiiFile: out of memory\n"); return nullptr; }
	const bool swap = H.swap;
	switch (H.datatype) {  // NIfTI datatype codes; slope/intercept deliberately ignored (see readNii.h)
	case 2: convert<uint8_t>(raw.data(), n, false, out); break;
	case 4: convert<int16_t>(raw.data(), n, swap, out); break;
	case 8: convert<int32_t>(raw.data(), n, swap, out); break;
	case 16: convert<float>(raw.data(), n, swap, out); break;
	case 64: convert<double>(raw.data(), n, swap, out); break;
	case 256: convert<int8_t>(raw.data(), n, false, out); break;
	case 512: convert<uint16_t>(raw.data(), n, swap, out); break;
	case 768: convert<uint32_t>(raw.data(), n, swap, out); break;
	default:
		fprintf(stderr, "readNiiFile: unsupported datatype %d\n", H.datatype);
		delete[] out;
		return nullptr;
	}
	nx = dx; ny = dy; nz = dz;
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

// tables.hip -- host-built constant tables of the extractor; each builder mirrors a reference routine with the same fp32 / fp64 mix.
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "ctx_internal.h"

#pragma clang fp contract(off)

namespace s3d {

// ---------------------------------------------------------------------------------------------
// host-side constant builders (each mirrors a reference routine; same fp32/fp64 mix)
// ---------------------------------------------------------------------------------------------
// GaussianSmooth_3D kernel generation, Src/cSIFT3D.cc:541-572
bool build_taps(float sigma, Taps &t) {
	sigma = sigma > 0 ? sigma : 0;
	int hw = 1;
	if (sigma > 0) {
		hw = (int)ceil((double)sigma * 3.0);
		if (hw < 1) hw = 1;
	}
	if (hw > kMaxHW) return false;
	t.hw = hw;
	const int width = 2 * hw + 1;
	float acc = 0;
	for (int i = 0; i < width; i++) {
		float x = (float)(i - hw);
		x = (float)((double)x / ((double)sigma + DBL_EPSILON));
		t.w[i] = (float)exp(-0.5 * (double)x * (double)x);
		acc += t.w[i];
	}
	for (int i = 0; i < width; i++) t.w[i] /= acc;
	for (int i = width; i < kMaxTaps; i++) t.w[i] = 0.f;
	return true;
}

// incremental blur schedule, Src/cSIFT3D.cc:272-287, 299
void level_sigmas(const sift3d_params &p, std::vector<float> &sig, float &base_sigma) {
	const int ng = p.num_kp_levels + 3;
	sig.assign(ng, 0.f);
	const float k = (float)pow(2.0, 1.0 / (double)p.num_kp_levels);
	const float base = (float)((double)p.sigma_default * pow(2.0, -1.0 / 3.0));
	sig[0] = base;
	for (int i = 1; i < ng; i++) {
		const float sig_prev = (float)(pow((double)k, (double)(i - 1)) * (double)base);
		const float sig_total = sig_prev * k;
		sig[i] = sqrtf(sig_total * sig_total - sig_prev * sig_prev);
	}
	base_sigma = sqrtf(sig[0] * sig[0] - p.sigma_n_default * p.sigma_n_default);
}

// icosahedron + hoisted cart2bary constants, Src/cUtil.cc:19-55, 113-175; Src/cSIFT3D.cc:1599-1619
static void cross3(const float *a, const float *b, float *o) {
	o[0] = a[1] * b[2] - a[2] * b[1];
	o[1] = a[2] * b[0] - a[0] * b[2];
	o[2] = a[0] * b[1] - a[1] * b[0];
}
static float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void build_faces(FaceConst *F) {
	const double gr = 1.6180339887;
	const double vert[12][3] = {{0, 1, gr}, {0, -1, gr}, {0, 1, -gr}, {0, -1, -gr}, {1, gr, 0}, {-1, gr, 0},
	                            {1, -gr, 0}, {-1, -gr, 0}, {gr, 0, 1}, {-gr, 0, 1}, {gr, 0, -1}, {-gr, 0, -1}};
	static const int faces[kFaces][3] = {{0, 1, 8}, {0, 8, 4}, {0, 4, 5}, {0, 5, 9}, {0, 9, 1}, {1, 6, 8}, {8, 6, 10},
	                                     {8, 10, 4}, {4, 10, 2}, {4, 2, 5}, {5, 2, 11}, {5, 11, 9}, {9, 11, 7}, {9, 7, 1},
	                                     {1, 7, 6}, {3, 6, 7}, {3, 7, 11}, {3, 11, 2}, {3, 2, 10}, {3, 10, 6}};
	for (int f = 0; f < kFaces; f++) {
		float v[3][3];
		for (int j = 0; j < 3; j++) {
			F[f].idx[j] = faces[f][j];
			float raw[3] = {(float)vert[faces[f][j]][0], (float)vert[faces[f][j]][1], (float)vert[faces[f][j]][2]};
			const double mag = (double)sqrtf(dot3(raw, raw));
			const double sca = 1.0 / mag;
			for (int c = 0; c < 3; c++) v[j][c] = (float)((double)raw[c] * sca);
		}
		float a[3], b[3], n[3];
		for (int c = 0; c < 3; c++) { a[c] = v[2][c] - v[1][c]; b[c] = v[1][c] - v[0][c]; }
		cross3(a, b, n);
		if (dot3(n, v[0]) < 0)
			for (int c = 0; c < 3; c++) std::swap(v[0][c], v[1][c]);
		for (int c = 0; c < 3; c++) {
			F[f].e1[c] = v[1][c] - v[0][c];
			F[f].e2[c] = v[2][c] - v[0][c];
			F[f].t[c] = (float)((double)v[0][c] * (-1.0));
		}
		cross3(F[f].t, F[f].e1, F[f].q);
		F[f].qe2 = dot3(F[f].q, F[f].e2);
		for (int c = 0; c < 3; c++) F[f].centre[c] = (v[0][c] + v[1][c] + v[2][c]) / 3.0f;
	}
}

// symmetry table of the face lookup (see FaceSym): for every (type, sign bits) find the mesh face whose three vertices are the
// sign-flipped canonical ones, and which of its vertices plays which role
bool build_facesym(const FaceConst *F, FaceSym *S) {
	const double gr = 1.6180339887, nrm = sqrt(1.0 + gr * gr);
	for (int type = 0; type < 4; type++)
		for (int bits = 0; bits < 8; bits++) {
			const double sx = (bits & 1) ? -1.0 : 1.0, sy = (bits & 2) ? -1.0 : 1.0, sz = (bits & 4) ? -1.0 : 1.0;
			const double A[3] = {0, sy, sz * gr}, B[3] = {sx, sy * gr, 0}, C[3] = {sx * gr, 0, sz};
			const double Am[3] = {0, -sy, sz * gr}, Bm[3] = {-sx, sy * gr, 0}, Cm[3] = {sx * gr, 0, -sz};  // mirrored across the straddled axis
			const double *role[3];
			switch (type) {
			case 0: role[0] = A; role[1] = B; role[2] = C; break;       // octant face
			case 1: role[0] = C; role[1] = Cm; role[2] = B; break;      // lambda_A < 0: across edge BC, the face that straddles z
			case 2: role[0] = A; role[1] = Am; role[2] = C; break;      // lambda_B < 0: across edge AC, straddles y
			default: role[0] = B; role[1] = Bm; role[2] = A; break;     // lambda_C < 0: across edge AB, straddles x
			}
			int found = -1, slot[3] = {-1, -1, -1};
			for (int f = 0; f < kFaces && found < 0; f++) {
				// geometric vertices of the face as the intersection test sees them (after the winding fix): v0 = -t, v1 = v0 + e1, v2 = v0 + e2
				double v[3][3];
				for (int c = 0; c < 3; c++) { v[0][c] = -(double)F[f].t[c]; v[1][c] = v[0][c] + (double)F[f].e1[c]; v[2][c] = v[0][c] + (double)F[f].e2[c]; }
				int sl[3] = {-1, -1, -1}, hit = 0;
				for (int r = 0; r < 3; r++)
					for (int j = 0; j < 3; j++) {
						double d = 0;
						for (int c = 0; c < 3; c++) d += fabs(v[j][c] - role[r][c] / nrm);
						if (d < 1e-4) { sl[r] = j; hit++; }
					}
				if (hit == 3 && sl[0] != sl[1] && sl[1] != sl[2] && sl[0] != sl[2]) { found = f; for (int r = 0; r < 3; r++) slot[r] = sl[r]; }
			}
			if (found < 0) return false;
			const int key = type * 8 + bits;
			S->face[key] = found;
			for (int r = 0; r < 3; r++) { S->slot[key][r] = slot[r]; S->vert[key][r] = F[found].idx[slot[r]]; }
		}
	return true;
}


// Gaussian window tables (see WinLut): orientation (Src/cSIFT3D.cc:915, 968-971) and descriptor
// (Src/cSIFT3D.cc:1155-1156, 1270, 1312) windows of every (octave, keypoint level).
// one table: which = 0 the orientation window (sigma, radius = 3 sigma), 1 the descriptor window of a keypoint of scale `scale`
// (sigma = 7.0711 scale, radius = 2 sigma) on a level of unit u; appended to `pool`.  Returns false when a descriptor table is too long for the LDS.
bool append_lut(std::vector<float> &pool, WinLut &L, int which, float sigma, float radius, float u, float scale) {
	bool fits_lds = true;
	const float r2 = radius * radius, uu = u * u;
	const int len = (int)floor((double)r2 / (double)uu) + 2;
	L.off = (int)pool.size(); L.len = len; L.nin = -1; L.radius = radius; L.sigma = sigma; L.fix_scale = 1.0f; L.list_off = -1; L.list_R = 0;
	if (which == 1) {
		// 32-bit histogram bins: a bin (cell, vertex) collects wgt * |g| * bary over the voxels within one cell of its centre;
		// the trilinear weights of those voxels sum to at most (cw + 2)^3 (cw = cell width in voxels = desc_width / (4u) =
		// 5 scale / u), |g| <= sqrt(3) (normalised data, |0.5 (a - b) / u| <= 1 per axis, weight <= 1), bary <= 1 + 2e-6
		const double cw = 5.0 * (double)scale / (double)u, bound = (cw + 2.0) * (cw + 2.0) * (cw + 2.0) * 1.7321 * 1.001;
		int k = (int)floor(log2(2147483647.0 / bound));
		k = std::max(0, std::min(k, 29));
		L.fix_scale = (float)ldexp(1.0, k);
	}
	if (which == 1 && len > kMaxDescLut) fits_lds = false;  // k_describe<false>: table read from global memory
	for (int n = 0; n < len; n++) {
		const float sq = (float)n * uu;  // exact: equals the reference's fp32 sum of squares
		float w;
		if (!(sq > r2)) L.nin = n;
		if (sq > r2) w = -1.0f;
		else if (which == 0) w = expf((float)(-0.5 * (double)sq / (double)(sigma * sigma)));
		else w = expf(-0.5f * sq / (sigma * sigma)) * (0.5f / u);  // exact scaling (u = 2^octave), see WinLut
		pool.push_back(w);
	}
	// sum of the weights over the lattice points of the sphere (k_describe's first guess of the gradient mass)
	const int R = (int)floor(sqrt((double)std::max(L.nin, 0)));
	double ws = 0.0;
	for (int dz = -R; dz <= R; dz++)
		for (int dy = -R; dy <= R; dy++)
			for (int dx = -R; dx <= R; dx++) {
				const int n = dx * dx + dy * dy + dz * dz;
				if (n <= L.nin) ws += (double)pool[(size_t)L.off + n] * (which == 1 ? (double)u / 0.5 : 1.0);
			}
	L.wsum = (float)std::max(ws, 1.0);
	if (which == 0 && R <= 127 && L.nin < 65536) {  // lattice points of the orientation sphere (WinLut::list_off)
		std::vector<unsigned> words((size_t)2 * R + 2);
		for (int dz = -R; dz <= R; dz++) {
			words[(size_t)(dz + R)] = (unsigned)(words.size() - ((size_t)2 * R + 2));
			for (int dy = -R; dy <= R; dy++)
				for (int dx = -R; dx <= R; dx++) {
					const int n = dx * dx + dy * dy + dz * dz;
					if (n <= L.nin) words.push_back((unsigned)(dx + 128) | (unsigned)(dy + 128) << 8 | (unsigned)n << 16);
				}
		}
		words[(size_t)2 * R + 1] = (unsigned)(words.size() - ((size_t)2 * R + 2));
		L.list_off = (int)pool.size(); L.list_R = R;
		pool.resize(pool.size() + words.size());
		memcpy(pool.data() + L.list_off, words.data(), words.size() * sizeof(unsigned));
	}
	return fits_lds;
}

}  // namespace s3d

// helpers.cpp -- host utilities of the reference's public header that operate on HOST data (Include/cSIFT3D.h:216-238,
// Include/cUtil.h:60).  NOT part of the extraction path: KpSiftAlgorithm runs on the GPU and never calls into this file (the kernels
// carry their own forms of these tests: kernels_detect.hip, kernels_orient.hip, kernels_desc.hip).  They exist so that user code that
// calls the reference's small helpers directly keeps compiling against this shell and gets the reference's answers.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <utility>

#include "../Include/cSIFT3D.h"

namespace CPUSIFT {

namespace {
const float kBaryEps = FLT_EPSILON * 1E1;  // Src/cSIFT3D.cc:23
inline float dot(const Cvec &a, const Cvec &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Cvec cross(const Cvec &a, const Cvec &b) { return Cvec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline Cvec sub(const Cvec &a, const Cvec &b) { return Cvec(a.x - b.x, a.y - b.y, a.z - b.z); }
}  // namespace

// transposition of two axes of a volume (dst is resized by the caller like the reference's callers do: MallocArrayMemory after the call
// would drop the data, so dst must already own nx*ny*nz floats)
void Im_permute(TexImage *src, TexImage *dst, int dim1, int dim2) {
	if (!src || !dst || !src->_Data || dim1 < 0 || dim1 > 2 || dim2 < 0 || dim2 > 2) return;
	int dim[3] = {src->GetDimX(), src->GetDimY(), src->GetDimZ()};
	float unit[3] = {src->GetUnitX(), src->GetUnitY(), src->GetUnitZ()};
	std::swap(dim[dim1], dim[dim2]);
	std::swap(unit[dim1], unit[dim2]);
	dst->SetImageSize(dim[0], dim[1], dim[2]);
	dst->SetImageUnit(unit[0], unit[1], unit[2]);
	if (!dst->_Data) dst->MallocArrayMemory();
	for (int z = 0; z < dim[2]; z++)
		for (int y = 0; y < dim[1]; y++)
			for (int x = 0; x < dim[0]; x++) {
				int c[3] = {x, y, z};
				std::swap(c[dim1], c[dim2]);
				dst->SetImageDataWithIdx(src->GetImageDataWithIdx(c[0], c[1], c[2]), x, y, z);
			}
}

// strict extremum of the centre voxel against its six face neighbours in `cur` and the same voxel one level down / up
bool IsExtrema_neighbor(TexImage *prev, TexImage *cur, TexImage *next, int x, int y, int z) {
	const float v = cur->GetImageDataWithIdx(x, y, z);
	const float nb[8] = {prev->GetImageDataWithIdx(x, y, z),    cur->GetImageDataWithIdx(x - 1, y, z), cur->GetImageDataWithIdx(x + 1, y, z),
	                     cur->GetImageDataWithIdx(x, y + 1, z), cur->GetImageDataWithIdx(x, y - 1, z), cur->GetImageDataWithIdx(x, y, z + 1),
	                     cur->GetImageDataWithIdx(x, y, z - 1), next->GetImageDataWithIdx(x, y, z)};
	bool is_min = true, is_max = true;
	for (float n : nb) { is_min = is_min && v < n; is_max = is_max && v > n; }
	return is_min || is_max;
}

bool DistinctEig(float a, float b, float c) {
	return !(fabs(a - b) < DBL_EPSILON || fabs(a - c) < DBL_EPSILON || fabs(c - b) < DBL_EPSILON);
}

void Swap_Element(float &a, float &b) { std::swap(a, b); }

void Transpose_Matrix(float *Rot) {  // 3 x 3, row major
	Swap_Element(Rot[1], Rot[3]);
	Swap_Element(Rot[2], Rot[6]);
	Swap_Element(Rot[5], Rot[7]);
}

// Moeller-Trumbore intersection of the ray through `cart` with the triangle: barycentric weights and the ray parameter k; -1 when the
// ray is parallel to the face.  The arithmetic follows the reference's operand order and its float / double mix.
int cart2bary(Cvec *cart, const Tri *const tri, Cvec *const bary, float *const k) {
	const Cvec *v = tri->v;
	Cvec e1 = sub(v[1], v[0]), e2 = sub(v[2], v[0]);
	Cvec t((float)(v[0].x * (-1.0)), (float)(v[0].y * (-1.0)), (float)(v[0].z * (-1.0)));
	Cvec p = cross(*cart, e2), q = cross(t, e1);
	const float det = dot(e1, p);
	if (fabsf(det) < kBaryEps) return -1;
	const float det_inv = (float)(1.0 / det);
	bary->y = det_inv * dot(p, t);
	bary->z = det_inv * dot(*cart, q);
	bary->x = 1 - bary->y - bary->z;
	*k = det_inv * dot(q, e2);
	return 0;
}

// first face (in mesh order) the gradient direction passes through, -1 for a vanishing gradient
int Check_intersect_faces(Mesh *mesh, Cvec *grad, Cvec *bary) {
	if (dot(*grad, *grad) < kBaryEps) return -1;
	for (int i = 0; i < mesh->num && i < ICOS_NFACES; i++) {
		float k;
		if (cart2bary(grad, mesh->tri + i, bary, &k) < 0) continue;
		if (bary->x < -kBaryEps || bary->y < -kBaryEps || bary->z < -kBaryEps || k < 0) continue;
		return i;
	}
	return -1;
}

// One window voxel into the 4 x 4 x 4 x 12 histogram (Src/cSIFT3D.cc:1383-1540): the face the gradient passes through, its barycentric
// weights, and the eight cells around the bin coordinates (base cell by TRUNCATION, fractions by floor -- the reference's pairing).  Host
// helper like the others of this file; the extraction path scatters inside k_describe.  trace (the _debug form with debug > 0): the
// fractions, barycentrics and face of voxel loop_idx, and per visited cell the three bin offsets and contributions.
namespace {
struct ScatterTrace { float *dvbins; int *face; float *bary; int *offset; float *accum; };
void scatter_voxel(Mesh *mesh, Keypoint &kp, const Cvec &vbins, Cvec &grad, int loop_idx, const ScatterTrace *trace) {
	const float frac[3] = {vbins.x - floorf(vbins.x), vbins.y - floorf(vbins.y), vbins.z - floorf(vbins.z)};
	const int cell[3] = {(int)vbins.x, (int)vbins.y, (int)vbins.z};
	Cvec bary;
	const int face = Check_intersect_faces(mesh, &grad, &bary);
	if (trace) {
		for (int a = 0; a < 3; a++) trace->dvbins[loop_idx * 3 + a] = frac[a];
		trace->bary[loop_idx * 3 + 0] = bary.x; trace->bary[loop_idx * 3 + 1] = bary.y; trace->bary[loop_idx * 3 + 2] = bary.z;
		trace->face[loop_idx] = face;
	}
	if (face < 0) return;
	const float mag = std::sqrt(dot(grad, grad));
	const float w3[3] = {bary.x, bary.y, bary.z};
	const Tri &T = mesh->tri[face];
	int visited = 0;
	for (int corner = 0; corner < 8; corner++) {  // x slowest, z fastest: the order the reference adds in
		const int up[3] = {corner >> 2, (corner >> 1) & 1, corner & 1};
		const int c[3] = {cell[0] + up[0], cell[1] + up[1], cell[2] + up[2]};
		bool inside = true;
		for (int a = 0; a < 3; a++) inside = inside && c[a] >= 0 && c[a] < NHIST_PER_DIM;
		if (!inside) continue;
		// the spatial weight is a double product (1.0 - f is a double expression in the reference), rounded once
		double wd = 1.0;
		for (int a = 0; a < 3; a++) wd *= up[a] ? (double)frac[a] : 1.0 - (double)frac[a];
		const float wgt = (float)wd;
		const int bin0 = (c[0] + NHIST_PER_DIM * c[1] + NHIST_PER_DIM * NHIST_PER_DIM * c[2]) * ICOS_NVERT;
		for (int v = 0; v < 3; v++) {
			const float add = mag * wgt * w3[v];
			kp.desc[bin0 + T.idx[v]] += add;
			if (trace) { trace->offset[(loop_idx * 8 + visited) * 3 + v] = bin0 + T.idx[v]; trace->accum[(loop_idx * 8 + visited) * 3 + v] = add; }
		}
		visited++;
	}
}
}  // namespace

void Trilinear_interpolation_over_desc(Mesh *mesh, Keypoint &kp, Cvec &vbins, Cvec &grad, int loop_idx) {
	scatter_voxel(mesh, kp, vbins, grad, loop_idx, nullptr);
}

void Trilinear_interpolation_over_desc_debug(Mesh *mesh, Keypoint &kp, Cvec &vbins, Cvec &grad, int loop_idx, float *host_dvbins,
                                             int *host_intersect_id, float *host_bary, int *host_offset, float *host_desc_accum, int debug) {
	const ScatterTrace t{host_dvbins, host_intersect_id, host_bary, host_offset, host_desc_accum};
	scatter_voxel(mesh, kp, vbins, grad, loop_idx, debug > 0 ? &t : nullptr);
}

void normailize_desc(float *desc) {
	float norm = 0.0f;
	for (int i = 0; i < DESC_NUMEL; i++) norm += desc[i] * desc[i];
	norm = (float)(sqrt(norm) + DBL_EPSILON);
	for (int i = 0; i < DESC_NUMEL; i++) desc[i] *= (float)(1.0 / norm);
}

int Initialize_geometry(Mesh *mesh) {
	static const double gr = 1.6180339887;
	static const double vert[ICOS_NVERT][3] = {{0, 1, gr}, {0, -1, gr}, {0, 1, -gr}, {0, -1, -gr}, {1, gr, 0},  {-1, gr, 0},
	                                           {1, -gr, 0}, {-1, -gr, 0}, {gr, 0, 1}, {-gr, 0, 1}, {gr, 0, -1}, {-gr, 0, -1}};
	static const int faces[ICOS_NFACES][3] = {{0, 1, 8}, {0, 8, 4},  {0, 4, 5},   {0, 5, 9},  {0, 9, 1},  {1, 6, 8},  {8, 6, 10},
	                                          {8, 10, 4}, {4, 10, 2}, {4, 2, 5},   {5, 2, 11}, {5, 11, 9}, {9, 11, 7}, {9, 7, 1},
	                                          {1, 7, 6},  {3, 6, 7},  {3, 7, 11},  {3, 11, 2}, {3, 2, 10}, {3, 10, 6}};
	mesh->num = ICOS_NFACES;
	mesh->tri = (Tri *)malloc(sizeof(Tri) * ICOS_NFACES);
	if (!mesh->tri) { mesh->num = 0; return 1; }
	for (int f = 0; f < ICOS_NFACES; f++) {
		Tri &T = mesh->tri[f];
		for (int j = 0; j < 3; j++) {
			T.idx[j] = faces[f][j];
			Cvec raw((float)vert[faces[f][j]][0], (float)vert[faces[f][j]][1], (float)vert[faces[f][j]][2]);
			const double sca = 1.0 / (double)sqrtf(dot(raw, raw));  // unit length, with the reference's float norm / double scale
			T.v[j] = Cvec((float)((double)raw.x * sca), (float)((double)raw.y * sca), (float)((double)raw.z * sca));
		}
		// a face whose normal points inwards has the COORDINATES of its first two vertices exchanged -- not their bin indices
		// (Src/cUtil.cc:164-171): the histogram bins of such a face trade places; the device tables reproduce the same quirk
		const Cvec n = cross(sub(T.v[2], T.v[1]), sub(T.v[1], T.v[0]));
		if (dot(n, T.v[0]) < 0) std::swap(T.v[0], T.v[1]);
	}
	return 0;
}

}  // namespace CPUSIFT

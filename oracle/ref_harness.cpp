/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Thin C driver around the UNTOUCHED reference sources in /root/reference/3DSIFT.  It is
 * compiled together with those sources (from where they lie; see oracle/Makefile target
 * `ref`) into oracle/_ref/libref3dsift.so and exports the ref_* flavour of oracle_api.h.
 * It is used only (i) to validate our own restatement oracle/sift3d_oracle.c and (ii) by
 * tests/golden/make_golden.py to emit the committed golden fixtures.
 *
 * Nothing here re-implements reference arithmetic: every function calls straight into the
 * reference's public API (cSIFT3D.h:142-239, cMatcher.h:59-86, cUtil.h).
 */
#define ORACLE_PREFIX_REF 1
#include "oracle_api.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <omp.h>

#include "Include/cSIFT3D.h"
#include "Include/cMatcher.h"
#include "Include/cUtil.h"

using namespace CPUSIFT;

namespace {

/* the reference prints unconditionally (cSIFT3D.cc:173,198,369,386,...): mute fd 1/2 */
struct Mute {
	int o1, o2;
	Mute() {
		fflush(stdout);
		fflush(stderr);
		std::cout.flush();
		std::cerr.flush();
		o1 = o2 = -1;
		if (getenv("ORC_VERBOSE")) return;
		int dn = open("/dev/null", O_WRONLY);
		o1 = dup(1);
		o2 = dup(2);
		dup2(dn, 1);
		dup2(dn, 2);
		close(dn);
	}
	~Mute() {
		fflush(stdout);
		fflush(stderr);
		std::cout.flush();
		std::cerr.flush();
		if (o1 < 0) return;
		dup2(o1, 1);
		dup2(o2, 2);
		close(o1);
		close(o2);
	}
};

/* derived class only to reach the protected result vectors */
struct Probe : public CSIFT3D {
	Probe(float *v, int x, int y, int z, int l, float s, float sn, float p, float e, float c)
	    : CSIFT3D(v, x, y, z, l, s, sn, p, e, c), levels(l) {}
	int levels;
	std::vector<Keypoint> extrema_snapshot;
	std::vector<Keypoint> &ex() { return extre; }
	std::vector<Keypoint> &kp() { return filter; }
	TexImage &input() { return Host_Im; }
	int octs() { return octave_num; }
};

void to_pod(const Keypoint &k, orc_kp *o) {
	o->x = k.x; o->y = k.y; o->z = k.z;
	o->scale = k.scale; o->octave = k.octave; o->level = k.level;
	o->rx = k.rx; o->ry = k.ry; o->rz = k.rz;
	o->win[0] = k.win.x; o->win[1] = k.win.y; o->win[2] = k.win.z;
	memcpy(o->eigvalue, k.eigvalue, sizeof(k.eigvalue));
	memcpy(o->eigvector, k.eigvector, sizeof(k.eigvector));
	memcpy(o->Rotation, k.Rotation, sizeof(k.Rotation));
	memcpy(o->str_tensor, k.str_tensor, sizeof(k.str_tensor));
}

void from_pod(const orc_kp *o, Keypoint &k) {
	k.x = o->x; k.y = o->y; k.z = o->z;
	k.scale = o->scale; k.octave = o->octave; k.level = o->level;
	k.rx = o->rx; k.ry = o->ry; k.rz = o->rz;
	k.win = Cvec(o->win[0], o->win[1], o->win[2]);
	memcpy(k.eigvalue, o->eigvalue, sizeof(k.eigvalue));
	memcpy(k.eigvector, o->eigvector, sizeof(k.eigvector));
	memcpy(k.Rotation, o->Rotation, sizeof(k.Rotation));
	memcpy(k.str_tensor, o->str_tensor, sizeof(k.str_tensor));
}

/* a TexImage that borrows caller memory (its dtor would free() _Data otherwise) */
struct Borrowed {
	TexImage im;
	Borrowed(const float *data, int nx, int ny, int nz, float unit) {
		im.SetImageSize(nx, ny, nz);
		im.SetImageUnit(unit, unit, unit);
		im.SetImageDataPt(const_cast<float *>(data));
	}
	~Borrowed() { im.SetImageDataPt(nullptr); }
};

} // namespace

extern "C" {

void *ref_create(const float *volume, int nx, int ny, int nz, int num_kp_levels, float sigma_default,
                 float sigma_n_default, float peak_thresh, float max_eig_thres, float corner_thresh) {
	Mute m;
	return new Probe(const_cast<float *>(volume), nx, ny, nz, num_kp_levels, sigma_default,
	                 sigma_n_default, peak_thresh, max_eig_thres, corner_thresh);
}

void ref_destroy(void *h) {
	Mute m;
	delete static_cast<Probe *>(h);
}

void ref_set_threads(int n) {
	if (n > 0) {
		sift_thread_num = n;
		omp_set_num_threads(n);
	}
}

void ref_run(void *h, int upto, double *times) {
	Mute m;
	Probe *p = static_cast<Probe *>(h);
	double t[7];
	t[0] = omp_get_wtime();
	p->Initialize();
	t[1] = omp_get_wtime();
	p->Build_Gaussian_Scale_Space();
	t[2] = omp_get_wtime();
	t[3] = t[4] = t[5] = t[6] = t[2];
	if (upto >= 2) { p->Build_DOG_Scale_Space(); t[3] = t[4] = t[5] = t[6] = omp_get_wtime(); }
	if (upto >= 3) {
		p->Detect_KeyPoints();
		p->extrema_snapshot = p->ex();
		t[4] = t[5] = t[6] = omp_get_wtime();
	}
	if (upto >= 4) { p->Assign_Orientation(); t[5] = t[6] = omp_get_wtime(); }
	if (upto >= 5) { p->Extract_Description(); t[6] = omp_get_wtime(); }
	if (times)
		for (int i = 0; i < 6; i++) times[i] = t[i + 1] - t[i];
}

int ref_num_octaves(void *h) { return static_cast<Probe *>(h)->octs(); }

static TexImage &lvl(void *h, int is_dog, int idx) {
	Probe *p = static_cast<Probe *>(h);
	return is_dog ? (*p->GET_DOG())[idx] : (*p->GET_GSS())[idx];
}

void ref_level_info(void *h, int is_dog, int idx, int *dims3, float *units3, float *scale) {
	TexImage &t = lvl(h, is_dog, idx);
	dims3[0] = t.GetDimX(); dims3[1] = t.GetDimY(); dims3[2] = t.GetDimZ();
	units3[0] = t.GetUnitX(); units3[1] = t.GetUnitY(); units3[2] = t.GetUnitZ();
	*scale = t.GetScale();
}

void ref_copy_level(void *h, int is_dog, int idx, float *out) {
	TexImage &t = lvl(h, is_dog, idx);
	memcpy(out, t._Data, sizeof(float) * (size_t)t.GetDimX() * t.GetDimY() * t.GetDimZ());
}

void ref_copy_input(void *h, float *out) {
	TexImage &t = static_cast<Probe *>(h)->input();
	memcpy(out, t._Data, sizeof(float) * (size_t)t.GetDimX() * t.GetDimY() * t.GetDimZ());
}

int ref_num_extrema(void *h) { return (int)static_cast<Probe *>(h)->extrema_snapshot.size(); }

void ref_copy_extrema(void *h, orc_kp *out) {
	Probe *p = static_cast<Probe *>(h);
	for (size_t i = 0; i < p->extrema_snapshot.size(); i++) to_pod(p->extrema_snapshot[i], out + i);
}

int ref_num_keypoints(void *h) { return (int)static_cast<Probe *>(h)->kp().size(); }

void ref_copy_keypoints(void *h, orc_kp *out, float *desc) {
	Probe *p = static_cast<Probe *>(h);
	std::vector<Keypoint> &v = p->kp();
	for (size_t i = 0; i < v.size(); i++) {
		to_pod(v[i], out + i);
		if (desc && v[i].desc) memcpy(desc + i * DESC_NUMEL, v[i].desc, sizeof(float) * DESC_NUMEL);
	}
}

void ref_gaussian_smooth(const float *src, int nx, int ny, int nz, float sigma, float *dst) {
	Mute m;
	Borrowed s(src, nx, ny, nz, 1.0f);
	TexImage d;
	d.SetImageSize(nx, ny, nz);
	d.SetImageUnit(1, 1, 1);
	d.MallocArrayMemory();
	GaussianSmooth_3D(&s.im, &d, sigma);
	memcpy(dst, d._Data, sizeof(float) * (size_t)nx * ny * nz);
}

int ref_gaussian_taps(float, float *) { return -1; /* the reference has no tap accessor (cSIFT3D.cc:546-572 is inline) */ }

int ref_mesh(float *verts, int *idx) {
	Mesh mesh;
	Initialize_geometry(&mesh);
	for (int f = 0; f < ICOS_NFACES; f++)
		for (int j = 0; j < 3; j++) {
			verts[(f * 3 + j) * 3 + 0] = mesh.tri[f].v[j].x;
			verts[(f * 3 + j) * 3 + 1] = mesh.tri[f].v[j].y;
			verts[(f * 3 + j) * 3 + 2] = mesh.tri[f].v[j].z;
			idx[f * 3 + j] = mesh.tri[f].idx[j];
		}
	free(mesh.tri);
	return ICOS_NFACES;
}

int ref_intersect(const float *g, float *bary3) {
	static Mesh mesh = {nullptr, 0};
	if (!mesh.tri) Initialize_geometry(&mesh);
	Cvec grad(g[0], g[1], g[2]), bary;
	int r = Check_intersect_faces(&mesh, &grad, &bary);
	bary3[0] = bary.x; bary3[1] = bary.y; bary3[2] = bary.z;
	return r;
}

int ref_orient_one(orc_kp *kp, const float *level, int nx, int ny, int nz, float unit, float sigma,
                   float max_eig_ratio, float corner_thresh) {
	Mute m;
	Borrowed g(level, nx, ny, nz, unit);
	Keypoint k;
	from_pod(kp, k);
	Initialize_Keypoint(k);
	int r = Assign_Orientation_Imp(k, &g.im, sigma, max_eig_ratio, corner_thresh);
	to_pod(k, kp);
	return r;
}

void ref_describe_one(orc_kp *kp, const float *level, int nx, int ny, int nz, float unit, float *desc768) {
	Mute m;
	static Mesh mesh = {nullptr, 0};
	if (!mesh.tri) Initialize_geometry(&mesh);
	Borrowed g(level, nx, ny, nz, unit);
	Keypoint k;
	from_pod(kp, k);
	memset(desc768, 0, sizeof(float) * DESC_NUMEL);
	k.desc = desc768;
	Extract_Descriptor_Imp(k, &g.im, &mesh);
	k.desc = nullptr;
	to_pod(k, kp);
}

int ref_match(const float *ref_desc, const float *ref_xyz, int n, const float *tar_desc, const float *tar_xyz,
              int m, double thresh, int mode, int *gIdx, int *sIdx, float *gDist, float *sDist, float *pairs6) {
	Mute mute;
	std::vector<Keypoint> a(n), b(m);
	for (int i = 0; i < n; i++) {
		a[i].desc = const_cast<float *>(ref_desc + (size_t)i * DESC_NUMEL);
		a[i].rx = ref_xyz[i * 3]; a[i].ry = ref_xyz[i * 3 + 1]; a[i].rz = ref_xyz[i * 3 + 2];
	}
	for (int i = 0; i < m; i++) {
		b[i].desc = const_cast<float *>(tar_desc + (size_t)i * DESC_NUMEL);
		b[i].rx = tar_xyz[i * 3]; b[i].ry = tar_xyz[i * 3 + 1]; b[i].rz = tar_xyz[i * 3 + 2];
	}
	muBruteMatcher mt;
	std::vector<Cvec> ra, rb;
	if (mode == 1) mt.injectMatch(ra, rb, a, b, thresh);
	else if (mode == 2) mt.bijectMatch(ra, rb, a, b, thresh);
	else mt.enhancedMatch(ra, rb, a, b, thresh);
	std::vector<int> gi = mt.getGlodenIdx(), si = mt.getSilverIdx();
	std::vector<float> gd = mt.getGlodenDistSquare(), sd = mt.getSilverDistSquare();
	for (int i = 0; i < n; i++) {
		if (gIdx) gIdx[i] = gi[i];
		if (sIdx) sIdx[i] = si[i];
		if (gDist) gDist[i] = gd[i];
		if (sDist) sDist[i] = sd[i];
	}
	for (size_t i = 0; i < ra.size(); i++) {
		pairs6[i * 6 + 0] = ra[i].x; pairs6[i * 6 + 1] = ra[i].y; pairs6[i * 6 + 2] = ra[i].z;
		pairs6[i * 6 + 3] = rb[i].x; pairs6[i * 6 + 4] = rb[i].y; pairs6[i * 6 + 5] = rb[i].z;
	}
	return (int)ra.size();
}

/* key-point coordinate lists: the reference's own writer / reader (Src/cUtil.cc:938-954, 1002-1016), for golden g10.
 * Not part of oracle_api.h: the C restatement has no file IO; these pin 3dsift_amd/host/src/io.cpp. */
void ref_write_sift_kp(const float *xyz, int n, const char *path) {
	std::vector<Cvec> v;
	for (int i = 0; i < n; i++) v.push_back(Cvec(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
	Mute m;
	write_sift_kp(v, path);
}

int ref_read_sift_kp(const char *path, float *xyz, int cap) {
	std::vector<Cvec> v;
	{
		Mute m;
		read_sift_kp(path, v);
	}
	for (int i = 0; i < (int)v.size() && i < cap; i++) { xyz[3 * i] = v[i].x; xyz[3 * i + 1] = v[i].y; xyz[3 * i + 2] = v[i].z; }
	return (int)v.size();
}

} /* extern "C" */

// cSIFT3D.cpp -- the CPUSIFT::CSIFT3D shell over the C-ABI (include/sift3d_hip.h).
// Mirrors the reference's construction / run / read-back contract (3DSIFT/Src/cSIFT3D.cc:103-235,
// 1659-1688): ctor copies + normalises the caller's volume, KpSiftAlgorithm() runs the pipeline,
// GetKeypoints() returns records whose desc pointers alias extractor-owned memory.  Like the
// reference it reports problems on stderr and carries on with an empty result (no exceptions).
#include "../Include/cSIFT3D.h"

#include <cstddef>
#include <cstdio>
#include <iostream>
#include <cstdlib>
#include <cstring>
#include <map>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/sift3d_hip.h"
#include "../Include/Util/matrixIO3D.h"

namespace CPUSIFT {

int sift_thread_num = 1;

static int g_device = -1;
void SetDevice(int device) { g_device = device; }

// ---- free functions of the reference's header (Include/cSIFT3D.h:208-218) ----
static bool contiguous(TexImage *t) {
	return t && t->_Data && t->GetXstride() == 1 && t->GetYstride() == (size_t)t->GetDimX() &&
	       t->GetZstride() == (size_t)t->GetDimX() * (size_t)t->GetDimY();
}
static void shape_like(TexImage *dst, TexImage *src) {
	if (dst->GetDimX() != src->GetDimX() || dst->GetDimY() != src->GetDimY() || dst->GetDimZ() != src->GetDimZ() || !dst->_Data) {
		dst->SetImageSize(src->GetDimX(), src->GetDimY(), src->GetDimZ());
		dst->MallocArrayMemory();
	}
	dst->SetImageUnit(src->GetUnitX(), src->GetUnitY(), src->GetUnitZ());
}
void GaussianSmooth_3D(TexImage *src, TexImage *dst, float sigma) {  // Src/cSIFT3D.cc:535-622
	if (!contiguous(src) || !dst) { std::cerr << "GaussianSmooth_3D: bad image" << std::endl; return; }
	shape_like(dst, src);
	const int rc = sift3d_gaussian_smooth(src->_Data, src->GetDimX(), src->GetDimY(), src->GetDimZ(), sigma, dst->_Data, GetDevice());
	if (rc) std::cerr << "GaussianSmooth_3D: " << sift3d_last_error() << std::endl;
}
void DownSample_3D(TexImage *src, TexImage *dst) {  // Src/cSIFT3D.cc:506-533
	if (!contiguous(src) || !contiguous(dst)) { std::cerr << "DownSample_3D: bad image" << std::endl; return; }
	const int rc = sift3d_downsample(src->_Data, src->GetDimX(), src->GetDimY(), src->GetDimZ(), dst->_Data, dst->GetDimX(), dst->GetDimY(),
	                                 dst->GetDimZ(), GetDevice());
	if (rc) std::cerr << "DownSample_3D: " << (rc == SIFT3D_ERR_ARG ? "dst does not fit src / 2" : sift3d_last_error()) << std::endl;
}
void Sub(TexImage *prev, TexImage *cur, TexImage *dog) {  // Src/cSIFT3D.cc:849-882
	if (!contiguous(prev) || !contiguous(cur) || !dog || prev->GetDimX() != cur->GetDimX() || prev->GetDimY() != cur->GetDimY() ||
	    prev->GetDimZ() != cur->GetDimZ()) {
		std::cout << "Wrong in Sub function" << std::endl;  // the reference's message (cc:855)
		return;
	}
	shape_like(dog, prev);
	const int rc = sift3d_dog_sub(prev->_Data, cur->_Data, (size_t)prev->GetDimX() * prev->GetDimY() * prev->GetDimZ(), dog->_Data, GetDevice());
	if (rc) std::cerr << "Sub: " << sift3d_last_error() << std::endl;
}
void GaussianSmooth_3D_Imp(TexImage *src, TexImage *dst, int dim, float /*unit: the reference blurs in voxel coordinates, cc:645*/, float *weight,
                           int width) {  // Src/cSIFT3D.cc:624-788
	if (!contiguous(src) || !dst || !weight) { std::cerr << "GaussianSmooth_3D_Imp: bad image" << std::endl; return; }
	shape_like(dst, src);
	const int rc = sift3d_conv_axis(src->_Data, src->GetDimX(), src->GetDimY(), src->GetDimZ(), dim, weight, width, dst->_Data, GetDevice());
	if (rc) std::cerr << "GaussianSmooth_3D_Imp: " << sift3d_last_error() << std::endl;
}

// One keypoint on a caller-held level (Include/cSIFT3D.h:224, 228 of the reference): the device kernels of the pipeline on the box of the
// level the window reaches (sift3d_orient_keypoint / sift3d_describe_keypoint).  As in the pipeline the keypoint sits on a voxel and the
// level has one power-of-two unit; other inputs are reported on stderr and rejected (orientation: return 0, "not run").
static bool kp_level_ok(const char *who, TexImage *g) {
	if (!contiguous(g) || g->GetUnitX() != g->GetUnitY() || g->GetUnitX() != g->GetUnitZ()) {
		std::cerr << who << ": the level must be contiguous with one unit on all axes" << std::endl;
		return false;
	}
	return true;
}
static_assert(sizeof(sift3d_keypoint) == offsetof(Keypoint, desc), "Keypoint = sift3d_keypoint + desc pointer");
int Assign_Orientation_Imp(Keypoint &kp, TexImage *gaussian, const float sigma, const float max_eig_ratio, const float corner_thresh) {  // cc:913-1138
	if (!kp_level_ok("Assign_Orientation_Imp", gaussian)) return 0;
	int code = 0;
	const int rc = sift3d_orient_keypoint(gaussian->_Data, gaussian->GetDimX(), gaussian->GetDimY(), gaussian->GetDimZ(), gaussian->GetUnitX(),
	                                      reinterpret_cast<sift3d_keypoint *>(&kp), sigma, max_eig_ratio, corner_thresh, GetDevice(), &code);
	if (rc) { std::cerr << "Assign_Orientation_Imp: " << sift3d_last_error() << std::endl; return 0; }
	return code;
}
void Extract_Descriptor_Imp(Keypoint &kp, TexImage *gaussian, Mesh *mesh) {  // cc:1152-1381; the mesh is the library's own icosahedron (Initialize_geometry)
	(void)mesh;
	if (!kp.desc) { std::cerr << "Extract_Descriptor_Imp: kp.desc must point to DESC_NUMEL floats" << std::endl; return; }
	if (!kp_level_ok("Extract_Descriptor_Imp", gaussian)) return;
	const int rc = sift3d_describe_keypoint(gaussian->_Data, gaussian->GetDimX(), gaussian->GetDimY(), gaussian->GetDimZ(), gaussian->GetUnitX(),
	                                        reinterpret_cast<sift3d_keypoint *>(&kp), kp.desc, GetDevice());
	if (rc) std::cerr << "Extract_Descriptor_Imp: " << sift3d_last_error() << std::endl;
}
int GetDevice() {
	if (g_device < 0) {
		const char *e = getenv("SIFT3D_DEVICE");
		g_device = e ? atoi(e) : 0;
	}
	return g_device;
}

// live extractors by the host address of their descriptor block (Keypoint::desc of row 0): lets the matcher recognise keypoint
// vectors that still alias an extractor and read that extractor's device-resident results instead
static std::mutex g_reg_mu;
static std::map<const float *, CSIFT3D *> g_reg;

// SIFT3D_DEVICES = "0,1,2,3" or "0-7": GPUs one extractor shards a volume over (z-slabs, halo exchange over RCCL; csrc/sharded.hip);
// SIFT3D_SIM_RANKS = n: that many ranks simulated on the one selected device (how a 1-GPU box tests the sharded driver)
static std::vector<int> shard_devices() {
	std::vector<int> d;
	const char *e = getenv("SIFT3D_DEVICES");
	if (!e) return d;
	for (const char *p = e; *p;) {
		char *q = nullptr;
		long a = strtol(p, &q, 10);
		if (q == p) break;
		long b = a;
		if (*q == '-') { const char *r = q + 1; b = strtol(r, &q, 10); if (q == r) break; }
		for (long v = a; v <= b && d.size() < 64; v++) d.push_back((int)v);
		p = (*q == ',') ? q + 1 : q;
		if (*q && *q != ',') break;
	}
	return d;
}
static int sim_ranks() {
	const char *e = getenv("SIFT3D_SIM_RANKS");
	return e ? atoi(e) : 0;
}
// SIFT3D_PARTIAL_WINDOWS: unset = the driver's rule (descriptor windows split along z over the ranks unless a slab is too thin for it),
// 1 = split or refuse, 0 = whole windows on the wide halos (sift3d_sharded_create_ex)
// SIFT3D_GHOST_OCTAVE0=1 (see below); SIFT3D_TRANSPORT=copies: the driver's copy transport instead of RCCL (events + device / peer copies; SIFT3D_DEVICES may then name a device
// several times: that many rank threads on it)
static unsigned shard_flags() {
	unsigned f = 0u;
	const char *t = getenv("SIFT3D_TRANSPORT");
	if (t && strcmp(t, "copies") == 0) f |= SIFT3D_SHARDED_COPY_TRANSPORT;
	const char *g = getenv("SIFT3D_GHOST_OCTAVE0");  // 1: octave 0 on ghost zones (recomputed instead of exchanged level by level; for link-bound nodes)
	if (g && atoi(g) > 0) f |= SIFT3D_SHARDED_GHOST_OCTAVE0;
	const char *e = getenv("SIFT3D_PARTIAL_WINDOWS");
	if (!e) return f;
	return f | (atoi(e) > 0 ? SIFT3D_SHARDED_PARTIAL_WINDOWS : SIFT3D_SHARDED_WHOLE_WINDOWS);
}

struct CSIFT3D::Impl {
	sift3d_handle h = nullptr;
	sift3d_sharded_handle sh = nullptr;  // set instead of h when the volume is sharded over several GPUs
	int device = 0;
	int levels = 3;
	int stage = 0;
	bool fetched = false;
	bool in_flight = false;  // KpSiftAlgorithmAsync() enqueued a run that Wait() has not completed
	unsigned long long desc_hash = 0;  // of the host descriptor block as fetched (OwnerOf: has the caller edited it since?)
};

// order-sensitive hash of the host descriptor block (four independent multiply-add lanes: ~2 ms per 35 MB): Keypoint::desc is a
// mutable float*, like the reference's, and a caller may edit descriptors in place (re-normalise, RootSIFT, zero rows) before
// matching -- the device-resident copy is only a valid stand-in for a block that still hashes to what was fetched
static unsigned long long hash_block(const float *p, size_t nfloats) {
	const unsigned long long *w = reinterpret_cast<const unsigned long long *>(p);
	const size_t n = nfloats / 2;
	unsigned long long h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
	size_t i = 0;
	for (; i + 4 <= n; i += 4)
		for (int k = 0; k < 4; k++) h[k] = h[k] * 0x100000001B3ull + w[i + k];
	for (; i < n; i++) h[0] = h[0] * 0x100000001B3ull + w[i];
	if (nfloats & 1) { unsigned v; memcpy(&v, p + nfloats - 1, 4); h[1] = h[1] * 0x100000001B3ull + v; }
	return (h[0] ^ (h[1] * 3)) + (h[2] ^ (h[3] * 5));
}

static void complain(const char *where, int rc) {
	if (rc != SIFT3D_OK) fprintf(stderr, "[3dsift_amd] %s: %s (%s)\n", where, sift3d_error_string(rc), sift3d_last_error());
}

bool cmp_kp(const Keypoint &a, const Keypoint &b) {
	if (a.rz != b.rz) return a.rz < b.rz;
	if (a.ry != b.ry) return a.ry < b.ry;
	return a.rx < b.rx;
}

bool cmp_kp_orig(const Keypoint &a, const Keypoint &b) {
	if (a.z != b.z) return a.z < b.z;
	if (a.y != b.y) return a.y < b.y;
	return a.x < b.x;
}

CSIFT3D::CSIFT3D() : impl(new Impl()) {}

CSIFT3D::CSIFT3D(float *volume, int x_dim, int y_dim, int z_dim, int num_kp_levels_, float sigma_default_,
                 float sigma_n_default_, float peak_thresh_, float max_eig_thres_, float corner_thresh_)
    : impl(new Impl()) {
	sift3d_params p;
	p.num_kp_levels = num_kp_levels_;
	p.sigma_default = sigma_default_;
	p.sigma_n_default = sigma_n_default_;
	p.peak_thresh = peak_thresh_;
	p.max_eig_thres = max_eig_thres_;
	p.corner_thresh = corner_thresh_;
	impl->levels = num_kp_levels_;
	impl->device = GetDevice();
	const std::vector<int> devs = shard_devices();
	const int sim = sim_ranks();
	if (devs.size() > 1 || sim > 1) {
		const int one = impl->device;
		const int rc = sift3d_sharded_create_ex(&impl->sh, volume, x_dim, y_dim, z_dim, &p, sim > 1 ? &one : devs.data(), sim > 1 ? 1 : (int)devs.size(),
		                                        sim > 1 ? sim : 0, 0, shard_flags());
		complain("CSIFT3D (sharded upload + normalise)", rc);
		return;
	}
	complain("CSIFT3D (upload + normalise)", sift3d_create(&impl->h, volume, x_dim, y_dim, z_dim, &p, impl->device, 0));
}

CSIFT3D::~CSIFT3D() {
	// out of the registry first (under the lock OwnerOf holds while it reads filter / impl / the block), then tear down
	if (global_descriptor) { std::lock_guard<std::mutex> lk(g_reg_mu); g_reg.erase(global_descriptor); }
	if (impl) {
		if (impl->h) sift3d_destroy(impl->h);
		if (impl->sh) sift3d_sharded_destroy(impl->sh);
		delete impl;
	}
	if (global_descriptor) free(global_descriptor);
}

CSIFT3D *CSIFT3D::OwnerOf(const std::vector<Keypoint> &kp) {
	if (kp.empty() || !kp[0].desc) return nullptr;
	// the registry lock is held for the whole check: ~CSIFT3D and fetch_results take it before they touch the block or the registry
	std::lock_guard<std::mutex> lk(g_reg_mu);
	auto it = g_reg.find(kp[0].desc);
	CSIFT3D *o = it != g_reg.end() ? it->second : nullptr;
	if (!o || o->filter.size() != kp.size() || !o->impl || o->impl->stage < 5 || !o->impl->fetched || !o->global_descriptor) return nullptr;
	for (size_t i = 0; i < kp.size(); i++) {
		const Keypoint &a = kp[i], &b = o->filter[i];
		if (a.desc != b.desc || a.rx != b.rx || a.ry != b.ry || a.rz != b.rz) return nullptr;
	}
	// same rows in the same order -- and the CONTENTS are still what the extractor produced (the reference matcher reads
	// kp[i].desc, so in-place edits of the descriptors must reach the match: they take the gather path)
	if (hash_block(o->global_descriptor, kp.size() * (size_t)DESC_NUMEL) != o->impl->desc_hash) return nullptr;
	return o;
}

void CSIFT3D::fetch_results() {
	filter.clear();
	if (global_descriptor) {
		{ std::lock_guard<std::mutex> lk(g_reg_mu); g_reg.erase(global_descriptor); }
		free(global_descriptor);
		global_descriptor = nullptr;
	}
	impl->fetched = true;
	if ((!impl->h && !impl->sh) || impl->stage < 4) return;
	int n = 0;
	if (impl->sh) sift3d_sharded_num_keypoints(impl->sh, &n);
	else sift3d_num_keypoints(impl->h, &n);
	if (n <= 0) return;
	std::vector<sift3d_keypoint> pod((size_t)n);
	const bool with_desc = impl->stage >= 5;
	global_descriptor = (float *)calloc((size_t)n * DESC_NUMEL, sizeof(float));
	int rc = impl->sh ? sift3d_sharded_get_keypoints(impl->sh, pod.data(), with_desc ? global_descriptor : nullptr)
	                  : sift3d_get_keypoints(impl->h, pod.data(), with_desc ? global_descriptor : nullptr);
	complain("GetKeypoints", rc);
	if (rc != SIFT3D_OK) return;
	filter.resize((size_t)n);
	for (int i = 0; i < n; i++) {
		Keypoint &k = filter[i];
		const sift3d_keypoint &s = pod[i];
		k.x = s.x; k.y = s.y; k.z = s.z; k.scale = s.scale; k.octave = s.octave; k.level = s.level;
		k.rx = s.rx; k.ry = s.ry; k.rz = s.rz;
		k.win = Cvec(s.win[0], s.win[1], s.win[2]);
		memcpy(k.eigvalue, s.eigvalue, sizeof(k.eigvalue));
		memcpy(k.eigvector, s.eigvector, sizeof(k.eigvector));
		memcpy(k.Rotation, s.Rotation, sizeof(k.Rotation));
		memcpy(k.str_tensor, s.str_tensor, sizeof(k.str_tensor));
		k.desc = global_descriptor + (size_t)i * DESC_NUMEL;
	}
	if (with_desc && !impl->sh) {  // (sharded results live on several devices: the matcher takes the host path)
		impl->desc_hash = hash_block(global_descriptor, (size_t)n * DESC_NUMEL);
		std::lock_guard<std::mutex> lk(g_reg_mu);
		g_reg[global_descriptor] = this;
	}
}

// a call that needs the single-GPU handle on an extractor that has none: say why instead of returning an empty result silently
static bool need_single(const char *what, sift3d_handle h, sift3d_sharded_handle sh) {
	if (h) return true;
	if (sh) fprintf(stderr, "[3dsift_amd] %s: this extractor shards its volume over several GPUs (SIFT3D_DEVICES / SIFT3D_SIM_RANKS); the "
	                        "stage-by-stage methods and the pyramid / extrema accessors need a single-GPU extractor -- use KpSiftAlgorithm()\n", what);
	else fprintf(stderr, "[3dsift_amd] %s: the extractor was not constructed (see the message of CreateCSIFT3D)\n", what);
	return false;
}

static void run_to(CSIFT3D *self, sift3d_handle h, int upto, int &stage, bool &fetched, SIFT_TimerPara &tm) {
	if (!h) return;
	int rc = sift3d_run_stages(h, upto);
	complain("KpSiftAlgorithm", rc);
	if (rc != SIFT3D_OK) return;
	stage = upto;
	fetched = false;
	double t[8];
	if (sift3d_stage_times(h, t) == SIFT3D_OK) {
		tm.d_TotalTime = t[0]; tm.d_Allocation = t[1]; tm.d_BuildGSS = t[2]; tm.d_BuildDOG = t[3];
		tm.d_Detect = t[4]; tm.d_AssignOrientation = t[5]; tm.d_Extraction = t[6]; tm.d_release = t[7];
	}
	(void)self;
}

void CSIFT3D::KpSiftAlgorithm() {
	if (impl->sh) {  // sharded over several GPUs: the whole pipeline in one call (the stage-by-stage methods need a single-GPU extractor)
		const int rc = sift3d_sharded_run(impl->sh);
		complain("KpSiftAlgorithm (sharded)", rc);
		if (rc != SIFT3D_OK) return;
		impl->stage = 5;
		impl->fetched = false;
		double sec[2] = {0, 0};
		sift3d_sharded_info(impl->sh, nullptr, nullptr, nullptr, sec);
		m_timer.d_TotalTime = sec[0];
		return;
	}
	if (need_single("KpSiftAlgorithm", impl->h, impl->sh)) run_to(this, impl->h, 5, impl->stage, impl->fetched, m_timer);
}
void CSIFT3D::KpSiftAlgorithmAsync() {
	if (impl->sh) { KpSiftAlgorithm(); return; }  // (the sharded driver joins its rank threads: nothing to overlap with)
	if (!need_single("KpSiftAlgorithmAsync", impl->h, impl->sh)) return;
	const int rc = sift3d_run_async(impl->h);
	complain("KpSiftAlgorithmAsync", rc);
	impl->in_flight = rc == SIFT3D_OK;
}
void CSIFT3D::Wait() {
	if (!impl->h || !impl->in_flight) return;
	impl->in_flight = false;
	const int rc = sift3d_wait(impl->h);
	complain("Wait", rc);
	if (rc != SIFT3D_OK) return;
	impl->stage = 5;
	impl->fetched = false;
	double t[8];
	if (sift3d_stage_times(impl->h, t) == SIFT3D_OK) {
		m_timer.d_TotalTime = t[0]; m_timer.d_Allocation = t[1]; m_timer.d_BuildGSS = t[2]; m_timer.d_BuildDOG = t[3];
		m_timer.d_Detect = t[4]; m_timer.d_AssignOrientation = t[5]; m_timer.d_Extraction = t[6]; m_timer.d_release = t[7];
	}
}
void CSIFT3D::Initialize() {}
void CSIFT3D::Build_Gaussian_Scale_Space() { if (need_single("Build_Gaussian_Scale_Space", impl->h, impl->sh)) run_to(this, impl->h, 1, impl->stage, impl->fetched, m_timer); }
void CSIFT3D::Build_DOG_Scale_Space() { if (need_single("Build_DOG_Scale_Space", impl->h, impl->sh)) run_to(this, impl->h, 2, impl->stage, impl->fetched, m_timer); }
void CSIFT3D::Detect_KeyPoints() { if (need_single("Detect_KeyPoints", impl->h, impl->sh)) run_to(this, impl->h, 3, impl->stage, impl->fetched, m_timer); }
void CSIFT3D::Assign_Orientation() { if (need_single("Assign_Orientation", impl->h, impl->sh)) run_to(this, impl->h, 4, impl->stage, impl->fetched, m_timer); }
void CSIFT3D::Extract_Description() { if (need_single("Extract_Description", impl->h, impl->sh)) run_to(this, impl->h, 5, impl->stage, impl->fetched, m_timer); }
void CSIFT3D::Release_SIFT() { Gss_Pyramid.clear(); DoG_Pyramid.clear(); level_extrema.clear(); }

void CSIFT3D::SetNumThreads(int t_num) {
	if (t_num > 0) sift_thread_num = t_num;
}

std::vector<Keypoint> CSIFT3D::GetKeypoints() {
	Wait();
	if (!impl->fetched) fetch_results();
	return filter;
}

static void copy_pyramid(sift3d_handle h, int is_dog, int count, std::vector<TexImage> &out) {
	out.clear();
	out.resize((size_t)count);
	for (int i = 0; i < count; i++) {
		int d[3];
		float u[3], s;
		if (sift3d_level_info(h, is_dog, i, d, u, &s) != SIFT3D_OK) continue;
		TexImage &t = out[i];
		t.SetImageSize(d[0], d[1], d[2]);
		t.SetImageUnit(u[0], u[1], u[2]);
		t.SetImageScale(s);
		t.MallocArrayMemory();
		complain("GET_GSS/GET_DOG", sift3d_copy_level(h, is_dog, i, t._Data));
	}
}

std::vector<TexImage> *CSIFT3D::GET_GSS() {
	int noct = 0;
	if (need_single("GET_GSS", impl->h, impl->sh) && impl->stage >= 1 && sift3d_num_octaves(impl->h, &noct) == SIFT3D_OK)
		copy_pyramid(impl->h, 0, noct * (impl->levels + 3), Gss_Pyramid);
	return &Gss_Pyramid;
}

std::vector<TexImage> *CSIFT3D::GET_DOG() {
	int noct = 0;
	if (need_single("GET_DOG", impl->h, impl->sh) && impl->stage >= 1 && sift3d_num_octaves(impl->h, &noct) == SIFT3D_OK)
		copy_pyramid(impl->h, 1, noct * (impl->levels + 2), DoG_Pyramid);
	return &DoG_Pyramid;
}

std::vector<std::vector<Keypoint>> *CSIFT3D::GET_LEVEL() {
	level_extrema.clear();
	int n = 0, noct = 0;
	if (!need_single("GET_LEVEL", impl->h, impl->sh) || impl->stage < 3 || sift3d_num_extrema(impl->h, &n) != SIFT3D_OK) return &level_extrema;
	sift3d_num_octaves(impl->h, &noct);
	level_extrema.resize((size_t)noct * impl->levels);
	std::vector<sift3d_keypoint> pod((size_t)(n > 0 ? n : 1));
	if (n > 0 && sift3d_get_extrema(impl->h, pod.data()) != SIFT3D_OK) return &level_extrema;
	for (int i = 0; i < n; i++) {
		Keypoint k;
		k.x = pod[i].x; k.y = pod[i].y; k.z = pod[i].z; k.scale = pod[i].scale; k.octave = pod[i].octave; k.level = pod[i].level;
		k.rx = k.ry = k.rz = -1.0f;
		memset(k.eigvalue, 0, sizeof(k.eigvalue)); memset(k.eigvector, 0, sizeof(k.eigvector));
		memset(k.Rotation, 0, sizeof(k.Rotation)); memset(k.str_tensor, 0, sizeof(k.str_tensor));
		const size_t slot = (size_t)k.octave * impl->levels + (k.level - 1);
		if (slot < level_extrema.size()) level_extrema[slot].push_back(k);
	}
	return &level_extrema;
}

bool CSIFT3D::GetDeviceResults(const float **d_desc, const float **d_xyz, int *n, int *device) {
	if (!impl->h) return false;
	Wait();
	if (device) *device = impl->device;
	return sift3d_device_results(impl->h, d_desc, d_xyz, n) == SIFT3D_OK;
}

std::vector<CSIFT3D::PairMatch> CSIFT3D::AllPairsMatch(const std::vector<CSIFT3D *> &ex, double thresHold, int mode) {
	std::vector<PairMatch> out;
	const int n = (int)ex.size();
	for (CSIFT3D *e : ex) {
		if (!e || !e->impl || !e->impl->h) { fprintf(stderr, "[3dsift_amd] AllPairsMatch: every extractor must be a constructed single-GPU extractor\n"); return out; }
		e->Wait();
	}
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++)
			if (i != j) { PairMatch p; p.ref = i; p.tar = j; out.push_back(p); }
	// one host thread per reference GPU: the matcher serialises the calls of a device, different devices run side by side
	std::map<int, std::vector<size_t>> by_dev;
	for (size_t k = 0; k < out.size(); k++) by_dev[ex[(size_t)out[k].ref]->impl->device].push_back(k);
	auto work = [&](const std::vector<size_t> &mine) {
		for (size_t k : mine) {
			PairMatch &p = out[k];
			int nk = 0;
			sift3d_num_keypoints(ex[(size_t)p.ref]->impl->h, &nk);
			p.glodenIdx.assign((size_t)nk, -1);
			std::vector<float> pairs((size_t)(nk > 0 ? nk : 1) * 6);
			int np = 0;
			const int rc = sift3d_match_handles(ex[(size_t)p.ref]->impl->h, ex[(size_t)p.tar]->impl->h, thresHold, mode, p.glodenIdx.data(), nullptr, nullptr,
			                                    nullptr, pairs.data(), &np, &p.seconds);
			if (rc != SIFT3D_OK) { fprintf(stderr, "[3dsift_amd] AllPairsMatch (%d, %d): %s (%s)\n", p.ref, p.tar, sift3d_error_string(rc), sift3d_last_error()); continue; }
			for (int q = 0; q < np; q++) {
				p.refMatch.push_back(Cvec(pairs[6 * q], pairs[6 * q + 1], pairs[6 * q + 2]));
				p.tarMatch.push_back(Cvec(pairs[6 * q + 3], pairs[6 * q + 4], pairs[6 * q + 5]));
			}
		}
	};
	if (by_dev.size() <= 1) { if (!by_dev.empty()) work(by_dev.begin()->second); return out; }
	std::vector<std::thread> th;
	for (auto &kv : by_dev) th.emplace_back(work, std::cref(kv.second));
	for (auto &t : th) t.join();
	return out;
}

CSIFT3D *CSIFT3DFactory::CreateCSIFT3D(float *volume, int x_dim, int y_dim, int z_dim, int num_kp_levels, float sigma_default,
                                       float sigma_n_default, float peak_thresh, float max_eig_thres, float corner_thresh) {
	return new CSIFT3D(volume, x_dim, y_dim, z_dim, num_kp_levels, sigma_default, sigma_n_default, peak_thresh, max_eig_thres,
	                   corner_thresh);
}

CSIFT3D *CSIFT3DFactory::CreateCSIFT3D(std::string path_, int num_kp_levels, float sigma_default, float sigma_n_default,
                                       float peak_thresh, float max_eig_thres, float corner_thresh) {
	int m = 0, n = 0, p = 0;
	float *volume = nullptr;
	if (ReadMatrixFromDisk(path_.c_str(), &m, &n, &p, &volume) != 0 || !volume) {
		fprintf(stderr, "[3dsift_amd] CreateCSIFT3D: cannot read %s\n", path_.c_str());
		return new CSIFT3D();
	}
	CSIFT3D *s = new CSIFT3D(volume, m, n, p, num_kp_levels, sigma_default, sigma_n_default, peak_thresh, max_eig_thres, corner_thresh);
	free(volume);
	return s;
}

}  // namespace CPUSIFT

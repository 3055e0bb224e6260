// cMatcher.cpp -- CPUSIFT::muBruteMatcher over sift3d_match (include/sift3d_hip.h).
// Reference: 3DSIFT/Src/cMatcher.cc:146-228 (bijectMatchBase and its three public wrappers).
#include "../Include/cMatcher.h"

#include <atomic>
#include <cstdio>
#include <cstring>

#include "../../../include/sift3d_hip.h"

namespace CPUSIFT {

// The matcher's kernels, stream and scratch exist before the first call -- on the device the calls will use (SetDevice / SIFT3D_DEVICE), not
// on GPU 0 whatever was selected.  A matcher constructed as a static object warms nothing up (the HIP runtime may not be initialised yet, and
// SetDevice has not been called): warmed stays false and the first run() does it.
muBruteMatcher::muBruteMatcher() {}
float muBruteMatcher::getCalculationTime() { return totalTime; }
std::vector<float> muBruteMatcher::getGlodenDistSquare() { return glodenDistSquare; }
std::vector<float> muBruteMatcher::getSilverDistSquare() { return silverDistSquare; }
std::vector<int> muBruteMatcher::getGlodenIdx() { return glodenIdx; }
std::vector<int> muBruteMatcher::getSilverIdx() { return silverIdx; }

void muBruteMatcher::run(std::vector<Cvec> &refMatch, std::vector<Cvec> &tarMatch, const std::vector<Keypoint> &ref_kp,
                         const std::vector<Keypoint> &tar_kp, double thresHold, int mode) {
	const int n = (int)ref_kp.size(), m = (int)tar_kp.size();
	glodenIdx.assign((size_t)n, -1); silverIdx.assign((size_t)n, -1);
	glodenDistSquare.assign((size_t)n, 0.f); silverDistSquare.assign((size_t)n, 0.f);
	std::vector<float> pairs((size_t)(n > 0 ? n : 1) * 6);
	int np = 0, rc;
	double sec = 0;
	usedDeviceResults = false;
	{
		static std::atomic<int> warmed_device{-1};  // (one warm-up per device and process is enough: the matcher's stream and scratch are the library's)
		const int dev = GetDevice();
		if (warmed_device.load() != dev) { (void)sift3d_match_warmup(dev); warmed_device.store(dev); }
	}
	// Keypoint vectors that still alias their extractors (the Example flow: GetKeypoints() straight into the matcher) are matched
	// from the descriptors and coordinates already resident on the device: no gather, no host round trip of N x 768 floats
	CSIFT3D *oa = CSIFT3D::OwnerOf(ref_kp), *ob = CSIFT3D::OwnerOf(tar_kp);
	const float *da = nullptr, *xa = nullptr, *db = nullptr, *xb = nullptr;
	int na = 0, nb = 0, deva = -1, devb = -2;
	if (oa && ob && oa->GetDeviceResults(&da, &xa, &na, &deva) && ob->GetDeviceResults(&db, &xb, &nb, &devb) && deva == devb && na == n &&
	    nb == m) {
		usedDeviceResults = true;
		rc = sift3d_match(da, xa, n, db, xb, m, thresHold, mode, 1, deva, glodenIdx.data(), silverIdx.data(), glodenDistSquare.data(),
		                  silverDistSquare.data(), pairs.data(), &np, &sec);
	} else {
		// gather the (possibly scattered) descriptor rows; Keypoint::desc points into extractor-owned memory
		std::vector<float> a((size_t)n * DESC_LENGTH), b((size_t)m * DESC_LENGTH), ax((size_t)n * 3), bx((size_t)m * 3);
		for (int i = 0; i < n; i++) {
			if (ref_kp[i].desc) memcpy(&a[(size_t)i * DESC_LENGTH], ref_kp[i].desc, sizeof(float) * DESC_LENGTH);
			ax[3 * i] = ref_kp[i].rx; ax[3 * i + 1] = ref_kp[i].ry; ax[3 * i + 2] = ref_kp[i].rz;
		}
		for (int j = 0; j < m; j++) {
			if (tar_kp[j].desc) memcpy(&b[(size_t)j * DESC_LENGTH], tar_kp[j].desc, sizeof(float) * DESC_LENGTH);
			bx[3 * j] = tar_kp[j].rx; bx[3 * j + 1] = tar_kp[j].ry; bx[3 * j + 2] = tar_kp[j].rz;
		}
		rc = sift3d_match(a.data(), ax.data(), n, b.data(), bx.data(), m, thresHold, mode, 0, GetDevice(), glodenIdx.data(),
		                  silverIdx.data(), glodenDistSquare.data(), silverDistSquare.data(), pairs.data(), &np, &sec);
	}
	if (rc != SIFT3D_OK) {
		fprintf(stderr, "[3dsift_amd] muBruteMatcher: %s (%s)\n", sift3d_error_string(rc), sift3d_last_error());
		return;
	}
	for (int i = 0; i < np; i++) {
		refMatch.push_back(Cvec(pairs[6 * i], pairs[6 * i + 1], pairs[6 * i + 2]));
		tarMatch.push_back(Cvec(pairs[6 * i + 3], pairs[6 * i + 4], pairs[6 * i + 5]));
	}
	matchTime = totalTime = (float)sec;
}

void muBruteMatcher::injectMatch(std::vector<Cvec> &r, std::vector<Cvec> &t, const std::vector<Keypoint> &a,
                                 const std::vector<Keypoint> &b, const double th) { run(r, t, a, b, th, 1); }
void muBruteMatcher::bijectMatch(std::vector<Cvec> &r, std::vector<Cvec> &t, const std::vector<Keypoint> &a,
                                 const std::vector<Keypoint> &b, const double th) { run(r, t, a, b, th, 2); }
void muBruteMatcher::enhancedMatch(std::vector<Cvec> &r, std::vector<Cvec> &t, const std::vector<Keypoint> &a,
                                   const std::vector<Keypoint> &b, const double th) { run(r, t, a, b, th, 3); }

}  // namespace CPUSIFT

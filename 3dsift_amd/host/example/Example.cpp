// Example.cpp -- the reference's demo flow (3DSIFT/Example.cpp:8-64) on this build: load (or
// synthesise) two volumes, extract, enhancedMatch(0.85), print the pairs.
//   example_sift3d ref.nii[.gz] tar.nii[.gz]          NIfTI inputs
//   example_sift3d --raw ref.bin tar.bin              raw matrix files (12-byte header + fp32)
//   example_sift3d --synth N                          two N^3 synthetic volumes (second shifted by one voxel)
#include <cmath>
#include <cstring>
#include <vector>

#include "../Include/Util/matrixIO3D.h"
#include "../Include/Util/readNii.h"
#include "../Include/cMatcher.h"
#include "../Include/cSIFT3D.h"

using namespace std;

static float *synth(int n, float shift) {
	float *v = new float[(size_t)n * n * n]();
	unsigned s = 12345u;
	auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
	const int blobs = max(8, n * n * n / 4096);
	for (int b = 0; b < blobs; b++) {
		const float cx = rnd() * n + shift, cy = rnd() * n, cz = rnd() * n, sg = 1.5f + 3.0f * rnd(), am = 0.3f + rnd();
		const int r = (int)ceilf(5 * sg);
		for (int z = max(0, (int)cz - r); z <= min(n - 1, (int)cz + r); z++)
			for (int y = max(0, (int)cy - r); y <= min(n - 1, (int)cy + r); y++)
				for (int x = max(0, (int)cx - r); x <= min(n - 1, (int)cx + r); x++) {
					const float d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz);
					v[(size_t)x + (size_t)n * (y + (size_t)n * z)] += am * expf(-0.5f * d2 / (sg * sg));
				}
	}
	return v;
}

int main(int argc, char **argv) {
	int nx = 0, ny = 0, nz = 0, nxT = 0, nyT = 0, nzT = 0;
	float *refVol = nullptr, *tarVol = nullptr;
	bool raw = false;
	if (argc >= 3 && strcmp(argv[1], "--synth") == 0) {
		nx = ny = nz = nxT = nyT = nzT = atoi(argv[2]);
		refVol = synth(nx, 0.f);
		tarVol = synth(nx, 1.f);
	} else if (argc >= 4 && strcmp(argv[1], "--raw") == 0) {
		raw = true;
		if (ReadMatrixFromDisk(argv[2], &nx, &ny, &nz, &refVol) || ReadMatrixFromDisk(argv[3], &nxT, &nyT, &nzT, &tarVol)) return 2;
	} else if (argc >= 3) {
		refVol = readNiiFile(argv[1], nx, ny, nz);
		tarVol = readNiiFile(argv[2], nxT, nyT, nzT);
	} else {
		cerr << "usage: " << argv[0] << " ref.nii tar.nii | --raw ref.bin tar.bin | --synth N" << endl;
		return 1;
	}
	if (!refVol || !tarVol) return 2;
	cout << "Dimensions of reference image:" << nx << " " << ny << " " << nz << endl;

	auto SIFT_ref = CPUSIFT::CSIFT3DFactory::CreateCSIFT3D(refVol, nx, ny, nz);
	SIFT_ref->KpSiftAlgorithm();
	auto vRefKp = SIFT_ref->GetKeypoints();
	cout << SIFT_ref->m_timer;

	cout << "Dimensions of target image:" << nxT << " " << nyT << " " << nzT << endl;
	auto SIFT_tar = CPUSIFT::CSIFT3DFactory::CreateCSIFT3D(tarVol, nxT, nyT, nzT);
	SIFT_tar->KpSiftAlgorithm();
	auto vTarKp = SIFT_tar->GetKeypoints();

	CPUSIFT::muBruteMatcher matcher;
	vector<CPUSIFT::Cvec> matchRefCoor, matchTarCoor;
	matcher.enhancedMatch(matchRefCoor, matchTarCoor, vRefKp, vTarKp, 0.85f);

	cout << "keypoints: " << vRefKp.size() << " / " << vTarKp.size() << ", matched pairs: " << matchRefCoor.size() << endl;
	cout << "Matched Points: reference coordinate(x,y,z);target coordinate(x,y,z)" << endl;
	for (size_t i = 0; i < matchRefCoor.size(); ++i)
		cout << matchRefCoor[i].x << "," << matchRefCoor[i].y << "," << matchRefCoor[i].z << ";" << matchTarCoor[i].x << ","
		     << matchTarCoor[i].y << "," << matchTarCoor[i].z << endl;

	if (raw) { free(refVol); free(tarVol); } else { delete[] refVol; delete[] tarVol; }
	delete SIFT_ref;
	delete SIFT_tar;
	return 0;
}

// asan_check.cpp -- driver of the ASan + UBSan build of the host-side code (make -C 3dsift_amd/host asan; tests/test_sanitizers_cpu.py).
//
//   asan_check <dir>
// <dir> holds files written by the test: good_*.nii[.gz] / .hdr + .img (every supported datatype / byte order / gzip / header version), bad_* (malformed
// headers, truncated payloads, random bytes), m.bin (raw matrix).  Every reader is run on every file; results are printed so the
// test can compare them; the sanitizers abort the process on any out-of-bounds access, use-after-free or undefined behaviour.
// Also walks the no-device error paths of the shell classes (on a box without a GPU the constructor fails loudly and every later
// call must stay defined and return empty results, like the reference after a failed allocation).
#include <dirent.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../Include/Util/matrixIO3D.h"
#include "../Include/Util/readNii.h"
#include "../Include/cMatcher.h"
#include "../Include/cSIFT3D.h"
#include "../Include/cUtil.h"

using namespace CPUSIFT;

static double checksum(const float *v, size_t n) {
	double s = 0;
	for (size_t i = 0; i < n; i++) s += (double)v[i] * (double)((i % 97) + 1);
	return s;
}

int main(int argc, char **argv) {
	if (argc < 2) return 2;
	const std::string dir = argv[1];
	std::vector<std::string> names;
	if (DIR *d = opendir(dir.c_str())) {
		while (dirent *e = readdir(d)) names.push_back(e->d_name);
		closedir(d);
	}
	for (const std::string &n : names) {
		if (n.find(".nii") == std::string::npos && n.find(".hdr") == std::string::npos && n.find(".img") == std::string::npos) continue;
		int nx = -1, ny = -1, nz = -1;
		float *v = readNiiFile((dir + "/" + n).c_str(), nx, ny, nz);
		if (v) {
			printf("nii %s %d %d %d %.9g\n", n.c_str(), nx, ny, nz, checksum(v, (size_t)nx * ny * nz));
			delete[] v;
		} else {
			printf("nii %s rejected\n", n.c_str());
		}
	}
	// raw matrix: read, write back, read again; plus a truncated file and a header with non-positive sizes
	{
		int m = 0, n = 0, p = 0;
		float *v = nullptr;
		if (ReadMatrixFromDisk((dir + "/m.bin").c_str(), &m, &n, &p, &v) == 0 && v) {
			printf("matrix %d %d %d %.9g\n", m, n, p, checksum(v, (size_t)m * n * p));
			WriteMatrixToDisk((dir + "/m2.bin").c_str(), m, n, p, v);
			free(v);
			v = nullptr;
			int m2 = 0, n2 = 0, p2 = 0;
			if (ReadMatrixFromDisk((dir + "/m2.bin").c_str(), &m2, &n2, &p2, &v) == 0 && v) {
				printf("matrix2 %d %d %d %.9g\n", m2, n2, p2, checksum(v, (size_t)m2 * n2 * p2));
				free(v);
			}
		}
		v = nullptr;
		printf("matrix_trunc %d\n", ReadMatrixFromDisk((dir + "/m_trunc.bin").c_str(), &m, &n, &p, &v));
		printf("matrix_bad %d\n", ReadMatrixFromDisk((dir + "/m_bad.bin").c_str(), &m, &n, &p, &v));
		printf("matrix_missing %d\n", ReadMatrixFromDisk((dir + "/nope.bin").c_str(), &m, &n, &p, &v));
	}
	// key-point CSV
	{
		std::vector<Cvec> a = {{1.5f, 2.25f, 3.0f}, {10.123456f, 0.f, -4.5f}, {1e6f, -1e-6f, 7.f}}, b, c;
		write_sift_kp(a, (dir + "/kp.csv").c_str());
		read_sift_kp((dir + "/kp.csv").c_str(), b);
		printf("csv %zu %.5f %.5f %.5f\n", b.size(), b.size() == 3 ? b[1].x : 0.f, b.size() == 3 ? b[1].y : 0.f, b.size() == 3 ? b[1].z : 0.f);
		read_sift_kp((dir + "/kp_garbage.csv").c_str(), c);
		printf("csv_garbage %zu\n", c.size());
		read_sift_kp((dir + "/nope.csv").c_str(), c);
	}
	// per-voxel / per-keypoint free functions (r05): the host scatter on cell coordinates at and beyond the block's faces, vanishing
	// gradients; the device-backed ones report "no device" here (or refuse bad inputs) and must leave their outputs defined
	{
		Mesh mesh;
		Initialize_geometry(&mesh);
		std::vector<float> d(DESC_NUMEL, 0.f), dv(3 * 64), br(3 * 64), acc(24 * 64, 0.f);
		std::vector<int> face(64), off(24 * 64, -1);
		Keypoint k;
		k.desc = d.data();
		int i = 0;
		for (float bx : {-0.5f, -0.25f, 0.f, 2.75f, 3.f, 3.49f, 4.f, -1.f})
			for (float gz : {0.f, 1.f, -2.f, 1e-9f}) {
				Cvec vb(bx, 3.25f - bx, 0.5f * bx), g(0.3f * gz, -gz, 0.5f * gz);
				Trilinear_interpolation_over_desc(&mesh, k, vb, g, i);
				Trilinear_interpolation_over_desc_debug(&mesh, k, vb, g, i, dv.data(), face.data(), br.data(), off.data(), acc.data(), 1);
				i++;
			}
		double sum = 0;
		for (float v : d) sum += v;
		printf("scatter %d voxels sum %.6f\n", i, sum);
		free(mesh.tri);
		TexImage lvl, out;
		lvl.SetImageSize(12, 10, 9);
		lvl.MallocArrayMemory();
		for (int j = 0; j < 12 * 10 * 9; j++) lvl._Data[j] = (float)(j % 17) * 0.01f;
		lvl.SetImageUnit(1.f, 1.f, 1.f);
		float w[4] = {0.25f, 0.5f, 0.25f, 0.f};
		GaussianSmooth_3D_Imp(&lvl, &out, 1, 1.f, w, 3);
		GaussianSmooth_3D_Imp(&lvl, &out, 1, 1.f, w, 4);  // even width: refused
		Keypoint q;
		q.x = 5; q.y = 4; q.z = 4; q.scale = 1.6f; q.desc = d.data();
		printf("orient code %d\n", Assign_Orientation_Imp(q, &lvl, 2.4f, 0.9f, 0.4f));
		Extract_Descriptor_Imp(q, &lvl, nullptr);
		q.x = 5.5f;
		printf("orient off-voxel code %d\n", Assign_Orientation_Imp(q, &lvl, 2.4f, 0.9f, 0.4f));
	}
	// shell classes: with a GPU this is a tiny real run; without one every call reports the error and returns empty results
	{
		std::vector<float> vol(24 * 20 * 28);
		for (size_t i = 0; i < vol.size(); i++) vol[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
		CSIFT3D *ex = CSIFT3DFactory::CreateCSIFT3D(vol.data(), 24, 20, 28);
		ex->KpSiftAlgorithm();
		std::vector<Keypoint> kp = ex->GetKeypoints();
		muBruteMatcher m;
		std::vector<Cvec> r, t;
		m.enhancedMatch(r, t, kp, kp, 0.85);
		printf("shell keypoints %zu pairs %zu\n", kp.size(), r.size());
		CSIFT3D *none = CSIFT3DFactory::CreateCSIFT3D(dir + "/nope.bin");
		none->KpSiftAlgorithm();
		printf("shell missing-file keypoints %zu\n", none->GetKeypoints().size());
		delete none;
		delete ex;
	}
	printf("asan_check done\n");
	return 0;
}

// helpers_check.cpp -- CPU test driver of the header surface the shell shares with the reference (tests/test_host_shell.py):
// reads directions from stdin, prints the mesh and Check_intersect_faces / cart2bary results (compared with golden g7, which
// the reference produced), and exercises the TexImage members, the typedefs and the templated matrix IO.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "Include/cSIFT3D.h"
#include "Include/Util/matrixIO3D.h"

using namespace CPUSIFT;

int main(int argc, char **argv) {
	Mesh mesh;
	if (Initialize_geometry(&mesh) != 0 || mesh.num != ICOS_NFACES) return 2;
	for (int f = 0; f < mesh.num; f++) {
		const Tri &t = mesh.tri[f];
		printf("tri %d %d %d", t.idx[0], t.idx[1], t.idx[2]);
		for (int j = 0; j < 3; j++) {
			unsigned b[3];
			memcpy(b, &t.v[j], 12);
			printf(" %08x %08x %08x", b[0], b[1], b[2]);
		}
		printf("\n");
	}
	int n = 0;
	if (scanf("%d", &n) != 1) return 3;
	for (int i = 0; i < n; i++) {
		unsigned b[3];
		if (scanf("%x %x %x", &b[0], &b[1], &b[2]) != 3) return 4;
		Cvec g, bary(0, 0, 0);
		memcpy(&g, b, 12);
		const int face = Check_intersect_faces(&mesh, &g, &bary);
		unsigned o[3];
		memcpy(o, &bary, 12);
		printf("dir %d %08x %08x %08x\n", face, o[0], o[1], o[2]);
	}
	free(mesh.tri);

	// TexImage surface (Include/Util/cTexImage.h:15-32 of the reference)
	TexImage a(4, 3, 2);
	printf("tex %zu %d %d %d %g %g\n", a._numsize, a.GetDimX(), a.GetDimY(), a.GetDimZ(), (double)a.GetScale(), (double)a.GetUnitX());
	float buf[24];
	for (int i = 0; i < 24; i++) buf[i] = (float)i;
	a.SetImageDataPt(buf);
	printf("view %g %g\n", (double)a.GetImageDataWithIdx(3, 2, 1), (double)a.GetImageDataWithIdx(1, 0, 1));
	TexImage p;
	Im_permute(&a, &p, 0, 2);
	printf("perm %d %d %d %g\n", p.GetDimX(), p.GetDimY(), p.GetDimZ(), (double)p.GetImageDataWithIdx(1, 2, 3));
	a.ReSetImageSize(2, 2, 2);
	printf("reset %zu %d %d\n", a._numsize, a.GetDimX(), a._Data == nullptr);
	a.SetImageSize(2, 2, 2);
	printf("nvox %zu\n", a._numsize);
	float R[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
	Transpose_Matrix(R);
	printf("tr %g %g %g %d %d\n", (double)R[1], (double)R[2], (double)R[5], (int)DistinctEig(1.f, 2.f, 3.f), (int)DistinctEig(1.f, 1.f, 3.f));
	Image im; EigenVal ev; (void)im; (void)ev;
	ev.val = 1.f; im.nx = 1;

	// templated matrix IO: a double and an int volume round-trip (the reference's templates, Include/Util/matrixIO3D.h:21-112)
	if (argc > 1) {
		const std::string dir = argv[1];
		double dv[6] = {0.5, 1.5, 2.5, 3.5, 4.5, 5.5};
		int iv[6] = {1, 2, 3, 4, 5, 6};
		int m, n2, q;
		double *dr = nullptr;
		int *ir = nullptr;
		int rc = WriteMatrixToDisk((dir + "/d.bin").c_str(), 3, 2, 1, dv) | WriteMatrixToDisk((dir + "/i.bin").c_str(), 1, 2, 3, iv);
		rc |= ReadMatrixFromDisk((dir + "/d.bin").c_str(), &m, &n2, &q, &dr);
		printf("dmat %d %d %d %d %g\n", rc, m, n2, q, dr ? dr[5] : -1.0);
		rc = ReadMatrixFromDisk((dir + "/i.bin").c_str(), &m, &n2, &q, &ir);
		printf("imat %d %d %d %d %d\n", rc, m, n2, q, ir ? ir[5] : -1);
		rc = ReadMatrixSizeFromDisk((dir + "/i.bin").c_str(), &m, &n2, &q);
		printf("size %d %d %d %d\n", rc, m, n2, q);
		free(dr); free(ir);
	}
	return 0;
}

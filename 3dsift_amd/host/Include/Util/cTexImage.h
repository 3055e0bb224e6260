// cTexImage.h -- host-side dense fp32 volume used only by the checking accessors GET_GSS / GET_DOG
// (reference Include/Util/cTexImage.h:8-60).  x fastest: idx = x + y*nx + z*nx*ny.  In this build the
// pyramids live on the GPU; a TexImage is a host COPY made on demand.
#pragma once

#include <cstddef>
#include <vector>

class TexImage {
public:
	float *_Data = nullptr;  // points into _store
	int _nx = 0, _ny = 0, _nz = 0;
	float _s = 0.f;
	size_t _xs = 1, _ys = 0, _zs = 0;
	float _ux = 1.f, _uy = 1.f, _uz = 1.f;

	TexImage() = default;
	TexImage(int width, int height, int depth) { SetImageSize(width, height, depth); }
	TexImage(const TexImage &o) { *this = o; }
	TexImage &operator=(const TexImage &o) {
		_store = o._store;
		_nx = o._nx; _ny = o._ny; _nz = o._nz; _s = o._s;
		_xs = o._xs; _ys = o._ys; _zs = o._zs; _ux = o._ux; _uy = o._uy; _uz = o._uz;
		_Data = _store.empty() ? nullptr : _store.data();
		return *this;
	}

	void SetImageSize(int width, int height, int depth) {
		_nx = width; _ny = height; _nz = depth;
		_xs = 1; _ys = (size_t)width; _zs = (size_t)width * height;
	}
	void SetImageScale(float scale) { _s = scale; }
	void SetImageUnit(float ux, float uy, float uz) { _ux = ux; _uy = uy; _uz = uz; }
	void MallocArrayMemory() {
		_store.assign((size_t)_nx * _ny * _nz, 0.f);
		_Data = _store.data();
	}
	void SetImageDataWithIdx(float v, int x, int y, int z) { _Data[x * _xs + y * _ys + z * _zs] = v; }

	float GetScale() { return _s; }
	size_t GetXstride() { return _xs; }
	size_t GetYstride() { return _ys; }
	size_t GetZstride() { return _zs; }
	float GetUnitX() { return _ux; }
	float GetUnitY() { return _uy; }
	float GetUnitZ() { return _uz; }
	int GetDimX() { return _nx; }
	int GetDimY() { return _ny; }
	int GetDimZ() { return _nz; }
	float GetImageDataWithIdx(int x, int y, int z) { return _Data[x * _xs + y * _ys + z * _zs]; }

private:
	std::vector<float> _store;
};

// cTexImage.h -- host-side dense fp32 volume used only by the checking accessors GET_GSS / GET_DOG
// (reference Include/Util/cTexImage.h:8-60).  x fastest: idx = x + y*nx + z*nx*ny.  In this build the
// pyramids live on the GPU; a TexImage is a host COPY made on demand.
#pragma once

#include <cstddef>
#include <vector>

class TexImage {
public:
	float *_Data = nullptr;  // points into _store, or at the caller's buffer after SetImageDataPt
	size_t _numsize = 0;     // as the reference leaves it: BYTES after the sizing constructor / ReSetImageSize, VOXELS after SetImageSize
	int _nx = 0, _ny = 0, _nz = 0;
	float _s = 0.f;
	size_t _xs = 1, _ys = 0, _zs = 0;
	float _ux = 1.f, _uy = 1.f, _uz = 1.f;

	TexImage() = default;
	TexImage(int width, int height, int depth) { ReSetImageSize(width, height, depth); }
	TexImage(const TexImage &o) { *this = o; }
	TexImage &operator=(const TexImage &o) {
		const bool external = o._Data != nullptr && (o._store.empty() || o._Data != o._store.data());
		_store = o._store;
		_nx = o._nx; _ny = o._ny; _nz = o._nz; _s = o._s; _numsize = o._numsize;
		_xs = o._xs; _ys = o._ys; _zs = o._zs; _ux = o._ux; _uy = o._uy; _uz = o._uz;
		_Data = external ? o._Data : (_store.empty() ? nullptr : _store.data());  // a view stays a view of the same buffer
		return *this;
	}

	void SetImageSize(int width, int height, int depth) {
		_nx = width; _ny = height; _nz = depth;
		_xs = 1; _ys = (size_t)width; _zs = (size_t)width * height;
		_numsize = (size_t)width * height * depth;
	}
	// reference Src/Util/cTexImage.cc:38-53: new dimensions, data pointer dropped, scale 0, units 1
	void ReSetImageSize(int width_new, int height_new, int depth_new) {
		_store.clear(); _Data = nullptr;
		SetImageSize(width_new, height_new, depth_new);
		_numsize *= sizeof(float);
		_s = 0.f; _ux = _uy = _uz = 1.f;
	}
	// reference Src/Util/cTexImage.cc:95-98: the image becomes a VIEW of the caller's buffer (x fastest, _nx*_ny*_nz floats).  Unlike
	// the reference, whose destructor frees whatever _Data points at, the caller keeps ownership here.
	void SetImageDataPt(float *data) { _store.clear(); _Data = data; }
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

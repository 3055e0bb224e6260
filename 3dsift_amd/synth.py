"""Deterministic synthetic volumes (BASELINE.md section 3 / SURVEY.md section 8d).

Sum of B = max(8, V/4096) isotropic Gaussian blobs: centres U[0,n) per axis,
sigma_b ~ U[1.5,4.5], amplitude ~ U[0.3,1.3], support cut at 5 sigma_b; optional
U[0,noise) noise; fp32, indexed [z, y, x] (x fastest, like the reference's TexImage).
``shift`` moves every blob centre (used for the matching target: +1 voxel in x).

The parameter stream comes from numpy's PCG64 (stable across numpy versions), drawn in
one fixed order, so every caller (tests, golden generator, bench) sees identical volumes.
``blobs_torch`` renders the same parameter list on a torch device for the large bench
sizes; its voxels may differ from the numpy rendering in the last bits (device exp), so a
run must hand ONE rendering to both the HIP path and the CPU checker.
"""
import numpy as np


def blob_params(shape, seed=1234, nblobs=None):
    nz, ny, nx = shape
    v = nx * ny * nz
    b = max(8, v // 4096) if nblobs is None else nblobs
    rng = np.random.Generator(np.random.PCG64(seed))
    cx = rng.uniform(0, nx, b)
    cy = rng.uniform(0, ny, b)
    cz = rng.uniform(0, nz, b)
    sg = rng.uniform(1.5, 4.5, b)
    am = rng.uniform(0.3, 1.3, b)
    return cx, cy, cz, sg, am


def blobs(shape, seed=1234, shift=(0.0, 0.0, 0.0), noise=0.0, noise_seed=99, nblobs=None):
    """numpy rendering; shape = (nz, ny, nx); shift = (dx, dy, dz)."""
    nz, ny, nx = shape
    cx, cy, cz, sg, am = blob_params(shape, seed, nblobs)
    vol = np.zeros(shape, np.float64)
    for i in range(len(cx)):
        x0, y0, z0, s, a = cx[i] + shift[0], cy[i] + shift[1], cz[i] + shift[2], sg[i], am[i]
        r = 5.0 * s
        xl, xh = max(0, int(np.floor(x0 - r))), min(nx - 1, int(np.ceil(x0 + r)))
        yl, yh = max(0, int(np.floor(y0 - r))), min(ny - 1, int(np.ceil(y0 + r)))
        zl, zh = max(0, int(np.floor(z0 - r))), min(nz - 1, int(np.ceil(z0 + r)))
        if xl > xh or yl > yh or zl > zh:
            continue
        gx = np.exp(-0.5 * ((np.arange(xl, xh + 1) - x0) / s) ** 2)
        gy = np.exp(-0.5 * ((np.arange(yl, yh + 1) - y0) / s) ** 2)
        gz = np.exp(-0.5 * ((np.arange(zl, zh + 1) - z0) / s) ** 2)
        vol[zl:zh + 1, yl:yh + 1, xl:xh + 1] += a * gz[:, None, None] * gy[None, :, None] * gx[None, None, :]
    if noise > 0:
        vol += np.random.Generator(np.random.PCG64(noise_seed)).uniform(0, noise, shape)
    return vol.astype(np.float32)


def blobs_torch(shape, device, seed=1234, shift=(0.0, 0.0, 0.0), nblobs=None):
    """Same blob list rendered on a torch device (fp32 accumulate); returns a torch tensor [z,y,x]."""
    import torch

    nz, ny, nx = shape
    cx, cy, cz, sg, am = blob_params(shape, seed, nblobs)
    vol = torch.zeros(shape, dtype=torch.float32, device=device)
    ax = torch.arange(max(shape), dtype=torch.float32, device=device)
    for i in range(len(cx)):
        x0, y0, z0, s, a = cx[i] + shift[0], cy[i] + shift[1], cz[i] + shift[2], float(sg[i]), float(am[i])
        r = 5.0 * s
        xl, xh = max(0, int(np.floor(x0 - r))), min(nx - 1, int(np.ceil(x0 + r)))
        yl, yh = max(0, int(np.floor(y0 - r))), min(ny - 1, int(np.ceil(y0 + r)))
        zl, zh = max(0, int(np.floor(z0 - r))), min(nz - 1, int(np.ceil(z0 + r)))
        if xl > xh or yl > yh or zl > zh:
            continue
        gx = torch.exp(-0.5 * ((ax[xl:xh + 1] - float(x0)) / s) ** 2)
        gy = torch.exp(-0.5 * ((ax[yl:yh + 1] - float(y0)) / s) ** 2)
        gz = torch.exp(-0.5 * ((ax[zl:zh + 1] - float(z0)) / s) ** 2) * a
        vol[zl:zh + 1, yl:yh + 1, xl:xh + 1] += gz[:, None, None] * gy[None, :, None] * gx[None, None, :]
    return vol

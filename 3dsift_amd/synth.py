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


def blobs_torch(shape, device, seed=1234, shift=(0.0, 0.0, 0.0), nblobs=None, brick=64, zrange=None):
    """Same blob list rendered on a torch device, brick by brick: every brick accumulates the blobs
    whose 5-sigma boxes overlap it as one small contraction sum_k gz[k,z] gy[k,y] gx[k,x] (a few
    thousand kernels instead of ~20 per blob).  fp32; returns a torch tensor [z, y, x].
    zrange = (za, zb): render only those planes of the volume (z-slab of a sharded volume), tensor [zb-za, y, x]."""
    import torch

    nz, ny, nx = shape
    cx, cy, cz, sg, am = blob_params(shape, seed, nblobs)
    cx = cx + shift[0]; cy = cy + shift[1]; cz = cz + shift[2]
    r = 5.0 * sg
    lo = [np.maximum(0, np.floor(c - r)).astype(np.int64) for c in (cx, cy, cz)]
    hi = [np.minimum(n - 1, np.ceil(c + r)).astype(np.int64) for c, n in ((cx, nx), (cy, ny), (cz, nz))]
    za, zb = (0, nz) if zrange is None else zrange
    vol = torch.zeros((zb - za, ny, nx), dtype=torch.float32, device=device)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
    for z0 in range(za, zb, brick):
        z1 = min(zb, z0 + brick)
        mz = (hi[2] >= z0) & (lo[2] < z1)
        for y0 in range(0, ny, brick):
            y1 = min(ny, y0 + brick)
            mzy = mz & (hi[1] >= y0) & (lo[1] < y1)
            for x0 in range(0, nx, brick):
                x1 = min(nx, x0 + brick)
                idx = np.nonzero(mzy & (hi[0] >= x0) & (lo[0] < x1))[0]
                if len(idx) == 0:
                    continue

                def factor(c, s_, l, h, a0, a1, amp=None):
                    ax = torch.arange(a0, a1, dtype=torch.float32, device=device)[None, :]
                    g = torch.exp(-0.5 * ((ax - t(c)[:, None]) / t(s_)[:, None]) ** 2)
                    g = g * ((ax >= t(l)[:, None]) & (ax <= t(h)[:, None]))
                    return g if amp is None else g * t(amp)[:, None]

                gx = factor(cx[idx], sg[idx], lo[0][idx], hi[0][idx], x0, x1)
                gy = factor(cy[idx], sg[idx], lo[1][idx], hi[1][idx], y0, y1)
                gz = factor(cz[idx], sg[idx], lo[2][idx], hi[2][idx], z0, z1, am[idx])
                yx = (gy[:, :, None] * gx[:, None, :]).reshape(len(idx), -1)
                vol[z0 - za:z1 - za, y0:y1, x0:x1] = (gz.t() @ yx).reshape(z1 - z0, y1 - y0, x1 - x0)
    return vol

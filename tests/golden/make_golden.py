#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference.

Runs only in the build container (needs oracle/_ref/libref3dsift.so, i.e. /root/reference compiled by
``make -C oracle ref``).  The fixtures are data: inputs (or the seed that regenerates them plus a
sha256 of the resulting array) and the outputs the untouched reference produced for them.  No
reference source text is stored.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Fixture inventory (the reference has no golden vectors of its own):
  g2_smooth.npz     GaussianSmooth_3D on a 20x14x10 ramp+noise volume, sigma in {0.538701, 2.452547}
                    (both boundary rules on all three axes), plus a 9x12x7 volume with sigma 1.2262
  g3_pyramid.npz    full GSS + DoG pyramids of a 24x20x28 volume (2 octaves) -- raw fp32
  g4_hashes.npz     sha256 of every GSS/DoG level, per-level abs-max, extrema lists for 40x48x56
                    (3 octaves, last 10x12x14: no out-of-bounds reads) and 64^3 (4 octaves; the
                    8^3 level-5 shell is undefined in the reference and excluded from the hash)
  g5_orient.npz     Assign_Orientation_Imp result for every extremum of the 40x48x56 volume
                    (return code, win, eigvalue, str_tensor, Rotation)
  g6_keypoints.npz  final keypoints + 768-d descriptors for 40x48x56 and 64^3
  g7_mesh.npz       icosahedron mesh + Check_intersect_faces on 1000 random vectors (+ face
                    vertices / edge midpoints, where the eps-tolerant first-hit rule matters)
  g8_match.npz      muBruteMatcher inject/biject/enhanced on two 64^3 keypoint sets (target = blobs
                    shifted +1 voxel in x), plus a permuted copy that puts a best match on index 0
  g1_taps.npz       impulse responses of GaussianSmooth_3D (centre line along x) for the six sigmas of the default schedule and a
                    few others: they pin the 1-D tap vectors of Src/cSIFT3D.cc:546-572, which have no accessor (G1)
  g9_nifti.npz      small NIfTI-1 files (bytes) of several datatypes / byte orders / gzip, with non-unit scl_slope, and the
                    float arrays the reference's readNiiFile (oracle/_ref/librefnii.so) returns for them

  g10_sift_kp.npz   a key-point coordinate list written by the reference's write_sift_kp and read back by its read_sift_kp (8f-4)

    python tests/golden/make_golden.py g1 g9 g10     # only these
"""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as ol  # noqa: E402

synth = importlib.import_module("3dsift_amd.synth")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ramp_noise(shape, seed):
    nz, ny, nx = shape
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    rng = np.random.Generator(np.random.PCG64(seed))
    return (0.01 * x + 0.02 * y - 0.015 * z + rng.uniform(-0.5, 0.5, shape)).astype(np.float32)


def interior_mask_hash(level, shell):
    """sha256 of a level; with shell=True only the [1:-1]^3 interior is hashed."""
    return sha(level[1:-1, 1:-1, 1:-1]) if shell else sha(level)


def nifti1_bytes(vol, dtype, code, big_endian=False, slope=2.0, inter=5.0, vox_offset=352.0, hdr_bytes=352):
    """a single-file NIfTI-1 image (348-byte header + 4 extension bytes + payload), written by hand; vox_offset / hdr_bytes let a
    fixture imitate lax writers (offset 0 or 348 with the payload straight behind the 348-byte header)"""
    import struct
    nz, ny, nx = vol.shape
    e = ">" if big_endian else "<"
    h = bytearray(352)
    struct.pack_into(e + "i", h, 0, 348)
    struct.pack_into(e + "8h", h, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(e + "h", h, 70, code)
    struct.pack_into(e + "h", h, 72, np.dtype(dtype).itemsize * 8)
    struct.pack_into(e + "8f", h, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)  # pixdim
    struct.pack_into(e + "f", h, 108, vox_offset)
    struct.pack_into(e + "f", h, 112, slope)
    struct.pack_into(e + "f", h, 116, inter)
    h[344:348] = b"n+1\0"
    return bytes(h[:hdr_bytes]) + vol.astype(np.dtype(dtype).newbyteorder(e)).tobytes()


def nifti2_header(vol, dtype, code, big_endian=False, magic=b"n+2\0\r\n\032\n", vox_offset=544, slope=2.0, inter=5.0):
    """the 540-byte NIfTI-2 header (+ 4 extension bytes) written by hand"""
    import struct
    nz, ny, nx = vol.shape
    e = ">" if big_endian else "<"
    h = bytearray(544)
    struct.pack_into(e + "i", h, 0, 540)
    h[4:12] = magic
    struct.pack_into(e + "2h", h, 12, code, np.dtype(dtype).itemsize * 8)
    struct.pack_into(e + "8q", h, 16, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(e + "8d", h, 104, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)  # pixdim
    struct.pack_into(e + "q", h, 168, vox_offset)
    struct.pack_into(e + "2d", h, 176, slope, inter)
    return bytes(h)


def payload(vol, dtype, big_endian=False):
    return vol.astype(np.dtype(dtype).newbyteorder(">" if big_endian else "<")).tobytes()


def make_g1(ref):
    """The reference builds its taps inline (no accessor), so G1 pins them through the impulse response of GaussianSmooth_3D: for a unit
    impulse the x pass leaves tap[d] exactly, the y and z passes multiply by the centre tap: line[d] = rn(tc * rn(tc * tap[d]))."""
    g1 = {}
    # default schedule (sigma_default 1.6, sigma_n 1.15, 3 keypoint levels) + odd ones
    sig = [1.112430, 1.226273, 1.545008, 1.946588, 2.452547, 3.090016, 0.538701, 0.8, 4.0, 5.5]
    for i, s_ in enumerate(sig):
        n = 2 * 24 + 5
        v = np.zeros((n, n, n), np.float32)
        c = n // 2
        v[c, c, c] = 1.0
        out = ref.gaussian_smooth(v, np.float32(s_))
        g1[f"sigma_{i}"] = np.float32(s_)
        g1[f"line_{i}"] = out[c, c, :].copy()
    np.savez_compressed(os.path.join(HERE, "g1_taps.npz"), **g1)


def make_g9():
    import ctypes as C
    import gzip
    import tempfile
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librefnii.so"))
    fn = lib._Z11readNiiFilePKcRiS1_S1_
    fn.restype = C.POINTER(C.c_float)
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.Generator(np.random.PCG64(9))
    base = (rng.random((5, 6, 7)) * 200 - 60)
    cases = [("f4_le", "f4", 16, False, False), ("f4_be", "f4", 16, True, False), ("f4_gz", "f4", 16, False, True),
             ("i2_le", "i2", 4, False, False), ("i2_be", "i2", 4, True, False), ("u1_le", "u1", 2, False, False),
             ("f8_le", "f8", 64, False, False), ("f8_be", "f8", 64, True, False), ("u2_le", "u2", 512, False, False),
             ("i4_le", "i4", 8, False, False), ("i1_le", "i1", 256, False, False), ("u4_gz", "u4", 768, False, True),
             # lax writers: vox_offset below the header size is read as 348 by the reference (nifti2_io.cpp:5187-5189)
             ("f4_off0_tight", "f4", 16, False, False), ("i2_off348_tight", "i2", 4, False, False), ("i2_off100", "i2", 4, False, False)]
    lax = {"f4_off0_tight": dict(vox_offset=0.0, hdr_bytes=348), "i2_off348_tight": dict(vox_offset=348.0, hdr_bytes=348),
           "i2_off100": dict(vox_offset=100.0)}
    # r05: NIfTI-2 single files, and two-file images (.hdr + .img: NIfTI-1 "ni1", NIfTI-2 "ni2", ANALYZE 7.5 without a magic), named by
    # either file.  (name, dtype, code, big endian, gz, kind)
    more = [("f4_nii2", "f4", 16, False, False, "n+2"), ("i2_nii2_be", "i2", 4, True, False, "n+2"), ("f8_nii2_gz", "f8", 64, False, True, "n+2"),
            ("i2_pair", "i2", 4, False, False, "ni1"), ("f4_pair_be_gz", "f4", 16, True, True, "ni1"), ("u2_pair_nii2", "u2", 512, False, False, "ni2"),
            ("u1_analyze", "u1", 2, False, False, "analyze"), ("f4_pair_by_img", "f4", 16, False, False, "ni1")]
    g9 = {"names": np.array([c[0] for c in cases] + [c[0] for c in more])}
    with tempfile.TemporaryDirectory() as t:
        for name, dt, code, be, gz, kind in more:
            info = np.iinfo(dt) if np.dtype(dt).kind in "iu" else None
            vol = np.clip(base, info.min, info.max).astype(dt) if info else base.astype(dt)
            z = (lambda b: gzip.compress(b, mtime=0)) if gz else (lambda b: b)
            sfx = ".gz" if gz else ""
            if kind == "n+2":
                files = {name + ".nii" + sfx: z(nifti2_header(vol, dt, code, be) + payload(vol, dt, be))}
                arg = name + ".nii" + sfx
            else:
                if kind == "ni2":
                    hdr = nifti2_header(vol, dt, code, be, magic=b"ni2\0\r\n\032\n", vox_offset=0)[:540]
                else:
                    hb = bytearray(nifti1_bytes(vol, dt, code, big_endian=be, vox_offset=0.0, hdr_bytes=348))
                    hb[344:348] = b"ni1\0" if kind == "ni1" else b"\0\0\0\0"
                    hdr = bytes(hb)
                files = {name + ".hdr" + sfx: z(hdr), name + ".img" + sfx: z(payload(vol, dt, be))}
                arg = name + (".img" if name.endswith("by_img") else ".hdr") + sfx
            for fn_, blob in files.items():
                open(os.path.join(t, fn_), "wb").write(blob)
            nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
            ptr = fn(os.path.join(t, arg).encode(), C.byref(nx), C.byref(ny), C.byref(nz))
            assert ptr and (nx.value, ny.value, nz.value) == (7, 6, 5), name
            g9[name + "_files"] = np.array(sorted(files))
            for fn_, blob in files.items():
                g9[name + "_blob_" + fn_] = np.frombuffer(blob, np.uint8)
            g9[name + "_arg"] = np.array(arg)
            g9[name + "_data"] = np.ctypeslib.as_array(ptr, shape=(5, 6, 7)).copy()
            g9[name + "_plain"] = vol.astype(np.float32)
    with tempfile.TemporaryDirectory() as t:
        for name, dt, code, be, gz in cases:
            info = np.iinfo(dt) if np.dtype(dt).kind in "iu" else None
            vol = np.clip(base, info.min, info.max).astype(dt) if info else base.astype(dt)
            blob = nifti1_bytes(vol, dt, code, big_endian=be, **lax.get(name, {}))
            if gz:
                blob = gzip.compress(blob, mtime=0)
            p = os.path.join(t, name + (".nii.gz" if gz else ".nii"))
            open(p, "wb").write(blob)
            nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
            ptr = fn(p.encode(), C.byref(nx), C.byref(ny), C.byref(nz))
            assert ptr and (nx.value, ny.value, nz.value) == (7, 6, 5), name
            g9[name + "_file"] = np.frombuffer(blob, np.uint8)
            g9[name + "_gz"] = np.bool_(gz)
            g9[name + "_data"] = np.ctypeslib.as_array(ptr, shape=(5, 6, 7)).copy()
            g9[name + "_plain"] = vol.astype(np.float32)  # the stored values cast to fp32 (no scl_slope / scl_inter)
    np.savez_compressed(os.path.join(HERE, "g9_nifti.npz"), **g9)
    for name in g9["names"]:
        print(name, "reference == plain cast:", bool(np.array_equal(g9[name + "_data"], g9[name + "_plain"])))


def make_g10():
    """G10: a key-point list written by the REFERENCE's write_sift_kp and read back by its read_sift_kp
    (Src/cUtil.cc:938-954, 1002-1016; through oracle/_ref): pins write_sift_kp / read_sift_kp of 3dsift_amd/host/src/io.cpp."""
    import ctypes as C
    import tempfile
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref3dsift.so"))
    lib.ref_write_sift_kp.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_char_p]
    lib.ref_read_sift_kp.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int]
    rng = np.random.Generator(np.random.PCG64(10))
    xyz = np.concatenate([
        rng.uniform(0, 512, (40, 3)),                                   # what toCvec emits: rx, ry, rz of matched key points
        np.array([[0, 0, 0], [-1.5, 2.25, -3.0], [10.123456, 0.000004, -0.000005], [1e6, 1e-6, 123456.789],
                  [0.000005, 0.999995, 2.5000049], [511.0, 1.0, 65536.5], [3.4e38, -3.4e38, 1.17549435e-38]]),
    ]).astype(np.float32)
    with tempfile.TemporaryDirectory() as t:
        p = os.path.join(t, "kp.csv")
        lib.ref_write_sift_kp(xyz.ctypes.data_as(C.POINTER(C.c_float)), len(xyz), p.encode())
        blob = open(p, "rb").read()
        back = np.zeros((len(xyz) + 8, 3), np.float32)
        n = lib.ref_read_sift_kp(p.encode(), back.ctypes.data_as(C.POINTER(C.c_float)), len(back))
    assert n == len(xyz), n
    np.savez_compressed(os.path.join(HERE, "g10_sift_kp.npz"), xyz=xyz, csv=np.frombuffer(blob, np.uint8), read_back=back[:n])
    print("g10:", len(xyz), "points,", len(blob), "bytes; first lines:", blob.decode().splitlines()[:2], "last:", blob.decode().splitlines()[-1])


def main():
    only = set(sys.argv[1:])
    ref = ol.load("ref")
    ref.set_threads(8)
    if not only or "g1" in only:
        make_g1(ref)
    if not only or "g9" in only:
        make_g9()
    if not only or "g10" in only:
        make_g10()
    if only and not (only - {"g1", "g9", "g10"}):
        return
    out = {}

    # ---- G2 ---------------------------------------------------------------------------------
    v1 = ramp_noise((10, 14, 20), 1)
    v2 = ramp_noise((7, 12, 9), 2)
    g2 = dict(v1=v1, v2=v2)
    for name, v, sig in (("v1_s0", v1, 0.538701), ("v1_s5", v1, 2.452547), ("v2_s2", v2, 1.226273)):
        g2[name] = ref.gaussian_smooth(v, np.float32(sig))
        g2[name + "_sigma"] = np.float32(sig)
    np.savez_compressed(os.path.join(HERE, "g2_smooth.npz"), **g2)

    # ---- G3 ---------------------------------------------------------------------------------
    shape3 = (28, 20, 24)  # nz, ny, nx  -> nx=24, ny=20, nz=28
    vol3 = synth.blobs(shape3, seed=11, noise=0.01, nblobs=12)
    ex = ref.extractor(vol3).run(3)
    g3 = dict(vol=vol3, noct=np.int32(ex.num_octaves), input=ex.input())
    for o in range(ex.num_octaves):
        for i in range(6):
            g3[f"gss_{o}_{i}"] = ex.gss(o, i)
        for i in range(5):
            g3[f"dog_{o}_{i}"] = ex.dog(o, i)
    e = ex.extrema()
    g3["extrema"] = np.stack([e["octave"], e["level"], e["x"].astype(np.int32), e["y"].astype(np.int32), e["z"].astype(np.int32)], 1)
    np.savez_compressed(os.path.join(HERE, "g3_pyramid.npz"), **g3)

    # ---- G4 / G5 / G6 -----------------------------------------------------------------------
    cases = {
        "a": dict(shape=(56, 48, 40), seed=7, noise=0.01),   # nx=40, ny=48, nz=56
        "b": dict(shape=(64, 64, 64), seed=1234, noise=0.0),
    }
    g4, g6 = {}, {}
    for tag, c in cases.items():
        vol = synth.blobs(c["shape"], seed=c["seed"], noise=c["noise"])
        g4[f"{tag}_shape"] = np.array(c["shape"], np.int32)
        g4[f"{tag}_seed"] = np.int32(c["seed"])
        g4[f"{tag}_noise"] = np.float64(c["noise"])
        g4[f"{tag}_vol_sha"] = sha(vol)
        ex = ref.extractor(vol).run(5)
        noct = ex.num_octaves
        g4[f"{tag}_noct"] = np.int32(noct)
        g4[f"{tag}_input_sha"] = sha(ex.input())
        hashes, absmax, scales = [], [], []
        for o in range(noct):
            for i in range(6):
                lv = ex.gss(o, i)
                shell = min(lv.shape) <= 9 and i == 5
                hashes.append(f"gss_{o}_{i}:{int(shell)}:" + interior_mask_hash(lv, shell))
                scales.append(ex.level_info(0, o * 6 + i)[2])
            for i in range(5):
                lv = ex.dog(o, i)
                shell = min(lv.shape) <= 9 and i == 4
                hashes.append(f"dog_{o}_{i}:{int(shell)}:" + interior_mask_hash(lv, shell))
                absmax.append(np.abs(lv).max())
        g4[f"{tag}_hashes"] = np.array(hashes)
        g4[f"{tag}_dog_absmax"] = np.array(absmax, np.float32)
        g4[f"{tag}_gss_scales"] = np.array(scales, np.float32)
        e = ex.extrema()
        g4[f"{tag}_extrema"] = np.stack([e["octave"], e["level"], e["x"].astype(np.int32), e["y"].astype(np.int32), e["z"].astype(np.int32)], 1)
        g4[f"{tag}_extrema_scale"] = e["scale"].copy()
        kp, desc = ex.keypoints()
        g6[f"{tag}_kp"] = kp
        g6[f"{tag}_desc"] = desc

        if tag == "a":
            # G5: orientation of every extremum, through the reference's free function
            codes, outs = [], np.zeros(len(e), ol.KP_DTYPE)
            for j, k in enumerate(e):
                lvl = ex.gss(int(k["octave"]), int(k["level"]))
                unit = ex.level_info(0, int(k["octave"]) * 6 + int(k["level"]))[1][0]
                code, ko = ref.orient_one(k, lvl, unit, np.float32(1.5) * k["scale"])
                codes.append(code)
                outs[j] = ko
            np.savez_compressed(os.path.join(HERE, "g5_orient.npz"), codes=np.array(codes, np.int32),
                                win=outs["win"], eigvalue=outs["eigvalue"], str_tensor=outs["str_tensor"],
                                Rotation=outs["Rotation"], eigvector=outs["eigvector"])
    np.savez_compressed(os.path.join(HERE, "g4_hashes.npz"), **g4)
    np.savez_compressed(os.path.join(HERE, "g6_keypoints.npz"), **g6)

    # ---- G7 ---------------------------------------------------------------------------------
    verts, idx = ref.mesh()
    rng = np.random.Generator(np.random.PCG64(5))
    dirs = rng.normal(size=(1000, 3)).astype(np.float32)
    dirs[:50] *= 1e-4   # some below the |g|^2 < eps rejection
    special = [verts.reshape(-1, 3)]                                   # exactly on vertices
    special.append(((verts[:, 0] + verts[:, 1]) * 0.5))                # edge midpoints
    special.append(verts.mean(1))                                      # face centres
    dirs = np.concatenate([dirs] + [s.astype(np.float32) for s in special], 0)
    faces = np.zeros(len(dirs), np.int32)
    bary = np.zeros((len(dirs), 3), np.float32)
    for j, d in enumerate(dirs):
        faces[j], b = ref.intersect(d)
        bary[j] = b if faces[j] >= 0 else 0
    np.savez_compressed(os.path.join(HERE, "g7_mesh.npz"), verts=verts, idx=idx, dirs=dirs, faces=faces, bary=bary)

    # ---- G8 ---------------------------------------------------------------------------------
    va = synth.blobs((64, 64, 64), seed=1234)
    vb = synth.blobs((64, 64, 64), seed=1234, shift=(1.0, 0.0, 0.0))
    ka, da = ref.extractor(va).run(5).keypoints()
    kb, db = ref.extractor(vb).run(5).keypoints()
    xa = np.stack([ka["rx"], ka["ry"], ka["rz"]], 1)
    xb = np.stack([kb["rx"], kb["ry"], kb["rz"]], 1)
    g8 = dict(da=da, db=db, xa=xa, xb=xb)
    sets = {"p": (da, xa, db, xb)}
    # permuted target so that some ref keypoint's best match is target index 0 (the
    # "index 0 can never be rejected" quirk, cMatcher.cc:93,142)
    r0 = ref.match(da, xa, db, xb, 0.85, 1)
    hit = int(r0["gIdx"][np.nonzero(r0["gIdx"] > 0)[0][0]])
    perm = np.arange(len(db)); perm[0], perm[hit] = perm[hit], perm[0]
    sets["q"] = (da, xa, db[perm], xb[perm])
    g8["perm"] = perm.astype(np.int32)
    for tag, (a, ax, b, bx) in sets.items():
        for mode in (1, 2, 3):
            for thr in (0.85, 0.95):
                r = ref.match(a, ax, b, bx, thr, mode)
                key = f"{tag}_m{mode}_t{int(thr * 100)}"
                for k, v in r.items():
                    g8[f"{key}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "g8_match.npz"), **g8)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()

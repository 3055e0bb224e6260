"""Z-slab sharding: the sharded pipeline, run with all ranks simulated on ONE GPU (3dsift_amd/slab.py SimComm:
neighbour sends become device copies, everything else is the code the RCCL path runs), must reproduce the single-volume
result BIT FOR BIT -- same arithmetic, only data placement changes."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

capi = importlib.import_module("3dsift_amd.capi")
slab = importlib.import_module("3dsift_amd.slab")
synth = importlib.import_module("3dsift_amd.synth")


def _volume(shape, seed):
    return synth.blobs(shape, seed=seed, noise=0.01)


def _single(vol):
    s = capi.CreateCSIFT3D(vol)
    s.KpSiftAlgorithm()
    return s


@pytest.mark.parametrize("shape,world,sharded,partial", [((128, 128, 128), 2, 1, True), ((128, 128, 128), 4, 2, True), ((160, 80, 96), 3, 2, True),
                                                         ((128, 128, 128), 2, 2, True), ((64, 64, 64), 8, 2, True), ((64, 64, 64), 8, 1, True),
                                                         ((128, 128, 128), 4, 2, False), ((160, 80, 96), 3, 2, False), ((64, 64, 64), 8, 1, False)])
def test_slab_equals_single_volume(shape, world, sharded, partial):
    """partial (r05, the default of the driver): descriptor windows split along z over the ranks -- records out, partial integer
    histograms back, 13-plane halos; not partial: whole windows on 38-plane halos (r02).  Both bit-identical to the single volume."""
    vol = _volume(shape, seed=11 + world)
    nz, ny, nx = shape
    ref = _single(vol)
    kp_ref, ds_ref = ref.GetKeypoints()
    assert len(kp_ref) > 10

    ex = slab.SlabExtractor((nx, ny, nz), slab.SimComm(world), sharded_octaves=sharded, desc_partial=partial)
    assert ex.S == sharded
    assert ex.halo == (13 if partial else 38)
    ex.load(volume=vol)
    ex.KpSiftAlgorithm()
    if partial:
        assert sum(ex.window_bytes()) > 0

    # pyramids of the sharded octaves: owned planes of every level, bit for bit
    ng, nd = ref.levels + 3, ref.levels + 2
    ext_ref = ref.extrema()
    key = lambda a: sorted(zip(a["level"].tolist(), a["z"].tolist(), a["y"].tolist(), a["x"].tolist()))
    for o in range(ex.S):
        for i in range(ng):
            want = ref.gss(o, i)
            for w in ex._wl():
                st = w.stages[o]
                got = st.ctx.held_level(0, i)[w.halo:w.halo + (st.z1 - st.z0)]
                assert np.array_equal(got, want[st.z0:st.z1]), f"octave {o} GSS level {i} rank {w.rank}"
        for i in range(nd):
            want = ref.dog(o, i)
            for w in ex._wl():
                st = w.stages[o]
                got = st.ctx.held_level(1, i)[w.halo:w.halo + (st.z1 - st.z0)]
                assert np.array_equal(got, want[st.z0:st.z1]), f"octave {o} DoG level {i} rank {w.rank}"
        # extrema: union over the slabs == single-volume list of that octave
        eo = ext_ref[ext_ref["octave"] == o]
        got = np.concatenate([w.stages[o].ctx.extrema() for w in ex._wl()])
        assert key(got) == key(eo)

    # keypoints + descriptors, in reference order
    kp, ds = ex.GetKeypoints()
    assert len(kp) == len(kp_ref)
    for f in kp_ref.dtype.names:
        assert np.array_equal(kp[f], kp_ref[f]), f
    assert np.array_equal(ds, ds_ref)
    ex.close()
    ref.close()


def test_random_slab_plans_equal_the_single_volume():
    """Eight random (shape, ranks, sharded octaves) draws -- odd depths, uneven slabs, slabs thinner than the halo, planes of one tile
    and of shifted tiles: keypoints and descriptors of the sharded run equal the single-volume extractor's, bit for bit."""
    rng = np.random.default_rng(404)
    done = 0
    for case in range(40):
        if done == 8:
            break
        nz = int(rng.integers(48, 150)); ny = int(rng.choice([48, 64, 70, 96, 100])); nx = int(rng.choice([48, 64, 72, 96, 130]))
        world = int(rng.integers(2, 7)); sharded = int(rng.integers(1, 3))
        try:
            ex = slab.SlabExtractor((nx, ny, nz), slab.SimComm(world), sharded_octaves=sharded)
        except ValueError:
            continue   # (too few planes for that many slabs)
        vol = _volume((nz, ny, nx), seed=500 + case)
        ref = _single(vol)
        kp_ref, ds_ref = ref.GetKeypoints()
        with capi.hook("march_tiles", done & 1):   # every second plan with the 64 x 32 tiles wherever a slab's levels fit them (global-z feed order, halo planes)
            ex.load(volume=vol)
            ex.KpSiftAlgorithm()
            kp, ds = ex.GetKeypoints()
        tag = ((nz, ny, nx), world, sharded, ex.S)
        assert len(kp) == len(kp_ref), tag
        for f in kp_ref.dtype.names:
            assert np.array_equal(kp[f], kp_ref[f]), (tag, f)
        assert np.array_equal(ds, ds_ref), tag
        ex.close(); ref.close()
        done += 1
    assert done == 8


def test_partial_windows_second_round_with_the_exact_unit():
    """The hook desc_mass_shift makes the first fixed-point unit of EVERY keypoint too fine (the estimate of the gradient mass is divided by
    2^s), so every record is flagged by its owner and repeated by all parts with the exact unit (the second exchange of the partial
    descriptor windows); the single-volume run takes its own second pass under the same hook.  Bit-identical descriptors."""
    shape, world = (128, 128, 128), 4
    vol = _volume(shape, seed=23)
    nz, ny, nx = shape
    with capi.hook("desc_mass_shift", 9):
        ref = _single(vol)
        kp_ref, ds_ref = ref.GetKeypoints()
        assert ref.debug_counters()["desc_second_passes"] > len(kp_ref) // 4
        ex = slab.SlabExtractor((nx, ny, nz), slab.SimComm(world), sharded_octaves=2)
        ex.load(volume=vol)
        ex.KpSiftAlgorithm()
        redone = sum(st.ctx.debug_counters()["desc_second_passes"] for w in ex._wl() for st in w.stages)
        kp, ds = ex.GetKeypoints()
    assert redone > 10
    assert len(kp) == len(kp_ref) > 50
    for f in kp_ref.dtype.names:
        assert np.array_equal(kp[f], kp_ref[f]), f
    assert np.array_equal(ds, ds_ref)
    ex.close()
    ref.close()


def test_config3_1024x1024x512_eight_slabs_equal_the_single_volume():
    """BASELINE configs[3] at full size: 1024 x 1024 x 512 over 8 (simulated) ranks, two sharded octaves (64- and 32-plane slabs, both
    thinner than the 38-plane halo) + replicated tail, against the single-volume extractor: same keypoints, same descriptors, bit
    for bit; the DoG maxima (reduced on the device) and the owned planes of the seed level of octave 1 as a pyramid probe."""
    import torch
    nx, ny, nz = 1024, 1024, 512
    vol = synth.blobs_torch((nz, ny, nx), "cuda", seed=4321)
    torch.cuda.synchronize()
    ref = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(nz, ny, nx))
    ref.KpSiftAlgorithm()
    kp_ref, ds_ref = ref.GetKeypoints()
    g10 = ref.gss(1, 0)
    ref.close()
    assert len(kp_ref) > 20000
    ex = slab.SlabExtractor((nx, ny, nz), slab.SimComm(8), sharded_octaves=2)
    assert ex.S == 2
    ex.load(device_slabs={r: vol[z0:z1] for r, (z0, z1) in enumerate(ex.bounds)})
    del vol
    ex.KpSiftAlgorithm()
    for w in ex._wl():
        st = w.stages[1]
        got = st.ctx.held_level(0, 0)[w.halo:w.halo + (st.z1 - st.z0)]
        assert np.array_equal(got, g10[st.z0:st.z1]), w.rank
    kp, ds = ex.GetKeypoints()
    ex.close()
    assert len(kp) == len(kp_ref)
    for f in kp_ref.dtype.names:
        assert np.array_equal(kp[f], kp_ref[f]), f
    assert np.array_equal(ds, ds_ref)


def test_slab_halo_too_small_is_refused():
    import torch
    nx = ny = nz = 64
    n = capi.SlabCSIFT3D.arena_floats(nx, ny, nz, 0, 32, 40, 4)
    arena = torch.zeros(n, dtype=torch.float32, device="cuda")
    with pytest.raises(capi.Sift3dError):
        capi.SlabCSIFT3D(nx, ny, nz, 0, 32, 4, 4, arena.data_ptr(), n)   # halo 4 < widest Gaussian (hw 8)
    with pytest.raises(capi.Sift3dError):
        capi.SlabCSIFT3D(nx, ny, nz, 40, 33, 40, 4, arena.data_ptr(), n)  # empty range
    capi.SlabCSIFT3D(nx, ny, nz, 1, 33, 40, 4, arena.data_ptr(), n).close()  # (an odd start is admissible since r06: the native driver's weighted slabs)


def test_seeded_tail_equals_octaves_of_single_volume():
    """a seeded context fed G[1][0] of a single-volume run reproduces octaves >= 1 exactly"""
    vol = _volume((96, 96, 96), seed=5)
    ref = _single(vol)
    kp_ref, ds_ref = ref.GetKeypoints()
    g10 = ref.gss(1, 0)
    t = capi.SeededCSIFT3D(g10.shape, 1, ref.num_octaves)
    t.seed_host(g10)
    t.KpSiftAlgorithm()
    kp, ds = t.GetKeypoints()
    m = kp_ref["octave"] >= 1
    assert m.sum() > 0 and len(kp) == m.sum()
    for f in kp_ref.dtype.names:
        assert np.array_equal(kp[f], kp_ref[f][m]), f
    assert np.array_equal(ds, ds_ref[m])
    # descriptor partition: rows not owned stay zero, owned rows are identical
    t.set_partition(1, 3)
    t.KpSiftAlgorithm()
    kp2, ds2 = t.GetKeypoints()
    own = np.zeros(len(kp2), bool)
    own[slab.described_rows(kp2["level"], 1, 3)] = True
    assert own.sum() in (len(kp2) // 3, len(kp2) // 3 + 1)
    assert np.array_equal(ds2[own], ds[own]) and not ds2[~own].any()
    # partitioned orientation: three handles orient a third of the extrema each; the integer sum of the packed rows
    # restores the full orientation result on every one of them
    import torch
    hs = []
    for r in range(3):
        h = capi.SeededCSIFT3D(g10.shape, 1, ref.num_octaves)
        h.seed_host(g10); h.set_partition(r, 3); h.run_partial_orientation()
        hs.append(h)
    n_ext = hs[0].num_extrema()
    assert n_ext == len(t.extrema())
    bufs = [torch.empty(n_ext * capi.ORIENT_WORDS, dtype=torch.int32, device="cuda") for _ in hs]
    for h, b in zip(hs, bufs):
        h.export_orientation(b.data_ptr())
    torch.cuda.synchronize()
    rows = [b.view(n_ext, -1).cpu().numpy() for b in bufs]
    for r in range(3):
        assert not rows[r][np.arange(n_ext) % 3 != r].any()
    total = bufs[0] + bufs[1] + bufs[2]
    torch.cuda.synchronize()
    full = np.zeros((len(kp), 768), np.float32)
    for r, h in enumerate(hs):
        h.import_orientation(total.data_ptr()); h.run_describe()
        kp3, ds3 = h.GetKeypoints()
        for f in kp.dtype.names:
            assert np.array_equal(kp3[f], kp[f]), f
        assert np.array_equal(h.orientation_codes(), t.orientation_codes())
        mine = slab.described_rows(kp3["level"], r, 3)
        full[mine] = ds3[mine]
        assert not ds3[slab.described_rows(kp3["level"], (r + 1) % 3, 3)].any()
        h.close()
    assert np.array_equal(full, ds)
    t.close(); ref.close()


def test_python_driver_over_rccl_world_of_one():
    """The python driver over `DistComm` with the nccl (= RCCL) backend, one rank on the one GPU of the box: process-group and
    communicator set-up (the tail's own group included), the count all-gather, the MAX / SUM all-reduces on device tensors, the empty
    point-to-point groups and the whole partial-window path with the rank's own part -- everything of the N > 1 run that does not need a
    second GPU.  Keypoints and descriptors equal the single volume bit for bit."""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, os, sys, numpy as np, torch
        sys.path.insert(0, %r)
        import torch.distributed as dist
        capi = importlib.import_module("3dsift_amd.capi"); slab = importlib.import_module("3dsift_amd.slab"); synth = importlib.import_module("3dsift_amd.synth")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        vol = synth.blobs((96, 112, 128), seed=31, noise=0.01)
        nz, ny, nx = vol.shape
        ref = capi.CreateCSIFT3D(vol); ref.KpSiftAlgorithm(); kp_ref, ds_ref = ref.GetKeypoints()
        ex = slab.SlabExtractor((nx, ny, nz), slab.DistComm(), device=0, sharded_octaves=2)
        assert ex.desc_partial and ex.halo == 13
        ex.load(volume=vol)
        ex.KpSiftAlgorithm(); ex.KpSiftAlgorithm()
        kp, ds = ex.GetKeypoints()
        assert len(kp) == len(kp_ref) > 50, (len(kp), len(kp_ref))
        for f in kp_ref.dtype.names:
            assert np.array_equal(kp[f], kp_ref[f]), f
        assert np.array_equal(ds, ds_ref)
        ex.close(); ref.close()
        dist.destroy_process_group()
        print("rccl world of one ok", len(kp))
    """) % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world of one ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_whole_window_describe_is_refused_on_a_partial_window_halo():
    """sift3d_slab_describe marches whole descriptor windows (38 planes of halo); on the 13-plane halo of the partial-window mode it
    would read planes nobody exchanged: refused loudly (ERR_STATE), the partial-window calls are the way."""
    vol = _volume((64, 64, 64), seed=3)
    ex = slab.SlabExtractor((64, 64, 64), slab.SimComm(2), sharded_octaves=1)   # desc_partial: the driver's default
    ex.load(volume=vol)
    ex.KpSiftAlgorithm()
    st = ex._wl()[0].stages[0]
    st.ctx.detect()
    with pytest.raises(capi.Sift3dError, match="whole descriptor windows"):
        st.ctx.describe()
    ex.close()

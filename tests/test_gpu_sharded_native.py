"""-m gpu: the NATIVE (C++) driver of the z-slab sharding (3dsift_amd/csrc/sharded.hip; SURVEY 8e, BASELINE.json configs[3]) through
the C-ABI sift3d_sharded_* and through the C++ shell (CSIFT3DFactory::CreateCSIFT3D with SIFT3D_SIM_RANKS / SIFT3D_DEVICES).

A 1-GPU box cannot run several RCCL ranks (one rank per device), so the driver is checked in two ways:
  * simulated ranks (device copies instead of ncclSend / ncclRecv, same plan, same slab contexts, same merge): 2 .. 8 ranks, one and
    two sharded octaves, the default plan (descriptor windows split along z, the tail once on the last rank), forced partial and forced
    whole windows -- keypoints AND descriptors bit-identical to the single-volume extractor;
  * the RCCL transport itself with a world of ONE rank on the one GPU: librccl is opened, three communicators are created and the MAX
    all-reduce of the DoG maxima really runs through RCCL on the device (the point-to-point halo sends and the gather of the tail's seed
    level need a second GPU)."""
import importlib
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

capi = importlib.import_module("3dsift_amd.capi")
synth = importlib.import_module("3dsift_amd.synth")
PKG = os.path.dirname(capi.__file__)


@pytest.fixture(scope="module")
def vol_and_single():
    vol = synth.blobs((160, 96, 128), seed=77, noise=0.01)   # nz 160: 8 ranks x 20 planes (5 after two halvings); ny 96, nx 128
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    assert len(kp) > 100 and ex.num_octaves >= 3
    return vol, kp, ds


@pytest.mark.parametrize("ranks,octs", [(2, 1), (2, 2), (3, 2), (4, 2), (8, 2), (5, 1)])
def test_simulated_ranks_equal_the_single_volume(vol_and_single, ranks, octs):
    """whole descriptor windows on the wide halos (SIFT3D_SHARDED_WHOLE_WINDOWS; r05's default, still what slabs too thin for the split get)"""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs, partial_windows=False)
    info = sh.info()
    assert info["world"] == ranks and 1 <= info["sharded_octaves"] <= octs and not info["partial_windows"] and info["halo"] == capi.slab_min_halo()
    for _ in range(2):   # a second run on the same contexts gives the same result
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp), (ranks, octs)
        assert np.array_equal(d2, ds), (ranks, octs)
    sh.close()


@pytest.mark.parametrize("ranks,octs", [(2, 1), (2, 2), (3, 2), (4, 2), (5, 1)])
def test_simulated_ranks_with_partial_descriptor_windows_equal_the_single_volume(vol_and_single, ranks, octs):
    """sift3d_sharded_create_ex(SIFT3D_SHARDED_PARTIAL_WINDOWS): records to the z-neighbours, partial integer histograms back, 13-plane
    level halos instead of 39 -- the same keypoints and descriptors, bit for bit, run after run on the same scratch."""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs, partial_windows=True)
    info = sh.info()
    assert info["world"] == ranks and info["halo"] == capi.slab_min_halo_partial() and info["partial_windows"]
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp), (ranks, octs)
        assert np.array_equal(d2, ds), (ranks, octs)
    sh.close()


@pytest.mark.parametrize("ranks,octs", [(2, 2), (4, 2), (8, 2), (6, 1)])
def test_default_plan_splits_the_windows_and_runs_the_tail_once(vol_and_single, ranks, octs):
    """r06 defaults of sift3d_sharded_create: descriptor windows split along z wherever a window spans at most six ranks (8 ranks x 2 octaves of
    the 160-plane volume: whole windows instead), the octaves behind the sharded ones run ONCE, on the last rank, which owns fewer planes."""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs)
    info = sh.info()
    assert info["partial_windows"] == (not (ranks == 8 and octs == 2)), info
    assert info["tail_rank"] == ranks - 1 and sum(info["planes"]) == vol.shape[0] and info["planes"][-1] <= min(info["planes"][:-1]), info
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp), info
        assert np.array_equal(d2, ds), info
    assert sh.info()["seconds_incl_merge"] > sh.info()["seconds"] > 0
    if True:
        # the solo re-run of every rank on the buffers the run left behind (what bench.py times per rank) leaves the results as they are
        ts = [sh.time_rank(r) for r in range(ranks)]
        assert all(0 < t < 1.0 for t in ts), ts
        k3, d3 = sh.GetKeypoints()
        assert np.array_equal(k3, kp) and np.array_equal(d3, ds)
        k4, d4 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k4, kp) and np.array_equal(d4, ds)
    sh.close()


def test_three_sharded_octaves_mixed_window_forms():
    """256^3 over 8 simulated ranks with THREE sharded octaves: the 32-plane slabs of octave 0 split their descriptor windows along z, the 16-
    and 8-plane slabs of octaves 1 and 2 (a window would span 7 and more ranks) carry whole windows on 38-plane halos -- one plan, per-octave
    forms, the tail (octaves >= 3) once on the last rank; bit-identical to the single volume, also after every rank's solo re-run"""
    import torch
    vol = synth.blobs_torch((256, 256, 256), "cuda", seed=31).cpu().numpy()
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    ex.close()
    sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=8, sharded_octaves=3)
    info = sh.info()
    assert info["sharded_octaves"] == 3 and info["stage_partial"] == [True, False, False] and not info["partial_windows"], info
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    assert all(0 < sh.time_rank(r) < 1.0 for r in range(8))
    k3, d3 = sh.KpSiftAlgorithm().GetKeypoints()
    sh.close()
    assert np.array_equal(k3, kp) and np.array_equal(d3, ds)


def test_native_list_regrow_without_readbacks(vol_and_single):
    """the slab contexts' lists start tiny (hook list_cap): the overflow is found when the counts are read (sift3d_slab_keypoints_count), the lists
    are regrown and detection + orientation repeated -- same results"""
    vol, kp, ds = vol_and_single
    with capi.hook("list_cap", 48):
        sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=3, sharded_octaves=2)
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        sh.close()
    assert np.array_equal(k2, kp) and np.array_equal(d2, ds)


def test_native_partial_windows_second_round_and_thin_slabs(vol_and_single):
    """(a) desc_mass_shift flags every record in the first round: the native driver compacts the flagged subset and repeats it with the exact
    unit over the same exchange; (b) 8 ranks x 2 octaves of the 160-plane volume would make a window span more than six ranks: refused."""
    vol, _, _ = vol_and_single
    with capi.hook("desc_mass_shift", 9):
        ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        kp, ds = ex.GetKeypoints()
        assert ex.debug_counters()["desc_second_passes"] > len(kp) // 4
        sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=3, sharded_octaves=2, partial_windows=True)
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        # a solo re-run of a rank behind a run that had a second round (r06: the second round has a scratch of its own, so the first round's
        # lists the solo rank reads from its neighbours are still in place -- they were overwritten, and at full size read out of bounds);
        # it times the first round only and leaves the flagged rows as the full run stored them
        assert all(0 < sh.time_rank(r) < 1.0 for r in range(3))
        k3, d3 = sh.GetKeypoints()
        k4, d4 = sh.KpSiftAlgorithm().GetKeypoints()
        sh.close(); ex.close()
    assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    assert np.array_equal(k3, kp) and np.array_equal(d3, ds) and np.array_equal(k4, kp) and np.array_equal(d4, ds)
    with pytest.raises(capi.Sift3dError, match="partial descriptor windows"):
        capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=8, sharded_octaves=2, partial_windows=True)


def test_random_native_plans_equal_the_single_volume():
    """Six random (shape, simulated ranks, sharded octaves, default / partial / whole descriptor windows) draws through the native driver: keypoints and descriptors
    bit-identical to the single-volume extractor (uneven slabs, odd depths, slabs thinner than the halo)."""
    rng = np.random.default_rng(808)
    done = 0
    for case in range(40):
        if done == 6:
            break
        nz = int(rng.integers(48, 150)); ny = int(rng.choice([48, 64, 70, 96])); nx = int(rng.choice([48, 64, 72, 96, 130]))
        ranks = int(rng.integers(2, 7)); octs = int(rng.integers(1, 3))
        vol = synth.blobs((nz, ny, nx), seed=900 + case, noise=0.01)
        try:   # every other case with the descriptor windows split along z
            sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=ranks, sharded_octaves=octs, partial_windows=(None, True, False)[case % 3])
        except capi.Sift3dError:
            continue   # (too few planes for that many slabs, or slabs too thin for partial windows)
        ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        kp, ds = ex.GetKeypoints()
        with capi.hook("march_tiles", done & 1):   # every second plan with the 64 x 32 tiles wherever a slab's levels fit them
            k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        tag = ((nz, ny, nx), ranks, octs, case % 3, sh.info())
        sh.close(); ex.close()
        assert np.array_equal(k2, kp), tag
        assert np.array_equal(d2, ds), tag
        done += 1
    assert done == 6


@pytest.mark.parametrize("partial", [False, True, None])
def test_rccl_transport_world_of_one(vol_and_single, partial):
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=0, sharded_octaves=2, partial_windows=partial)   # one real rank: every collective goes through librccl
    assert sh.info()["world"] == 1
    k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
    sh.close()
    assert np.array_equal(k2, kp) and np.array_equal(d2, ds)


def test_cpp_shell_shards_for_the_unchanged_user():
    """The reference's user code (CreateCSIFT3D / KpSiftAlgorithm / GetKeypoints, Example.cpp:21-44) unchanged: with SIFT3D_SIM_RANKS=4
    the shell shards the volume with the native driver; the keypoint list equals the single-GPU one bit for bit."""
    src = r"""
    #include "Include/cSIFT3D.h"
    #include <cstdio>
    using namespace CPUSIFT;
    int main(int, char** a) {
        CSIFT3D *A = CSIFT3DFactory::CreateCSIFT3D(std::string(a[1]));
        A->KpSiftAlgorithm();
        std::vector<Keypoint> k = A->GetKeypoints();
        FILE* f = fopen(a[2], "wb");
        for (auto& p : k) { fwrite(&p.x, 4, 3, f); fwrite(&p.octave, 4, 2, f); fwrite(p.Rotation, 4, 9, f); fwrite(p.desc, 4, 768, f); }
        fclose(f);
        printf("keypoints %zu total %.6f\n", k.size(), A->m_timer.d_TotalTime);
        delete A;
        return 0;
    }"""
    vol = synth.blobs((96, 80, 112), seed=5, noise=0.01)
    with tempfile.TemporaryDirectory() as t:
        with open(os.path.join(t, "v.bin"), "wb") as f:
            f.write(struct.pack("<3i", 112, 80, 96) + vol.tobytes())
        open(os.path.join(t, "m.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(PKG, "host"), "-o", os.path.join(t, "m"), os.path.join(t, "m.cpp"),
                               "-L" + PKG, "-lsift3d", "-lsift3d_hip", "-Wl,-rpath," + PKG])
        outs = []
        for env in ({}, {"SIFT3D_SIM_RANKS": "4"}, {"SIFT3D_DEVICES": "0"}, {"SIFT3D_SIM_RANKS": "3", "SIFT3D_PARTIAL_WINDOWS": "1"}, {"SIFT3D_SIM_RANKS": "3", "SIFT3D_PARTIAL_WINDOWS": "0"},
                    {"SIFT3D_DEVICES": "0,0,0", "SIFT3D_TRANSPORT": "copies"},    # (r06: three rank THREADS on the one GPU, copy transport)
                    {"SIFT3D_SIM_RANKS": "3", "SIFT3D_GHOST_OCTAVE0": "1"}):        # (r06: octave 0 on ghost zones)
            e = dict(os.environ, **env)
            o = subprocess.check_output([os.path.join(t, "m"), os.path.join(t, "v.bin"), os.path.join(t, "k.bin")], env=e, stderr=subprocess.STDOUT).decode()
            outs.append((o.strip().splitlines()[-1].split()[1], open(os.path.join(t, "k.bin"), "rb").read()))
    assert int(outs[0][0]) > 30
    assert all(o == outs[0] for o in outs[1:])


def test_bench_runs_the_native_legs_in_a_child_process():
    """bench.py at N > 1 starts the native driver's legs as CHILD processes of rank 0 (a hard fault of the transport's first multi-GPU run must
    not cost the headline line).  Here: the helper itself on the one GPU (an RCCL world of one), both window forms; a child that fails
    (a volume too small to shard) comes back as a reason, not as an exception."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    for partial in (None, False):
        res, err = bench.run_slab_native_child("256x192x128", 1, 2, 1, partial, 240)
        assert err is None, err
        assert res["keypoints"] > 100 and res["ms_per_step"] > 0
        assert res["descriptor_windows"] == ("whole windows on plane halos" if partial is False else "partial integer histograms")
    res, err = bench.run_slab_native_child("256x192x128", 1, 2, 1, None, 240, transport="copies")   # (r06: the leg over the copy transport)
    assert err is None and res["keypoints"] > 100 and "copy transport" in res["workload"], (res, err)
    res, err = bench.run_slab_native_child("8x8x4", 1, 1, 0, False, 120)
    assert res is None and err


# ---- r06: the COPY transport -- the multi-threaded driver (one host thread per rank, its own streams, rendezvous, the tail's thread) on ONE GPU ----
@pytest.mark.parametrize("ranks,octs,partial", [(2, 1, None), (2, 2, True), (3, 2, None), (4, 2, False), (5, 1, True), (8, 2, None), (8, 2, False)])
def test_rank_threads_on_one_gpu_equal_the_single_volume(vol_and_single, ranks, octs, partial):
    """SIFT3D_SHARDED_COPY_TRANSPORT with `devices` = the one GPU named `ranks` times: every rank has its host thread, its streams per sharded
    octave and its deferred streams, exactly as over RCCL -- only a neighbour's data is fetched by a device copy behind the sender's event instead
    of ncclSend / ncclRecv.  What the simulated ranks (one thread, shared streams) cannot show: the rendezvous of the counts, the ranks' threads
    enqueueing against each other, the tail's own thread.  Keypoints and descriptors bit-identical to the single volume, run after run."""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,) * ranks, sharded_octaves=octs, partial_windows=partial, transport="copies")
    info = sh.info()
    assert info["world"] == ranks and 1 <= info["sharded_octaves"] <= octs
    for _ in range(3):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp), (ranks, octs, partial)
        assert np.array_equal(d2, ds), (ranks, octs, partial)
    sh.close()


def test_rank_threads_three_sharded_octaves_and_the_second_round():
    """256 x 256 x 256 over 8 rank threads: three sharded octaves (partial / whole / whole windows by the driver's rule), the tail on the last
    rank); then 4 rank threads with the first fixed-point unit forced to fail (hook desc_mass_shift): the second round of the flagged records."""
    vol = synth.blobs((256, 256, 256), seed=31, noise=0.01)
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    ex.close()
    sh = capi.ShardedCSIFT3D(vol, devices=(0,) * 8, sharded_octaves=3, transport="copies")
    assert sh.info()["sharded_octaves"] == 3 and sh.info()["stage_partial"] == [True, False, False] and sh.info()["tail_rank"] == 7
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    sh.close()
    with capi.hook("desc_mass_shift", 9):
        ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
        kp9, ds9 = ex.GetKeypoints()
        assert ex.debug_counters()["desc_second_passes"] > 0
        ex.close()
        sh = capi.ShardedCSIFT3D(vol, devices=(0,) * 4, sharded_octaves=2, partial_windows=True, transport="copies")
        for _ in range(2):
            k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
            assert np.array_equal(k2, kp9) and np.array_equal(d2, ds9)
        sh.close()


def test_rccl_refuses_a_device_named_twice_and_copies_take_it():
    vol = synth.blobs((64, 64, 64), seed=5)
    with pytest.raises(capi.Sift3dError, match="one rank per device"):
        capi.ShardedCSIFT3D(vol, devices=(0, 0))
    sh = capi.ShardedCSIFT3D(vol, devices=(0, 0), transport="copies")
    assert sh.info()["world"] == 2
    sh.close()
    with pytest.raises(capi.Sift3dError, match="at most 16 ranks"):
        capi.ShardedCSIFT3D(vol, devices=(0,) * 17, transport="copies")


def test_sixteen_rank_threads(vol_and_single):
    """the copy transport's upper end: sixteen rank threads on the one GPU (ten planes each: whole windows, one sharded octave, halos from several ranks)"""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=(0,) * 16, sharded_octaves=1, transport="copies")
    assert sh.info()["world"] == 16 and not any(sh.info()["stage_partial"])
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    sh.close()


@pytest.mark.parametrize("transport,devices,victim", [("copies", (0, 0, 0, 0), 2), ("copies", (0, 0, 0), 0), ("rccl", (0,), 0)])
def test_a_failing_rank_releases_the_others_and_kills_the_handle(vol_and_single, transport, devices, victim):
    """The failure protocol of the rank threads (never taken in a healthy run): hook sharded_fail_rank makes ONE rank give up behind its pyramid
    while the others go on to the rendezvous of the counts.  The run must come back (no rank left waiting), name the rank that failed first,
    leave the handle dead -- a second run is refused -- and destroy must return (streams drained, the tail's run in flight dropped; over RCCL the
    communicators aborted, not destroyed twice).  A fresh handle afterwards gives the single volume's result."""
    import time
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, devices=devices, sharded_octaves=2, transport=transport)
    sh.KpSiftAlgorithm()
    with capi.hook("sharded_fail_rank", victim + 1):
        t0 = time.time()
        with pytest.raises(capi.Sift3dError, match=f"rank {victim}: injected failure"):
            sh.KpSiftAlgorithm()
        assert time.time() - t0 < 20
    with pytest.raises(capi.Sift3dError, match="dead"):
        sh.KpSiftAlgorithm()
    sh.close()
    sh = capi.ShardedCSIFT3D(vol, devices=devices, sharded_octaves=2, transport=transport)
    k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
    assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    sh.close()


def test_thin_volumes_get_a_plan_that_runs():
    """r06 (scripts/soak_random.py): a volume of 32 planes asked for three sharded octaves was ACCEPTED -- octave 1 has 16 planes, octave 2 has 8, fewer
    than the 2 hw + 2 a z-march column needs, and slabs have no separable fallback -- and failed in its first run ("no fused kernel for this level").
    The plan now stops at the last octave the slab contexts admit (sift3d_slab_admits), in both drivers; parameters no slab takes are refused at
    create."""
    slab = importlib.import_module("3dsift_amd.slab")
    vol = synth.blobs((32, 96, 128), seed=3, noise=0.01)
    ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithm()
    kp, ds = ex.GetKeypoints()
    ex.close()
    assert len(kp) > 10
    assert capi.slab_admits(128, 96, 32) and capi.slab_admits(64, 48, 16, False) and not capi.slab_admits(32, 24, 8, False)
    for kw in (dict(sim_ranks=2), dict(sim_ranks=4), dict(devices=(0, 0, 0), transport="copies")):
        sh = capi.ShardedCSIFT3D(vol, sharded_octaves=3, **kw)
        assert sh.info()["sharded_octaves"] == 2
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        sh.close()
        assert np.array_equal(k2, kp) and np.array_equal(d2, ds)
    pex = slab.SlabExtractor((128, 96, 32), slab.SimComm(2), sharded_octaves=3)
    assert pex.S == 2
    pex.load(volume=vol)
    pex.KpSiftAlgorithm()
    k3, d3 = pex.GetKeypoints()
    pex.close()
    for f in kp.dtype.names:
        assert np.array_equal(k3[f], kp[f]), f
    assert np.array_equal(d3, ds)
    with pytest.raises(capi.Sift3dError, match="do not fit the slab kernels"):
        capi.ShardedCSIFT3D(vol, sim_ranks=2, sigma_default=2.4)   # half widths beyond 8


@pytest.mark.parametrize("kw", [dict(sim_ranks=2, sharded_octaves=1), dict(sim_ranks=3, sharded_octaves=2, partial_windows=True), dict(sim_ranks=4, sharded_octaves=2, partial_windows=False),
                                dict(sim_ranks=8, sharded_octaves=2), dict(devices=(0,) * 4, sharded_octaves=2, transport="copies"), dict(devices=(0,) * 5, sharded_octaves=1, transport="copies", partial_windows=True)])
def test_octave0_on_ghost_zones_equals_the_single_volume(vol_and_single, kw):
    """SIFT3D_SHARDED_GHOST_OCTAVE0 (r06): every rank holds its planes + 35 per side of the input and produces octave 0's levels on ghost ranges that shrink
    level by level -- no plane of octave 0 is exchanged (sift3d_sharded_traffic: only the octaves below and the window records / histograms).  The same keypoints
    and descriptors bit for bit: a ghost plane holds exactly what its owner computes for it."""
    vol, kp, ds = vol_and_single
    sh = capi.ShardedCSIFT3D(vol, ghost_octave0=True, **kw)
    ref = capi.ShardedCSIFT3D(vol, **kw)
    for _ in range(2):
        k2, d2 = sh.KpSiftAlgorithm().GetKeypoints()
        assert np.array_equal(k2, kp) and np.array_equal(d2, ds), kw
    ref.KpSiftAlgorithm()
    h_g, w_g = sh.traffic(); h_r, w_r = ref.traffic()
    assert sum(h_g) < sum(h_r) and (sum(w_g) > 0) == (sum(w_r) > 0)   # (the slabs' weights differ with ghost zones: other boundaries, other window neighbours)
    if sh.info()["sharded_octaves"] == 1:
        assert sum(h_g) == 0
    sh.close(); ref.close()

"""-m gpu: BASELINE.json configs[4] on the HIP path -- 8 volumes, the descriptors of all of them gathered, all 56 ordered
(ref, tar) pairs through muBruteMatcher::enhancedMatch (Src/cMatcher.cc:146-228) = sift3d_match on DEVICE-RESIDENT inputs, dealt to
8 ranks by 3dsift_amd/dist.py exactly as `bench.py --allpairs` does on 8 GPUs (there the gather is an RCCL all-gather; here the
eight "ranks" run one after the other on the one GPU of the test box).  Every one of the 56 results is compared bit for bit with
the oracle's matcher on the same descriptors.  Volumes: 256^3, seeds 1234 .. 1237, each once as generated and once shifted by one
voxel in x (SURVEY 8d's "target" construction), so 8 of the ordered pairs are true correspondences and 48 are unrelated."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

capi = importlib.import_module("3dsift_amd.capi")
dist = importlib.import_module("3dsift_amd.dist")
synth = importlib.import_module("3dsift_amd.synth")

N = 256
WORLD = 8


def test_config4_all_56_ordered_pairs_vs_oracle(orc):
    import os
    import torch

    dev = torch.device("cuda", 0)
    descs, xyzs, host = [], [], []
    for k in range(WORLD):
        vol = synth.blobs_torch((N, N, N), dev, seed=1234 + k // 2, shift=(float(k % 2), 0.0, 0.0))
        torch.cuda.synchronize()
        ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(N, N, N)).KpSiftAlgorithm()
        kp, ds = ex.GetKeypoints()
        n = len(kp)
        assert n > 500
        # what a rank contributes to the all-gather: its device-resident descriptors and coordinates, exported into buffers the
        # communication layer owns (sift3d_export_device) -- no host hop
        d_t = torch.empty((n, 768), dtype=torch.float32, device=dev)
        x_t = torch.empty((n, 3), dtype=torch.float32, device=dev)
        ex.export_device(d_t.data_ptr(), x_t.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_t.cpu().numpy(), ds)
        descs.append(d_t); xyzs.append(x_t)
        host.append((ds, np.stack([kp["rx"], kp["ry"], kp["rz"]], 1)))
        ex.close()
        del vol
    mt = capi.muBruteMatcher()

    def match_fn(da, xa, db, xb):
        return mt.enhancedMatch(da.data_ptr(), xa.data_ptr(), db.data_ptr(), xb.data_ptr(), 0.85, on_device=True, n=da.shape[0], m=db.shape[0])

    got = {}
    for rank in range(WORLD):
        mine = dist.match_pairs(descs, xyzs, match_fn, rank, WORLD)
        assert len(mine) == 7
        got.update(mine)
    assert sorted(got) == sorted(dist.ordered_pairs(WORLD)) and len(got) == 56
    orc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    related = 0
    for (i, j), res in sorted(got.items()):
        want = orc.match(host[i][0], host[i][1], host[j][0], host[j][1], 0.85, 3)
        for key in want:
            assert np.array_equal(res[key], want[key]), (i, j, key)
        if i // 2 == j // 2:
            related += 1
            assert len(want["pairs"]) > 100, (i, j, len(want["pairs"]))   # the shifted copy: most keypoints correspond
    assert related == 8


def test_config4_at_full_size_eight_512_cubed_volumes(orc):
    """BASELINE configs[4] at its FULL size (the test above runs 8 x 256^3): eight 512^3 volumes extracted one after the other
    (only the device-resident descriptors are kept), all 56 ordered pairs through enhancedMatch on the device.  The oracle's matcher takes
    ~3 s per pair of 11 000-keypoint sets on the box's cores, so 16 of the 56 -- the 8 true correspondences and 8 unrelated pairs -- are
    compared with it bit for bit; every pair is checked for the properties the reference's filter guarantees (Src/cMatcher.cc:81-144,
    146-228) and against a second run through the host-input path of sift3d_match."""
    import os
    import torch

    NF = 512
    dev = torch.device("cuda", 0)
    descs, xyzs, host = [], [], []
    for k in range(WORLD):
        vol = synth.blobs_torch((NF, NF, NF), dev, seed=1234 + k // 2, shift=(float(k % 2), 0.0, 0.0))
        torch.cuda.synchronize()
        ex = capi.CSIFT3D(None, device_ptr=vol.data_ptr(), shape=(NF, NF, NF)).KpSiftAlgorithm()
        kp, ds = ex.GetKeypoints()
        n = len(kp)
        assert n > 5000
        d_t = torch.empty((n, 768), dtype=torch.float32, device=dev)
        x_t = torch.empty((n, 3), dtype=torch.float32, device=dev)
        ex.export_device(d_t.data_ptr(), x_t.data_ptr())
        torch.cuda.synchronize()
        descs.append(d_t); xyzs.append(x_t)
        host.append((ds, np.stack([kp["rx"], kp["ry"], kp["rz"]], 1)))
        ex.close()
        del vol
    mt = capi.muBruteMatcher()

    def match_fn(da, xa, db, xb):
        return mt.enhancedMatch(da.data_ptr(), xa.data_ptr(), db.data_ptr(), xb.data_ptr(), 0.85, on_device=True, n=da.shape[0], m=db.shape[0])

    got = {}
    for rank in range(WORLD):
        got.update(dist.match_pairs(descs, xyzs, match_fn, rank, WORLD))
    assert sorted(got) == sorted(dist.ordered_pairs(WORLD)) and len(got) == 56
    for (i, j), res in sorted(got.items()):
        n, m = len(host[i][0]), len(host[j][0])
        g, sdx, gd, sd = res["gIdx"], res["sIdx"], res["gDist"], res["sDist"]
        assert len(g) == n and ((g > -m) & (g < m)).all()                                 # a rejected match keeps its index, negated (Src/cMatcher.cc:93)
        ok = g >= 0
        assert (gd <= sd).all() and len(res["pairs"]) == int(ok.sum())                      # best <= second best; one pair per survivor (toCvec)
        assert np.array_equal(res["pairs"][:, :3], host[i][1][ok]) and np.array_equal(res["pairs"][:, 3:], host[j][1][g[ok]])
        if i // 2 == j // 2:
            assert ok.sum() > 0.5 * n, (i, j, int(ok.sum()))                              # the shifted copy: most keypoints correspond
    # the same 56 through the host-input path (H2D of both sets inside the call): the same answers
    for (i, j) in [(0, 1), (2, 5), (7, 3)]:
        again = mt.enhancedMatch(host[i][0], host[i][1], host[j][0], host[j][1], 0.85)
        for key in again:
            assert np.array_equal(again[key], got[(i, j)][key]), (i, j, key)
    orc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    checked = [(i, j) for (i, j) in sorted(got) if i // 2 == j // 2] + [(0, 2), (1, 4), (2, 7), (3, 0), (4, 6), (5, 1), (6, 3), (7, 5)]
    assert len(checked) == 16
    for (i, j) in checked:
        want = orc.match(host[i][0], host[i][1], host[j][0], host[j][1], 0.85, 3)
        for key in want:
            assert np.array_equal(got[(i, j)][key], want[key]), (i, j, key)


def test_match_handles_all_ordered_pairs_and_peer_copy_path(orc):
    """sift3d_match_handles (r04): the native building block of configs[4] -- the device-resident results of two extractors matched
    wherever they live.  Four live extractors, all twelve ordered pairs, every output equal to the oracle's matcher on the host copies;
    again with the hook peer_copy, which stages the target through the peer-to-peer scratch (the path taken when the target sits on
    another GPU: the copy itself is hipMemcpyPeer, here from the device to itself)."""
    exs, host = [], []
    for k in range(4):
        vol = synth.blobs((96, 96, 96), seed=77 + k // 2, shift=(float(k % 2), 0.0, 0.0))
        ex = capi.CreateCSIFT3D(vol).KpSiftAlgorithmAsync()   # (runs in flight are completed by the match)
        exs.append(ex)
    for ex in exs:
        kp, ds = ex.GetKeypoints()
        assert len(kp) > 30
        host.append((ds, np.stack([kp["rx"], kp["ry"], kp["rz"]], 1)))
    mt = capi.muBruteMatcher()
    for hook_on in (0, 1):
        with capi.hook("peer_copy", hook_on):
            for i, j in dist.ordered_pairs(4):
                for mode, code in (("enhanced", 3), ("inject", 1)):
                    got = mt.matchExtractors(exs[i], exs[j], 0.85, mode)
                    want = orc.match(host[i][0], host[i][1], host[j][0], host[j][1], 0.85, code)
                    for key in want:
                        assert np.array_equal(got[key], want[key]), (hook_on, i, j, mode, key)

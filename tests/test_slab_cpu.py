"""CPU tests of the z-slab plan and of the communication wrapper (gloo, world_size 2, rendezvous on 127.0.0.1).
The HIP contexts are replaced by fake workers whose "level buffers" are CPU tensors holding plane ids."""
import importlib
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
slab = importlib.import_module("3dsift_amd.slab")
capi = importlib.import_module("3dsift_amd.capi")
from test_dist_cpu import free_port  # noqa: E402


def test_octave_count_matches_reference_rule():
    # (int)log2f(min) - 2, Src/cSIFT3D.cc:254-255
    assert slab.octaves_total(64, 64, 64) == 4
    assert slab.octaves_total(512, 512, 512) == 7
    assert slab.octaves_total(1024, 1024, 512) == 7
    assert slab.octaves_total(100, 90, 120) == 4
    assert slab.octaves_total(7, 64, 64) == 0


@pytest.mark.parametrize("nz,world", [(512, 8), (512, 1), (160, 3), (129, 4), (64, 8), (16, 8)])
def test_slab_bounds_cover_even_aligned(nz, world):
    b = slab.slab_bounds(nz, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == nz
    for (a0, a1), (b0, b1) in zip(b, b[1:]):
        assert a1 == b0
    for z0, z1 in b:
        assert z0 % 2 == 0 and z1 > z0 and (z1 % 2 == 0 or z1 == nz)
    sizes = [z1 - z0 for z0, z1 in b]
    assert max(sizes) - min(sizes) <= 3
    # decimated planes (2k) of a slab stay inside it and tile octave 1
    cnt = [min(z1 // 2, nz // 2) - z0 // 2 for z0, z1 in b]
    assert sum(cnt) == nz // 2


def test_slab_bounds_refuses_too_many_ranks():
    with pytest.raises(ValueError):
        slab.slab_bounds(14, 8)


def test_bounds_for_two_sharded_octaves_stay_aligned():
    b0 = slab.slab_bounds(512, 8, align=4)
    assert all(z0 % 4 == 0 for z0, _ in b0) and b0[-1][1] == 512
    b1 = slab.halve_bounds(b0, 512)
    assert b1[0][0] == 0 and b1[-1][1] == 256 and all(z0 % 2 == 0 for z0, _ in b1)
    for (a0, a1), (c0, c1) in zip(b1, b1[1:]):
        assert a1 == c0
    # odd depth: the last plane of octave 0 has no image in octave 1
    b0 = slab.slab_bounds(161, 3, align=4)
    b1 = slab.halve_bounds(b0, 161)
    assert b0[-1][1] == 161 and b1[-1][1] == 80 and sum(z1 - z0 for z0, z1 in b1) == 80


@pytest.mark.parametrize("nz,world,h", [(512, 8, 38), (128, 4, 38), (160, 3, 8), (64, 8, 38)])
def test_halo_transfers_fill_exactly_the_halo(nz, world, h):
    b = slab.slab_bounds(nz, world)
    for lo in (0, 3):
        tr = slab.halo_transfers(b, nz, slab.KIND_GSS, 2, lo, h)
        for r, (z0, z1) in enumerate(b):
            want = set(range(max(0, z0 - h), max(0, z0 - lo))) | set(range(min(nz, z1 + lo), min(nz, z1 + h)))
            got = []
            for t in tr:
                if t.dst == r:
                    q0, q1 = b[t.src]
                    assert q0 <= t.zg0 < t.zg1 <= q1 and t.src != r  # the sender owns what it sends
                    got += list(range(t.zg0, t.zg1))
            assert len(got) == len(set(got)) and set(got) == want
    assert slab.halo_transfers(b, nz, 1, 0, 5, 5) == []


def test_described_rows_partition_everything_once():
    lv = np.array([1, 1, 2, 3, 3, 1, 2, 2, 3, 1, 1, 2, 3])
    parts = [slab.described_rows(lv, r, 3) for r in range(3)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(lv)))
    # level-descending stable order: positions 0..3 are the four level-3 rows in index order -> ranks 0,1,2,0
    assert parts[0].tolist() == [1, 3, 7, 10, 12] and 4 in parts[1] and 8 in parts[2]


def test_merge_keypoints_restores_reference_order():
    rng = np.random.default_rng(0)
    n = 200
    kp = np.zeros(n, capi.KP_DTYPE)
    kp["level"] = rng.integers(1, 4, n); kp["z"] = rng.integers(1, 63, n); kp["y"] = rng.integers(1, 63, n); kp["x"] = rng.integers(1, 63, n)
    _, first = np.unique(np.stack([kp["level"], kp["z"], kp["y"], kp["x"]], 1), axis=0, return_index=True)
    kp = kp[np.sort(first)]
    order = np.lexsort((kp["x"], kp["y"], kp["z"], kp["level"]))
    kp = kp[order]
    ds = np.arange(len(kp) * 768, dtype=np.float32).reshape(-1, 768)
    parts = []
    for z0, z1 in slab.slab_bounds(64, 4):
        m = (kp["z"] >= z0) & (kp["z"] < z1)
        parts.append((kp[m], ds[m]))   # per-slab lists are (level, z, y, x)-ordered like the device lists
    tail_kp = np.zeros(3, capi.KP_DTYPE); tail_kp["octave"] = 1
    tail = (tail_kp, np.ones((3, 768), np.float32))
    mk, md = slab.merge_keypoints(parts, tail)
    assert np.array_equal(mk[: len(kp)], kp) and np.array_equal(md[: len(kp)], ds)
    assert np.array_equal(mk[len(kp):], tail_kp)


WORKER = textwrap.dedent("""
    import importlib, os, sys, numpy as np, torch
    sys.path.insert(0, %r)
    import torch.distributed as dist
    d = importlib.import_module("3dsift_amd.dist")
    slab = importlib.import_module("3dsift_amd.slab")
    rank, world = d.init_from_env(backend="gloo")
    nz, halo, plane = 40, 9, 6
    bounds = slab.slab_bounds(nz, world)
    z0, z1 = bounds[rank]

    class FakeWorker:
        # two buffers of planes [z0-halo, z1+halo); owned planes hold 1000*buffer + global plane id, halo planes -1
        def __init__(self):
            self.zoff = z0 - halo
            self.buf = {}
            for key in ((1, 0), (2, 1)):
                t = torch.full((z1 - z0 + 2 * halo, plane), -1.0)
                for z in range(z0, z1):
                    t[z - self.zoff] = 1000.0 * key[0] + z
                self.buf[key] = t.reshape(-1)
        def view(self, kind, idx, a, b, stage=0):
            return self.buf[(kind, idx)][(a - self.zoff) * plane:(b - self.zoff) * plane]

    w = FakeWorker()
    comm = slab.DistComm()
    assert comm.local_ranks() == [rank] and comm.world == world
    # urgent + deferred groups in flight together, waited in order
    h1 = comm.exchange([w], slab.halo_transfers(bounds, nz, 1, 0, 0, 3))
    h2 = comm.exchange([w], slab.halo_transfers(bounds, nz, 1, 0, 3, halo) + slab.halo_transfers(bounds, nz, 2, 1, 0, 1))
    comm.wait(h1); comm.wait(h2)
    g = w.buf[(1, 0)].reshape(-1, plane)
    for z in range(z0 - halo, z1 + halo):
        want = 1000.0 + z if 0 <= z < nz else -1.0
        assert (g[z - w.zoff] == want).all(), (rank, z, g[z - w.zoff])
    dg = w.buf[(2, 1)].reshape(-1, plane)
    for z in range(z0 - halo, z1 + halo):
        inside = z0 - 1 <= z < z1 + 1 and 0 <= z < nz
        assert (dg[z - w.zoff] == (2000.0 + z if inside else -1.0)).all()
    # reductions
    m = comm.allreduce_max([np.array([1.0 + rank, 5.0 - rank], np.float32)])
    assert m[0].tolist() == [2.0, 5.0]
    cnt = [3, 2]
    mine = torch.full((3, 4), float(rank))
    out = torch.zeros((5, 4))
    comm.allgather_planes([mine], [out], cnt)
    assert out[:3].eq(0).all() and out[3:].eq(1).all()
    eq = torch.full((2, 4), float(rank)); out2 = torch.zeros((4, 4))
    comm.allgather_planes([eq], [out2], [2, 2])
    assert out2[:2].eq(0).all() and out2[2:].eq(1).all()
    rows = torch.zeros(6, 3); rows[rank::2] = 1.0 + torch.arange(6.)[rank::2, None]
    comm.allreduce_sum_([rows])
    assert torch.equal(rows, (1.0 + torch.arange(6.))[:, None].expand(6, 3))
    objs = comm.gather_objects([("r", rank)])
    assert objs == [("r", 0), ("r", 1)]
    dist.barrier()
    print("rank", rank, "ok")
""") % ROOT


def test_gloo_world2_halo_exchange_and_reductions():
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_window_neighbours_plan():
    """r05: which ranks take part in the descriptor windows of a rank's keypoints (owned planes within the window's reach)"""
    b = slab.slab_bounds(512, 8, align=4)          # 64-plane slabs, reach 39: one neighbour either side
    n = slab.window_neighbours(b, 39)
    assert n[0] == [1] and n[3] == [2, 4] and n[7] == [6]
    b2 = slab.halve_bounds(b, 512)                 # 32-plane slabs of octave 1: two hops
    n2 = slab.window_neighbours(b2, 39)
    assert n2[0] == [1, 2] and n2[4] == [2, 3, 5, 6]
    for nn in (n, n2):                             # symmetric
        for r, qs in enumerate(nn):
            assert all(r in nn[q] for q in qs)
    assert slab.window_neighbours([(0, 10), (10, 10), (10, 30)], 5) == [[2], [], [0]]   # an empty slab takes no part


PARTIAL_WORKER = textwrap.dedent("""
    import ctypes, importlib, os, sys, types, numpy as np, torch
    sys.path.insert(0, %r)
    import torch.distributed as dist
    d = importlib.import_module("3dsift_amd.dist")
    slab = importlib.import_module("3dsift_amd.slab")
    capi = importlib.import_module("3dsift_amd.capi")
    rank, world = d.init_from_env(backend="gloo")
    RW = capi.slab_record_words()

    def arr(ptr, n, ct, dt):
        return np.frombuffer((ct * n).from_address(ptr), dtype=dt) if n else np.zeros(0, dt)

    class FakeCtx:
        # the z part of rank `me`: hist[k][e] = 1000 * record id + 10 * me + e %% 7 (+ 500 with a second-round unit), mass = id + me / 8
        def __init__(self, me): self.me, self.finished = me, []
        def describe_partial(self, lists):
            assert lists[0][5:7] == bounds[self.me]                 # this rank's own list comes first ...
            assert [t[5:7] for t in lists[1:]] == [bounds[1 - self.me]]   # ... then its neighbour's, each with its owner's planes
            for rec_ptr, n, units_ptr, hist_ptr, mass_ptr, _z0, _z1 in lists:
                rec = arr(rec_ptr, n * RW, ctypes.c_int32, np.int32).reshape(n, RW)
                h = arr(hist_ptr, n * 768, ctypes.c_int32, np.int32).reshape(n, 768)
                m = arr(mass_ptr, n, ctypes.c_float, np.float32)
                un = arr(units_ptr, n, ctypes.c_float, np.float32) if units_ptr else None
                for k in range(n):
                    h[k] = 1000 * rec[k, 0] + 10 * self.me + np.arange(768) %% 7 + (500 if un is not None and un[k] == 0.25 else 0)
                    m[k] = rec[k, 0] + self.me / 8.0
        def describe_finish(self, rec_ptr, n, parts, units_ptr=None, final_round=False, redo_ptr=None, units_next_ptr=None):
            rec = arr(rec_ptr, n * RW, ctypes.c_int32, np.int32).reshape(n, RW).copy() if n else np.zeros((0, RW), np.int32)
            assert len(parts) == (2 if n else 0)            # the owner's part and its neighbour's, ascending rank
            h = sum(arr(hp, n * 768, ctypes.c_int32, np.int32).reshape(n, 768).astype(np.int64) for hp, _ in parts) if n else None
            m = None
            for _, mp in parts:                              # (the library adds the masses in the order given)
                t = arr(mp, n, ctypes.c_float, np.float32).copy()
                m = t if m is None else (m + t).astype(np.float32)
            self.finished.append((rec, h, m, final_round))
            k = 0
            if n and not final_round:
                redo = arr(redo_ptr, n, ctypes.c_int32, np.int32); nxt = arr(units_next_ptr, n, ctypes.c_float, np.float32)
                for i in range(n):
                    if rec[i, 1] == 1: redo[i] = 1; nxt[i] = 0.25; k += 1
            return k

    bounds = [(0, 20), (20, 40)]
    neigh = slab.window_neighbours(bounds, 6)
    assert neigh == [[1], [0]]
    counts = [3, 2]
    me = types.SimpleNamespace(rank=rank, arena=types.SimpleNamespace(device=torch.device("cpu")))
    ctx = FakeCtx(rank)
    sts = {rank: types.SimpleNamespace(ctx=ctx, bounds=bounds)}
    rec = torch.zeros((counts[rank], RW), dtype=torch.int32)
    rec[:, 0] = 100 * (rank + 1) + torch.arange(counts[rank], dtype=torch.int32)   # record id
    rec[:, 1] = torch.tensor([0, 1, 0][: counts[rank]], dtype=torch.int32)          # "the first unit fails"
    ex = types.SimpleNamespace(comm=slab.DistComm(), world=world)
    n_redo, redo, units_next = slab.SlabExtractor._partial_round(ex, [me], sts, neigh, counts, {rank: rec}, None, False)
    assert n_redo[rank] == 1 and redo[rank].tolist() == [0, 1, 0][: counts[rank]]
    frec, h, m, fin = ctx.finished[0]
    assert not fin and np.array_equal(frec, rec.numpy())
    for k in range(counts[rank]):
        rid = int(rec[k, 0])
        assert np.array_equal(h[k], 2 * 1000 * rid + 10 * (0 + 1) + 2 * (np.arange(768) %% 7)), (rank, k)     # both parts' integers
        assert m[k] == np.float32(np.float32(rid + 0 / 8.0) + np.float32(rid + 1 / 8.0))                       # masses in rank order
    tot = ex.comm.allgather_ints([n_redo[rank]])
    assert tot == [1, 1]
    idx = torch.nonzero(redo[rank]).flatten()
    rec2, un2 = rec.index_select(0, idx).contiguous(), units_next[rank].index_select(0, idx).contiguous()
    slab.SlabExtractor._partial_round(ex, [me], sts, neigh, tot, {rank: rec2}, {rank: un2}, True)
    frec, h, m, fin = ctx.finished[1]
    assert fin and len(frec) == 1 and int(frec[0, 0]) == 100 * (rank + 1) + 1
    assert np.array_equal(h[0], 2 * 1000 * int(frec[0, 0]) + 10 + 2 * (np.arange(768) %% 7) + 1000)   # both parts saw the owner's unit
    dist.barrier()
    print("rank", rank, "ok")
""") % ROOT


def test_gloo_world2_partial_descriptor_windows_protocol():
    """r05: records to the z-neighbours, partial integer histograms + masses back, the owner's sums in rank order, the second round of
    the flagged records with the owner's units -- the driver code of slab.py over gloo with a stand-in for the device context."""
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", PARTIAL_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o

"""CPU tests of the N>1 path with the gloo backend (world_size 2, rendezvous on 127.0.0.1)."""
import importlib
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib, os, sys, torch
    sys.path.insert(0, %r)
    d = importlib.import_module("3dsift_amd.dist")
    rank, world = d.init_from_env(backend="gloo")
    assert world == 2
    # ragged all-gather of descriptors: rank r holds (3 + 2r) rows whose values encode (rank, row)
    n = 3 + 2 * rank
    t = torch.arange(n * 768, dtype=torch.float32).reshape(n, 768) + 1000.0 * rank
    parts = d.allgather_ragged(t)
    assert [p.shape[0] for p in parts] == [3, 5]
    for r, p in enumerate(parts):
        want = torch.arange(p.shape[0] * 768, dtype=torch.float32).reshape(-1, 768) + 1000.0 * r
        assert torch.equal(p, want)
    # max-over-ranks timing
    assert d.max_over_ranks(1.0 + rank) == 2.0
    # pair deal covers every ordered pair exactly once
    mine = d.my_pairs(rank, world)
    import torch.distributed as dist
    allp = [None, None]
    dist.all_gather_object(allp, mine)
    flat = sorted(p for l in allp for p in l)
    assert flat == sorted(d.ordered_pairs(world)) == [(0, 1), (1, 0)]
    dist.barrier()
    print("rank", rank, "ok")
""") % ROOT


WORKER_ALLPAIRS = textwrap.dedent("""
    import importlib, os, sys, pickle, numpy as np, torch
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    d = importlib.import_module("3dsift_amd.dist")
    import oracle_lib as ol
    orc = ol.load("orc"); orc.set_threads(2)
    rank, world = d.init_from_env(backend="gloo")
    sets = pickle.load(open(os.environ["S3D_SETS"], "rb"))          # [(desc, xyz)] per volume; this rank owns volume `rank`
    desc, xyz = sets[rank]
    fn = lambda a, ax, b, bx: orc.match(a.numpy(), ax.numpy(), b.numpy(), bx.numpy(), 0.85, 3)
    mine = d.allpairs_match(torch.from_numpy(desc), torch.from_numpy(xyz), fn)
    pickle.dump(mine, open(os.environ["S3D_OUT"] + str(rank), "wb"))
    import torch.distributed as dist
    dist.barrier()
    print("rank", rank, "ok", sorted(mine))
""") % (ROOT, ROOT)


def test_gloo_world2_allpairs_equals_single_process(tmp_path):
    """BASELINE configs[4] on the CPU: every rank contributes the descriptors of its own volume (ragged), the ordered pairs are dealt
    to the ranks, and the union of the per-rank enhancedMatch results equals the single-process results (the oracle's matcher is the
    match function here; the GPU bench passes sift3d_match with device-resident inputs to the same code)."""
    import pickle
    import sys as _sys
    import numpy as np
    _sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    orc = ol.load("orc")
    rng = np.random.Generator(np.random.PCG64(3))
    base = np.clip(rng.normal(0.02, 0.03, size=(60, 768)), 0, None).astype(np.float32)
    sets = []
    for n in (40, 53):   # ragged: volume 1 holds more keypoints; both are noisy views of the same 60 features
        idx = rng.permutation(60)[:n] if n <= 60 else np.arange(60)
        dsc = base[idx % 60][:n] + rng.normal(0, 0.004, size=(n, 768)).astype(np.float32)
        dsc = np.concatenate([dsc, np.clip(rng.normal(0.02, 0.03, size=(max(0, n - len(dsc)), 768)), 0, None).astype(np.float32)])[:n]
        dsc = (dsc / np.linalg.norm(dsc, axis=1, keepdims=True)).astype(np.float32)
        sets.append((dsc, rng.uniform(0, 100, (n, 3)).astype(np.float32)))
    pickle.dump(sets, open(tmp_path / "sets.pkl", "wb"))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   S3D_SETS=str(tmp_path / "sets.pkl"), S3D_OUT=str(tmp_path / "out"))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER_ALLPAIRS], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
    got = {}
    for r in range(2):
        got.update(pickle.load(open(str(tmp_path / "out") + str(r), "rb")))
    assert sorted(got) == [(0, 1), (1, 0)]
    for (i, j), res in got.items():
        want = orc.match(sets[i][0], sets[i][1], sets[j][0], sets[j][1], 0.85, 3)
        assert len(want["pairs"]) > 5
        for k in want:
            assert np.array_equal(res[k], want[k]), (i, j, k)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_world2_allgather_and_pairs():
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_pair_deal_8():
    d = importlib.import_module("3dsift_amd.dist")
    pairs = d.ordered_pairs(8)
    assert len(pairs) == 56
    dealt = [d.my_pairs(r, 8) for r in range(8)]
    assert all(len(x) == 7 for x in dealt)
    assert sorted(p for x in dealt for p in x) == sorted(pairs)

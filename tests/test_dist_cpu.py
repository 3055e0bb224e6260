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

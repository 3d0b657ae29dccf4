"""N>1 path on CPU: world_size-2/3 gloo runs of the slice + all-reduce logic that bench.py and
graphite_amd.tr_hits use on the GPUs (there the slice evaluator is gr_tr_hits; here a table lookup)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphite_amd import engine


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, C, T, target, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    preds = np.random.RandomState(11).randint(0, 5, size=C * T)       # the same "classifier" on every rank
    calls = []

    def eval_slice(begin, count):
        calls.append((begin, count))
        hits = np.zeros(C, np.int32)
        sl = preds[begin:begin + count]
        np.add.at(hits, (np.arange(begin, begin + count) // T)[sl == target], 1)
        return torch.from_numpy(hits)

    out = engine.sharded_hits(eval_slice, C * T, distributed=True)
    plain = engine.sharded_hits(eval_slice, C * T)           # opt-in: without the flag an initialised group is ignored
    assert calls.pop() == (0, C * T) and plain.sum() == (preds == target).sum()
    q.put((rank, out.numpy().tolist(), calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,C,T", [(2, 7, 13), (2, 1, 100), (3, 5, 10), (2, 3, 1)])
def test_sharded_hits_gloo(world, C, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, T, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    preds = np.random.RandomState(11).randint(0, 5, size=C * T)
    want = (preds.reshape(C, T) == 2).sum(1).tolist()
    covered = []
    for rank, hits, calls in res:
        assert hits == want                                 # every rank ends with the full vector
        assert len(calls) == 1
        covered.append(calls[0])
    covered.sort()
    pos = 0
    for b, c in covered:                                    # slices tile [0, C*T) exactly
        assert b == pos
        pos += c
    assert pos == C * T
    sizes = [c for _, c in covered]
    assert max(sizes) - min(sizes) <= 1


def _disagree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # first an agreeing call of shape A on both ranks (a per-shape cache of the check would now hold A on both) ...
    engine.sharded_hits(lambda b, c: torch.zeros(4, dtype=torch.int32), 40, distributed=True)
    # ... then rank 0 repeats A while rank 1 believes there is one more candidate: the check runs on EVERY call
    C = 4 + rank
    try:
        engine.sharded_hits(lambda b, c: torch.zeros(C, dtype=torch.int32), C * 10, distributed=True)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_disagree_on_the_work_fail_loudly():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_disagree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("ranks disagree" in msg for _, msg in res), res


def test_shard_range_properties():
    for total in (0, 1, 7, 100, 4096000):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                b, c = engine.shard_range(total, r, world)
                assert b == pos and c >= 0
                pos += c
            assert pos == total


def test_single_process_is_unsharded():
    got = engine.sharded_hits(lambda b, c: torch.tensor([b, c]), 42)
    assert got.tolist() == [0, 42]

"""RCCL smoke on the one GPU a test box has: a single-rank `nccl` process group runs the int32[C] all-reduce (and the
work-agreement check) that the sharded tr_hits path issues on a multi-GPU node.  It cannot show scaling, but it does show
that the collective the N > 1 path depends on initialises, accepts an int32 device vector and leaves the counts intact."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from cases import gtsrb_case, gtsrb_candidates
from graphite_amd import engine, models, synth

pytestmark = pytest.mark.gpu


def test_single_rank_nccl_allreduce_of_hit_counts():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        case = gtsrb_case()
        masks, thetas = gtsrb_candidates(case, 5)
        m = models.GTSRBNet()
        m.load_state_dict(synth.gtsrb_state_dict())
        plain = engine.tr_hits(m, case["vic"], masks, thetas, case["tab"], 1)
        # world size 1: sharded_hits short-circuits, so drive the collective pieces directly on the device vector
        h = plain.clone()
        engine._assert_same_work(5 * 100, h, None)
        out = engine.allreduce_hits(h)                     # world 1 -> returns as is
        dist.all_reduce(h, op=dist.ReduceOp.SUM)            # the RCCL call itself, int32 on the GPU
        torch.cuda.synchronize()
        assert h.dtype == torch.int32 and torch.equal(h, plain) and torch.equal(out, plain)
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py's max-over-ranks timing reduction
        dist.barrier()
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()

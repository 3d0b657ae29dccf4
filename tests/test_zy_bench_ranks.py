"""bench.py's rank path on the GPU box: the self-launched two-rank run (both ranks on the one GPU, gloo -- a rehearsal of
the sharding, not a scaling measurement) must give the single-rank hit counts, and its line must list both ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--candidates", "48", "--transforms", "100", "--chunk", "1600", "--steps", "1", "--warmup", "0", "--no-extras",
        "--no-cpu-baseline", "--no-profile"]


def _run(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rank-limit-s", "200"] + ARGS,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_self_launched_ranks_reproduce_the_single_rank_counts():
    one, two = _run(1), _run(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["hits_checksum"] == two["hits_checksum"] and one["hits_checksum"] > 0
    assert [r["rank"] for r in two["ranks_seen"]] == [0, 1]
    assert len({r["pid"] for r in two["ranks_seen"]}) == 2
    assert two["config"]["images_per_step"] == 4800 and two["scaling"] == "strong"

"""Host logic of bench.py's self-launcher (no GPU needed): GPUs are counted without the HIP runtime, and a failing rank
ends the run with its stderr tail instead of a hang."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_visible_gpu_count_reads_the_environment_not_the_runtime(monkeypatch):
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4")                 # the more specific variable wins
    assert bench.visible_gpu_count() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def spawn_ranks("):src.index("def ranks_seen(")]
    assert "torch.cuda" not in body and "hipGetDeviceCount" not in body    # the parent never touches the runtime


def test_failing_ranks_end_the_run_with_their_stderr_tail():
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the ranks would run the real job")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rank-limit-s", "120"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench.py: rank" in r.stderr and "stderr tail" in r.stderr
    assert r.stdout.strip() == ""                                  # no JSON line from a failed run

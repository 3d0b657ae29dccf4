#!/usr/bin/env python3
"""Per-layer timing of the GTSRBNet forward (8192 synthetic images) under the current
GRAPHITE_CONV / GRAPHITE_HALO_VARIANT settings; prints TFLOP/s per kernel.  Tuning aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphite_amd import _hip, models, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = models.GTSRBNet()
m.load_state_dict(synth.gtsrb_state_dict())
_hip.set_chunk(max(64, N))          # one launch per layer, as bench.py does with --chunk
x = torch.rand(N, 3, 32, 32, device="cuda") - 0.5
for _ in range(2):
    m.forward(x)
torch.cuda.synchronize()
_hip.profile_reset(); _hip.profile_enable(True)
for _ in range(5):
    m.forward(x)
torch.cuda.synchronize()
_hip.profile_enable(False)
prof = _hip.profile_read()
tot = 0.0
line = []
for k, v in prof.items():
    ms = v["ms"] / v["launches"]
    tot += v["ms"] / 5
    line.append("%s %.3fms %.0fTF" % (k.replace("k_conv_", ""), ms, v["flops"] / v["ms"] / 1e9 if v["ms"] else 0))
print(os.environ.get("GRAPHITE_HALO_VARIANT", "-"), "| total %.3f ms/fwd %.0f img/s |" % (tot, N / tot * 1e3), " ; ".join(line), flush=True)

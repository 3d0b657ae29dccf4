#!/usr/bin/env python3
"""Per-kernel timing of a WideResNet forward (synthetic weights) through the C ABI; prints TFLOP/s per kernel.
usage: tune_wrn.py [N images] [depth] [widen]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphite_amd import _hip, models, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 34
widen = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = models.WideResNet(depth=depth, widen_factor=widen)
m.load_state_dict(synth.wrn_state_dict("wrn-v1", depth, widen, 10, include_unused=True))
if os.environ.get("GRAPHITE_PREC") == "bf16":      # the throughput mode (stride-1 3x3 convs on bf16 operands)
    m.set_precision("bf16")
x = torch.rand(N, 3, 32, 32, device="cuda")
m.forward(x)
torch.cuda.synchronize()
REP = 2
_hip.profile_reset(); _hip.profile_enable(True)
for _ in range(REP):
    m.forward(x)
torch.cuda.synchronize()
_hip.profile_enable(False)
prof = _hip.profile_read()
tot = sum(v["ms"] for v in prof.values()) / REP
fl = sum(v["flops"] for v in prof.values()) / REP
print("WRN-%d-%d N=%d | total %.3f ms/fwd %.0f img/s %.1f TF overall" % (depth, widen, N, tot, N / tot * 1e3, fl / tot / 1e9), flush=True)
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-44s launches %4d  %9.3f ms/launch  %6.1f TF  %5.1f%% of time" % (
        k, v["launches"], v["ms"] / v["launches"], v["flops"] / v["ms"] / 1e9 if v["ms"] else 0, 100 * v["ms"] / REP / tot), flush=True)

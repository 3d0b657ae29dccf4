#!/usr/bin/env python3
"""Per-kernel timing of the PatchGuard classifier (bicubic + BagNet-17 + masking_defense) on N synthetic 32x32 inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from graphite_amd import _hip, synth
from graphite_amd.patchguard.patchguard_predict import PatchGuard

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = PatchGuard(synth.bagnet_state_dict())
if len(sys.argv) > 2:
    m.set_precision(sys.argv[2])           # "bf16": the throughput mode
x = torch.rand(N, 3, 32, 32, device="cuda") - 0.5
for _ in range(2):
    m.predict(x)
torch.cuda.synchronize()
_hip.profile_reset(); _hip.profile_enable(True)
for _ in range(3):
    p = m.predict(x)
torch.cuda.synchronize()
_hip.profile_enable(False)
prof = _hip.profile_read()
tot = sum(v["ms"] for v in prof.values()) / 3
fl = sum(v["flops"] for k, v in prof.items() if k.startswith(("k_conv", "k_gemm"))) / 3
print("N %d: %.2f ms per forward, %.0f images/s, convs %.1f TFLOP/s; checksum %d" % (N, tot, N / tot * 1e3, fl / tot / 1e9, int(p.sum())))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-40s %4d launches %9.3f ms  %6.1f TF  %7.1f GB/s (algorithmic)" % (k, v["launches"] // 3, v["ms"] / 3, v["flops"] / v["ms"] / 1e9 if v["ms"] else 0, v["bytes"] / v["ms"] / 1e6 if v["ms"] else 0))

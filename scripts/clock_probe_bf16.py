#!/usr/bin/env python3
"""GTSRBNet forwards in the bf16 mode, back to back for ~8 s (32,000 images each), while a child process samples
rocm-smi's shader clock (read-only): does the clock give under the bf16 MFMA load?  Prints the samples and the rate."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from graphite_amd import _hip, models, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = 32000
_hip.set_chunk(N)
model = models.GTSRBNet().load_state_dict(synth.gtsrb_state_dict())
model.set_precision(prec)
x = torch.rand(N, 3, 32, 32, device="cuda") - 0.5
model.forward(x); torch.cuda.synchronize()
sampler = subprocess.Popen(["bash", "-c", "for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep sclk | tr -s ' '; sleep 0.5; done"],
                           stdout=subprocess.PIPE, text=True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 8.0:
    for _ in range(20):
        model.forward(x)
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
out = sampler.communicate()[0]
print(f"{prec}: {n} forwards of {N} images in {dt:.2f} s = {n * N / dt:,.0f} images/s")
print(out)

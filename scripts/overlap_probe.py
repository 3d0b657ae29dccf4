#!/usr/bin/env python3
"""Does the warp stage overlap with the classifier when they run on two HIP streams?  (Timing probe only: both calls use
the context's one workspace, so the concurrent run's VALUES are garbage.)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from cases import gtsrb_case, gtsrb_candidates
from graphite_amd import _hip, models, synth
c = gtsrb_case(num_xforms=1000)
masks, thetas = gtsrb_candidates(c, 32)
m = models.GTSRBNet(); m.load_state_dict(synth.gtsrb_state_dict())
_hip.set_chunk(32000)
x = torch.rand(32000, 3, 32, 32, device="cuda") - 0.5
img, md, td = _hip.to_dev(c["vic"]), _hip.to_dev(masks), _hip.to_dev(thetas)
xf = _hip.xform_table(c["tab"])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def A(n=4):
    with torch.cuda.stream(sa):
        for _ in range(n): m.hip_model.forward(x)
def B(n=40):
    with torch.cuda.stream(sb):
        for _ in range(n): _hip.warp_batch_dev(img, md, td, xf, 1.0, True)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
A(1); B(2)
ta, tb = timed(A), timed(B)
tab = timed(lambda: (A(), B()))
print("classifier alone %.1f ms, warp alone %.1f ms, both streams %.1f ms (sum %.1f)" % (ta, tb, tab, ta + tb))

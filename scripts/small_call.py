#!/usr/bin/env python3
"""Wall time of one small evaluator call (C candidates x 100 transforms) against the sum of its kernels' durations:
how much of a small call is launch gaps / host work rather than kernels."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from cases import gtsrb_case
from graphite_amd import _hip, models, synth
from graphite_amd.evaluator import TREvaluator
c = gtsrb_case(num_xforms=100)
m = models.GTSRBNet(); m.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
ev = TREvaluator(m, torch.from_numpy(c["vic"]), c["xf"], c["pt_file"], 'GTSRB', 14, 27)
for C in (1, 2, 4, 8, 11, 32):
    masks = [torch.from_numpy(c["hull"])] * C
    thetas = [torch.from_numpy(((c["tar_s"] - c["vic_s"]) * c["hull"]).astype(np.float32))] * C
    for _ in range(5): ev.errors(masks, thetas)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): ev.errors(masks, thetas)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    _hip.profile_reset(); _hip.profile_enable(True)
    for _ in range(n): ev.errors(masks, thetas)
    torch.cuda.synchronize()
    _hip.profile_enable(False)
    prof = _hip.profile_read()
    ks = sum(v["ms"] for v in prof.values()) / n
    print("C=%2d  wall %.3f ms per call, kernels %.3f ms, rest %.3f ms" % (C, wall, ks, wall - ks))
    if len(sys.argv) > 1:                              # per kernel: microseconds per call (launches per call)
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            print("      %-40s %7.1f us  (%g launches)" % (k, v["ms"] / n * 1e3, v["launches"] / n))

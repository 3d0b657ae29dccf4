#!/usr/bin/env python3
"""The 16x16x4 linear-layer kernel of small launches (fc_small.h) against the big tiles: logits of the first rows of a
large batch (igemm path) must equal those of the same images sent alone (fc_small path), bit for bit; then the time of
the three linear layers per call.  GRAPHITE_FC16_VARIANT / GRAPHITE_FC16_ROWS select the tiling / the switch-over."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from graphite_amd import _hip, models, synth
m = models.GTSRBNet(); m.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
g = torch.Generator().manual_seed(7)
x = (torch.rand((6000, 3, 32, 32), generator=g) - 0.5).cuda()
big = m.forward(x)
ok = True
for n in (1, 15, 16, 17, 100, 129, 800, 1024):
    small = m.forward(x[:n])
    same = torch.equal(small, big[:n])
    ok &= same
    print("rows %5d: logits == large-batch logits: %s" % (n, same))
print("variant", os.environ.get("GRAPHITE_FC16_VARIANT", "0"), "bit-identical" if ok else "MISMATCH")
for n in (100, 200, 300, 400, 600, 800, 1100, 1600, 2400, 3200):
    xs = x[:n]
    for _ in range(3): m.forward(xs)
    torch.cuda.synchronize()
    _hip.profile_reset(); _hip.profile_enable(True)
    for _ in range(20): m.forward(xs)
    torch.cuda.synchronize(); _hip.profile_enable(False)
    prof = _hip.profile_read()
    fc = {k: v["ms"] / 20 * 1e3 for k, v in prof.items() if "fc_small" in k or "igemm" in k}
    print("rows %5d: linear layers %s us per forward" % (n, {k: round(v, 1) for k, v in fc.items()}))
sys.exit(0 if ok else 1)

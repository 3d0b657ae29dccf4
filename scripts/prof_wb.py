#!/usr/bin/env python3
"""20 whitebox EOT steps (100 transforms each) for a kernel trace: k_wb_*, the GTSRBNet forward / transposed layers."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import torch
from cases import gtsrb_case, PT14
from graphite_amd import _hip, engine, models, synth, transforms as T
c = gtsrb_case(num_xforms=100)
orig = torch.from_numpy(0.04 + 0.92 * c["vic"])
att = torch.clamp(orig + 0.25 * torch.randn(orig.shape, generator=torch.Generator().manual_seed(0)), 0.03, None)
mask = torch.zeros(3, 244, 244); mask[:, 40:200, 30:215] = 1.0
tab = T.build_wb_table(c["xf"], 244, 244, PT14)
m = models.GTSRBNet(); m.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
od, ad, md = _hip.to_dev(orig), _hip.to_dev(att), _hip.to_dev(mask)
for _ in range(20):
    g, lg = engine.eot_gradient(m, od, ad, md, tab, 27)
torch.cuda.synchronize()
print("ok", float(g.abs().sum()))

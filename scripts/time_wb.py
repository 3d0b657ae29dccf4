#!/usr/bin/env python3
"""Time of one whitebox EOT step (100 transforms, GTSRBNet) and its kernels."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from graphite_amd import _hip, engine, models, scenes, synth, transforms as T
c = scenes.stop_sign_scene(100)
orig = torch.from_numpy(0.04 + 0.92 * c["vic"])
att = torch.clamp(orig + 0.25 * torch.randn(orig.shape, generator=torch.Generator().manual_seed(0)), 0.03, None)
mask = torch.zeros(3, 244, 244); mask[:, 40:200, 30:215] = 1.0
tab = T.build_wb_table(c["xf"], 244, 244, scenes.PT14)
m = models.GTSRBNet(); m.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
od, ad, md = _hip.to_dev(orig), _hip.to_dev(att), _hip.to_dev(mask)
for _ in range(3):
    engine.eot_gradient(m, od, ad, md, tab, 27)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g, lg = engine.eot_gradient(m, od, ad, md, tab, 27)
torch.cuda.synchronize()
print("EOT step %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
_hip.profile_reset(); _hip.profile_enable(True)
for _ in range(5):
    engine.eot_gradient(m, od, ad, md, tab, 27)
torch.cuda.synchronize(); _hip.profile_enable(False)
for k, v in sorted(_hip.profile_read().items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print("  %-40s %4d launches %8.3f ms" % (k, v["launches"] // 5, v["ms"] / 5))

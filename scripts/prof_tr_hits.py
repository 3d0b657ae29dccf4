#!/usr/bin/env python3
"""Small tr_hits workload for counter passes: 16 candidates x 1000 GTSRB transforms (two 8192-image chunks, the bench's
chunk shape), twice.  Exercises k_composite, k_warp_net, the GTSRBNet kernels and k_argmax_hits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from cases import gtsrb_case
from graphite_amd import _hip, engine, models, synth

case = gtsrb_case(num_xforms=1000)
C = 16
masks = synth.synth_patch_masks("bench-sweep", C, case["hull"])
thetas = ((case["tar_s"] - case["vic_s"])[None] * masks).astype(np.float32)
model = models.GTSRBNet()
model.load_state_dict(synth.gtsrb_state_dict())
img_d, md, td = _hip.to_dev(case["vic"]), _hip.to_dev(masks), _hip.to_dev(thetas)
xf = _hip.xform_table(case["tab"])
for _ in range(2):
    h = engine.tr_hits(model, img_d, md, td, xf, 1)
torch.cuda.synchronize()
print("hits", h.cpu().numpy().tolist())

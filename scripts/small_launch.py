#!/usr/bin/env python3
"""Cost of SMALL tr_hits launches (the attack's shape: C candidates x 100 transforms): wall time per call with the
inputs already on the device, and the per-kernel HIP-event times, for C = 1 .. 64."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from cases import gtsrb_case
from graphite_amd import _hip, engine, models, synth
case = gtsrb_case(num_xforms=100, target_id=27)
model = models.GTSRBNet(); model.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
if len(sys.argv) > 1: model.set_precision(sys.argv[1])
img = _hip.to_dev(case["vic"]); xf = _hip.xform_table(case["tab"])
for C in (1, 2, 4, 8, 16, 32, 64):
    masks = synth.synth_patch_masks("small", C, case["hull"])
    thetas = ((case["tar_s"] - case["vic_s"])[None] * masks).astype(np.float32)
    md, td = _hip.to_dev(masks), _hip.to_dev(thetas)
    for _ in range(3): engine.tr_hits(model, img, md, td, xf, 27).cpu()
    t0 = time.perf_counter()
    for _ in range(30): engine.tr_hits(model, img, md, td, xf, 27).cpu()
    wall = (time.perf_counter() - t0) / 30
    _hip.profile_reset(); _hip.profile_enable(True)
    for _ in range(10): engine.tr_hits(model, img, md, td, xf, 27).cpu()
    _hip.profile_enable(False); pr = _hip.profile_read()
    gpu = sum(v["ms"] for v in pr.values()) / 10
    print("C=%3d wall %.3f ms  sum of kernel times %.3f ms  | " % (C, 1e3 * wall, gpu) +
          " ".join("%s %.0f" % (k.replace("k_conv_halo", "h").replace("k_conv_igemm", "ig"), 100 * v["ms"]) for k, v in pr.items()), flush=True)

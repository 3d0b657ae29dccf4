#!/usr/bin/env python3
"""Wall-clock of the config-2 attack against the speculative window of the fine reduction (generate_mask.FINE_WINDOW)."""
import io, os, random, sys, time, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from graphite_amd.scenes import INP
from graphite_amd import main as gmain, models, parsearguments, synth, generate_mask as GM
a = parsearguments.getarguments(["-v", "14", "-t", "27", "--tr_lo", "0.65", "--tr_hi", "0.85", "-m", "100", "-b", "100", "--coarse_mode", "binary", "--seed", "42"])
model = models.GTSRBNet(); model.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
def run():
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        r = gmain.attack_network(model, os.path.join(INP, "GTSRB/images/14.png"), os.path.join(INP, "GTSRB/images/27.png"),
            os.path.join(INP, "GTSRB/Hulls/14.png"), 14, 27, os.path.join(INP, "GTSRB/Points/14.csv"), "score.py", "Target",
            a.coarse_error, a.reduce_error, 1, 100, 100, model_type='GTSRB', args=a, write_outputs=False)
    torch.cuda.synchronize()
    return r
run()
for w, wmin in ((32, 4), (16, 4), (16, 3), (16, 2), (8, 2), (8, 3), (12, 3), (24, 3), (32, 3), (32, 4)):
    GM.FINE_WINDOW, GM.FINE_WINDOW_MIN = w, wmin
    dt = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = run(); dt = min(dt, time.perf_counter() - t0)
    print("window max %3d min %2d: %.3f s  queries %d pixels %d TR %.2f" % (w, wmin, dt, r["queries"], r["pixels"], r["transform_robustness"]), flush=True)

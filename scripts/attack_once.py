#!/usr/bin/env python3
"""The config-2 attack twice (warm-up + measured), nothing else: the program to put under `rocprofv3 --kernel-trace --stats`
when the question is how much of the attack's wall-clock is kernel time."""
import contextlib, io, os, random, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from graphite_amd.scenes import INP
from graphite_amd import main as gmain, models, parsearguments, synth
a = parsearguments.getarguments(["-v", "14", "-t", "27", "--tr_lo", "0.65", "--tr_hi", "0.85", "-m", "100", "-b", "100", "--coarse_mode", "binary", "--seed", "42"])
model = models.GTSRBNet(); model.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
def run():
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        return gmain.attack_network(model, os.path.join(INP, "GTSRB/images/14.png"), os.path.join(INP, "GTSRB/images/27.png"),
            os.path.join(INP, "GTSRB/Hulls/14.png"), 14, 27, os.path.join(INP, "GTSRB/Points/14.csv"), "score.py", "Target",
            a.coarse_error, a.reduce_error, 1, 100, 100, model_type='GTSRB', args=a, write_outputs=False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2:
    from graphite_amd import _hip
    _hip.set_chunk(int(sys.argv[2]))           # images per classifier launch (library default 8192)
for i in range(n):
    t0 = time.perf_counter(); r = run(); torch.cuda.synchronize()
    print("run %d: %.3f s, %d queries" % (i, time.perf_counter() - t0, r["queries"]))

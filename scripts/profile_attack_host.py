#!/usr/bin/env python3
"""cProfile of the config-2 attack sorted by SELF time: where the host spends the time it does not spend waiting for the GPU."""
import cProfile, io, os, pstats, random, sys, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
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
run(); run()
pr = cProfile.Profile(); pr.enable(); r = run(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])

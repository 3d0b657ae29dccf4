#!/usr/bin/env python3
"""Trace of the fine reduction of the config-2 attack: per evaluated candidate the bits it would remove, the error rate
before and after and the outcome -- the data a speculation policy (generate_mask.get_fine_reduced_mask) has to predict."""
import io, os, random, sys, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + "/tests"): sys.path.insert(0, p)
import numpy as np, torch
from graphite_amd.scenes import INP
from graphite_amd import main as gmain, models, parsearguments, synth, score as S
a = parsearguments.getarguments(["-v", "14", "-t", "27", "--tr_lo", "0.65", "--tr_hi", "0.85", "-m", "100", "-b", "100", "--coarse_mode", "binary", "--seed", "42"])
model = models.GTSRBNet(); model.load_state_dict(synth.gtsrb_state_dict(readout="toy3"))
rows = []
_sf = S.score_fn
def _rec(theta, mask, tr_err, object_size, coord=None, **k):
    v = _sf(theta, mask, tr_err, object_size, coord, **k)
    rows.append((float(mask.sum() / 3), float(tr_err), float(k.get("best_score", -1)), float(v), float(object_size), coord))
    return v
S.score_fn = _rec
np.random.seed(42); torch.manual_seed(42); random.seed(42)
with contextlib.redirect_stdout(io.StringIO()):
    gmain.attack_network(model, os.path.join(INP, "GTSRB/images/14.png"), os.path.join(INP, "GTSRB/images/27.png"),
        os.path.join(INP, "GTSRB/Hulls/14.png"), 14, 27, os.path.join(INP, "GTSRB/Points/14.csv"), "score.py", "Target",
        a.coarse_error, a.reduce_error, 1, 100, 100, model_type='GTSRB', args=a, write_outputs=False)
bits0 = err0 = None
print("# bits_before err_before dbits err_cand outcome coord object_size")
for bits, err, best, v, obj, coord in rows:
    if best < 0:
        bits0, err0 = bits, err; print("# start bits %g err %g object %g" % (bits, err, obj)); continue
    acc = v < best
    print("%g %.2f %g %.2f %s %s" % (bits0, err0, bits0 - bits, err, "A" if acc else "r", coord))
    if acc: bits0, err0 = bits, err

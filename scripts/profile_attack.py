#!/usr/bin/env python3
"""cProfile of the whole config-2 attack (host-side overheads around the GPU launches)."""
import cProfile, io, os, pstats, random, sys, contextlib
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
run()
pr = cProfile.Profile(); pr.enable(); r = run(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])

# ---- per-launch log: candidates per evaluator call and the wall time of each (upload + kernels + download) ----
import time
from graphite_amd import evaluator as E
log = []
_orig = E.TREvaluator.errors
def _timed(self, masks, thetas):
    t0 = time.perf_counter(); r = _orig(self, masks, thetas); log.append((len(r), time.perf_counter() - t0)); return r
E.TREvaluator.errors = _timed
t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
E.TREvaluator.errors = _orig
import collections
by = collections.OrderedDict()
for c, dt in log:
    k = by.setdefault(c, [0, 0.0]); k[0] += 1; k[1] += dt
print("wall %.3f s, %d evaluator calls, %.3f s inside them, queries %d" % (wall, len(log), sum(d for _, d in log), r["queries"]))
for c, (n, dt) in sorted(by.items()):
    print("  C=%4d  calls %4d  total %.1f ms  per call %.3f ms  images/s %.0f" % (c, n, 1e3 * dt, 1e3 * dt / n, c * 100 * n / dt))

# ---- the accept / reject sequence of the fine reduction (what a speculation policy has to predict) ----
from graphite_amd import generate_mask as GM, score as S
seq = []
_sf = S.score_fn
def _rec(*a, **k):
    v = _sf(*a, **k)
    if "best_score" in k:
        seq.append("A" if v < k["best_score"] else "r")
    return v
S.score_fn = _rec
run()
S.score_fn = _sf
print("fine reduction outcomes (%d evaluated):" % len(seq))
print("".join(seq))

# ---- three-way split of the wall clock: device / discarded speculation / host ----
from graphite_amd import _hip
log = []
E.TREvaluator.errors = _timed
_hip.profile_reset(); _hip.profile_enable(True)
t0 = time.perf_counter(); r = run(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
_hip.profile_enable(False)
E.TREvaluator.errors = _orig
prof = _hip.profile_read()
kern = sum(v["ms"] for v in prof.values()) * 1e-3
inside = sum(d for _, d in log)
issued = int(r.get("queries_issued", r["queries"]))
need = int(r["queries"])
print("three-way split of %.3f s (profiling events on):" % wall)
print("  device kernels            %.3f s  (%d images issued; %.3f s of it for the %d discarded speculative images at the average rate)"
      % (kern, issued, kern * (issued - need) / issued, issued - need))
print("  evaluator calls, not kernel %.3f s  (%d calls: H2D of masks / thetas, launch gaps, the D2H sync of the hit counts)" % (inside - kern, len(log)))
print("  host outside the evaluator  %.3f s  (PNG / CSV decode, mask algebra, score_fn, RNG, driver control flow)" % (wall - inside))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print("    %-44s %5d launches %8.2f ms" % (k, v["launches"], v["ms"]))

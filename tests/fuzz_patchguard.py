#!/usr/bin/env python3
"""Randomised HIP-vs-oracle sweep of the PatchGuard classifier: random BagNet geometries (blocks per stage, strides,
3x3 placement, up-sampling size), random inputs, random masking windows / thresholds.  Bit for bit, or it prints the case.

    python tests/fuzz_patchguard.py [n_cases]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np                                           # noqa: E402

import oracle_lib as O                                       # noqa: E402
from graphite_amd import _hip, synth, weights                # noqa: E402
from graphite_amd.patchguard.patchguard_predict import PatchGuard, PatchGuardNoDefense   # noqa: E402


def run_case(seed):
    rng = np.random.RandomState(1000 + seed)
    layers = tuple(int(v) for v in rng.randint(1, 3, size=4))
    strides = tuple(int(v) for v in rng.randint(1, 3, size=4))
    kernel3 = tuple(int(v) for v in rng.randint(0, 2, size=4))
    arch = dict(layers=layers, strides=strides, kernel3=kernel3)
    size = int(rng.choice([40, 48, 56, 64, 72]))
    ncls = int(rng.choice([3, 10, 17]))
    n = int(rng.randint(1, 6))
    h = O.bagnet_out_size(size, **arch)
    if h < 1:
        return True, "skipped (empty map)"
    sd = synth.bagnet_state_dict("fuzz-%d" % seed, layers, kernel3, num_classes=ncls, calibrated=False)
    x = synth.synth_images("fuzz-pg-%d" % seed, n)
    tl = weights.bagnet_tensor_list(sd, layers)
    ref = O.bagnet_forward(tl, O.pg_preprocess(x, size), num_classes=ncls, **arch)
    win = int(rng.randint(1, h + 1))
    m = PatchGuard(sd, size=size, num_classes=ncls, window_size=win, **arch)
    got = m.local_logits(x).cpu().numpy()
    ok = np.array_equal(got, ref)
    s_ref, p_ref = O.masking_defense(ref, win, 0.0)
    ok = ok and m.predict(x).cpu().tolist() == p_ref.tolist()
    ok = ok and np.array_equal(m.hip_model.forward(x).cpu().numpy(), s_ref)
    nd = PatchGuardNoDefense(sd, size=size, num_classes=ncls, **arch)
    ok = ok and nd.predict(x).cpu().tolist() == O.pg_mean_logits(ref).argmax(1).tolist()
    # the masking kernel alone on centred random evidence (ties and empty classes are likelier there)
    lo = (rng.randn(n, h, h, ncls) * 2).astype(np.float32)
    lo[rng.rand(n, h, h, ncls) < 0.3] = 0.0
    thres = float(rng.choice([0.0, 0.1, 0.5]))
    a, b = O.masking_defense(lo, win, thres)
    c, d = _hip.pg_masking_defense(lo, win, thres)
    ok = ok and np.array_equal(c.cpu().numpy(), a) and np.array_equal(d.cpu().numpy(), b)
    return ok, "layers %s strides %s kernel3 %s size %d -> %dx%d map, %d classes, %d images, window %d" % (
        layers, strides, kernel3, size, h, h, ncls, n, win)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    t0, bad = time.time(), 0
    for s in range(n):
        ok, msg = run_case(s)
        bad += not ok
        print("case %2d %s  %s" % (s, "ok " if ok else "MISMATCH", msg), flush=True)
    print("%d cases, %d mismatches, %.0f s" % (n, bad, time.time() - t0))
    sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Randomised parity sweep (by hand on a GPU box for many seeds; tests/test_gpu_full_size.py runs 8 of them under pytest): for many seeds, transform-table flavours
and candidate sets, warped pixels and per-candidate hit counts from the HIP path must equal the CPU oracle's, bit for bit.

    python tests/fuzz_parity.py [n_seeds]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np                                           # noqa: E402

import oracle_lib as O                                       # noqa: E402
from cases import INP, PT14                                  # noqa: E402
from graphite_amd import _hip, engine, models, synth, transforms as T, weights   # noqa: E402
from graphite_amd.imageio import load_image_chw               # noqa: E402


_STATE = {}


def _setup():
    if not _STATE:
        sd = synth.gtsrb_state_dict(readout="toy3")
        model = models.GTSRBNet()
        model.load_state_dict(sd)
        _STATE.update(model=model, tl=weights.gtsrb_tensor_list(sd),
                      signs={k: load_image_chw(os.path.join(INP, f"GTSRB/images/{k}.png")) for k in (14, 1, 27)},
                      hull=(load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32))
    return _STATE


def run_case(seed):
    """One randomised case -> (pixels equal, hits and labels equal, description)."""
    st = _setup()
    model, tl, signs, hull = st["model"], st["tl"], st["signs"], st["hull"]
    rng = np.random.RandomState(1000 + seed)
    vic_id, tar_id = [(14, 1), (14, 27), (1, 27), (27, 14)][seed % 4]
    vic, tar = signs[vic_id], signs[tar_id]
    vic_s, tar_s = O.resize_linear(vic, 32, 32), O.resize_linear(tar, 32, 32)
    baseline = bool(seed % 3 == 1)
    Tn = int(rng.choice([7, 33, 100]))
    np.random.seed(seed)
    xf = T.get_transform_params(Tn, "GTSRB", baseline=baseline)
    tab = T.build_xform_table(xf, 244, 244, PT14 if seed % 5 else None)
    C = int(rng.choice([1, 3, 9]))
    masks = synth.synth_patch_masks(f"fuzz-{seed}", C, hull, p_keep=float(rng.choice([0.1, 0.5, 0.9])))
    noise = rng.uniform(-0.3, 0.3, size=(C, 3, 32, 32)).astype(np.float32)
    thetas = (((tar_s - vic_s)[None] + noise) * masks).astype(np.float32)
    lbd = float(rng.choice([1.0, 0.5]))
    px = _hip.warp_batch(vic, masks, thetas, tab, lbd, True)
    ref = O.warp_batch(vic, masks, thetas, tab, lbd=lbd)
    ok_px = np.array_equal(px, ref)
    lg = O.gtsrb_forward(tl, ref.reshape(-1, 3, 32, 32))
    target = int(rng.choice([14, 1, 27]))
    _, h_ref = O.argmax_hits(lg, Tn, target)
    hits, preds = engine.tr_hits(model, vic, masks, thetas, tab, target, lbd=lbd, return_preds=True)
    ok_h = np.array_equal(hits.cpu().numpy(), h_ref) and np.array_equal(preds.cpu().numpy().reshape(-1), lg.argmax(1))
    return ok_px, ok_h, (f"seed {seed:3d} vic {vic_id:2d} tar {tar_id:2d} T {Tn:3d} C {C} baseline {int(baseline)} lbd {lbd} "
                         f"pixels {'ok' if ok_px else 'DIFF'} hits {'ok' if ok_h else 'DIFF'} {hits.cpu().numpy().tolist()}")


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    bad = 0
    t0 = time.time()
    for seed in range(n_seeds):
        ok_px, ok_h, msg = run_case(seed)
        print(msg, flush=True)
        bad += (not ok_px) + (not ok_h)
    print(f"{n_seeds} cases, {bad} mismatches, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

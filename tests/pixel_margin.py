"""How much could a last-ulp difference between the oracle and real OpenCV matter?  

The blur / warp / LUT / 244 -> 32 resize stages of the oracle restate OpenCV 4.2 from its published algorithms; OpenCV
itself is not installable here, so those stages are "parity unpinned".  What CAN be measured is how sensitive the results
are to the kind of difference an unpinned restatement could hide: the same arithmetic rounded differently.  Three probes,
each a variant of the oracle (oracle/graphite_oracle.c, never the product):

  fma       the oracle compiled with -ffp-contract=fast: every a*b + c of resize / blur / warp blend is ONE rounding, as
            OpenCV's AVX2 v_muladd / v_fma filter and resize paths round
  nudge+1   every warped value moved one ulp up before the uint8 truncation of transforms.py:69 (the only discontinuous
  nudge-1   step of the pixel path), resp. one ulp down

For each probe, against the oracle proper, on the same candidates and transforms:
  u8 stage   |trunc(255 v') - trunc(255 v)| of the warped pixels: must be <= 1 level = 1/255 (the north star's pixel
             tolerance) -- asserted
  net input  the [3,32,32] images after the gamma LUT and the resize: values that differ, largest difference in levels
             (a gamma < 1 table is steep near black: lut[1] - lut[0] is 52 levels for gamma 1/3.5, so one level before
             the LUT can be many after it -- reported, not bounded)
  labels     images whose argmax changes
  hits       candidates whose hit count changes, and by how much

    python tests/pixel_margin.py [--gpu] [--cands 64] [--transforms 1000] [--out profiles/r03_pixel_margin.txt]

--gpu classifies with the HIP library (bit-identical to the oracle's classifier, tests/test_gpu_parity.py) so that
64 x 1000 x 4 images take seconds; without it the oracle's classifier runs on the host cores.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as O                                                      # noqa: E402
from cases import gtsrb_case                                                # noqa: E402
from graphite_amd import scenes, synth, weights                             # noqa: E402

PROBES = [("fma", dict(fma=True)), ("nudge+1", dict(nudge=1)), ("nudge-1", dict(nudge=-1)),
          ("fma,nudge+1", dict(fma=True, nudge=1)), ("fma,nudge-1", dict(fma=True, nudge=-1))]


def u8_stage(case, masks, thetas, tsel, probe):
    """trunc(255 * warped) of the oracle proper and of a probe, for the given candidates x transforms."""
    def run(**kw):
        out = []
        with O.variant(**{k: v for k, v in kw.items() if k == "fma"}):
            for m, th in zip(masks, thetas):
                _, pert, ml = O.add_noise(case["vic"], m, th)
                att = {}
                for t in tsel:
                    xf = case["tab"][t]
                    b = int(xf["blur"])
                    if b not in att:
                        att[b] = O.composite(case["vic"], pert, ml, b)
                    _, w = O.warp_one(att[b], xf, want_warped=True)
                    n = kw.get("nudge", 0)
                    if n:
                        w = np.nextafter(w, np.float32(np.inf if n > 0 else -np.inf), dtype=np.float32)
                    out.append((np.maximum(w, np.float32(0)) * np.float32(255.0)).astype(np.int32).ravel())
        return np.concatenate(out)
    return run(), run(**probe)


def margin(case, masks, thetas, classify, target, tag, log):
    T = len(case["tab"])
    base_px = O.warp_batch(case["vic"], masks, thetas, case["tab"])
    base_lab = classify(base_px.reshape(-1, 3, 32, 32)).argmax(1)
    base_hits = (base_lab.reshape(len(masks), T) == target).sum(1)
    log("%s: %d candidates x %d transforms = %d images; base hits of target %d: min %d / mean %.1f / max %d of %d"
        % (tag, len(masks), T, base_px.shape[0] * T, target, base_hits.min(), base_hits.mean(), base_hits.max(), T))
    rows = []
    for name, kw in PROBES:
        with O.variant(**kw):
            px = O.warp_batch(case["vic"], masks, thetas, case["tab"])
        d = np.abs(px.astype(np.float64) - base_px) * 255.0
        lab = classify(px.reshape(-1, 3, 32, 32)).argmax(1)
        hits = (lab.reshape(len(masks), T) == target).sum(1)
        row = dict(probe=name, values=int(d.size), values_moved=int((d > 0).sum()), values_moved_over_half_level=int((d > 0.5).sum()),
                   max_levels=float(d.max()), images_touched_over_half_level=int((d.reshape(-1, 3072).max(1) > 0.5).sum()),
                   images=int(lab.size), labels_flipped=int((lab != base_lab).sum()),
                   candidates=int(len(masks)), candidates_hit_change=int((hits != base_hits).sum()),
                   max_hit_change=int(np.abs(hits - base_hits).max()))
        rows.append(row)
        log("  %-12s net-input values moved %d of %d (> 0.5 level: %d, largest %.2f levels; images with such a value: %d of %d); "
            "labels flipped %d of %d; candidates whose hit count changes %d of %d (largest change %d of %d)"
            % (name, row["values_moved"], row["values"], row["values_moved_over_half_level"], row["max_levels"],
               row["images_touched_over_half_level"], row["images"], row["labels_flipped"], row["images"],
               row["candidates_hit_change"], row["candidates"], row["max_hit_change"], T))
    return rows


def u8_report(case, masks, thetas, tsel, log):
    worst = 0
    for name, kw in PROBES:
        a, b = u8_stage(case, masks, thetas, tsel, kw)
        d = np.abs(a - b)
        worst = max(worst, int(d.max()))
        log("  u8 stage %-12s warped values %d, moved by one level %d (%.4f %%), by more %d"
            % (name, d.size, int((d == 1).sum()), 100.0 * (d == 1).mean(), int((d > 1).sum())))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--cands", type=int, default=64)
    ap.add_argument("--transforms", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    def classifier(sd):
        tl = weights.gtsrb_tensor_list(sd)
        if a.gpu:
            import torch
            from graphite_amd import models
            m = models.GTSRBNet()
            m.load_state_dict(sd)
            return lambda x: np.concatenate([m.hip_model.forward(torch.from_numpy(x[i:i + 8000]).cuda()).cpu().numpy()
                                             for i in range(0, len(x), 8000)])
        return lambda x: O.gtsrb_forward(tl, x, nthreads=min(16, os.cpu_count() or 1))

    t0 = time.time()
    log("pixel margin report (tests/pixel_margin.py); classifier on %s" % ("the GPU (HIP library)" if a.gpu else "the host (oracle)"))
    for wtag, sd, target in (("seeded GTSRBNet, calibrated last bias (bench.py's weights), target 1", synth.gtsrb_state_dict(), 1),
                             ("seeded GTSRBNet + fitted 3-sign read-out (the attack's weights), target 27",
                              synth.gtsrb_state_dict(readout="toy3"), 27)):
        log("== " + wtag)
        cl = classifier(sd)
        tid = 1 if target == 1 else 27
        c1 = gtsrb_case(target_id=tid, num_xforms=100)
        m1, t1 = scenes.stop_sign_candidates(c1, 1)
        margin(c1, m1, t1, cl, target, "config 1 (hull mask, 100 transforms)", log)
        c4 = gtsrb_case(target_id=tid, num_xforms=a.transforms)
        m4 = synth.synth_patch_masks("bench-sweep", a.cands, c4["hull"])
        t4 = ((c4["tar_s"] - c4["vic_s"])[None] * m4).astype(np.float32)
        margin(c4, m4, t4, cl, target, "config 4, first %d candidates" % a.cands, log)
    log("== u8 stage (warped pixels before the LUT), config 1 hull + 3 config-4 candidates x 100 transforms")
    c1 = gtsrb_case(target_id=1, num_xforms=100)
    m, t = scenes.stop_sign_candidates(c1, 1)
    m4 = synth.synth_patch_masks("bench-sweep", 3, c1["hull"])
    mm = np.concatenate([m, m4]); tt = np.concatenate([t, ((c1["tar_s"] - c1["vic_s"])[None] * m4).astype(np.float32)])
    worst = u8_report(c1, mm, tt, range(100), log)
    log("largest u8-stage move: %d level(s)  [north star: warped pixels within 1/255]" % worst)
    log("elapsed %.0f s" % (time.time() - t0))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return 0 if worst <= 1 else 1


if __name__ == "__main__":
    sys.exit(main())

"""Bounded version of tests/pixel_margin.py (the full report is profiles/r03_pixel_margin.txt): how far the kind of
difference an UNPINNED restatement of OpenCV could hide -- fused multiply-adds, a last-ulp difference in front of the
uint8 truncation (transforms.py:69) -- moves pixels, labels and hit counts.  Oracle against oracle variants, CPU only."""
import numpy as np

import pixel_margin as PM
from cases import gtsrb_case
from graphite_amd import scenes, synth, weights


def test_probes_move_warped_pixels_by_at_most_one_level():
    c = gtsrb_case(target_id=1, num_xforms=100)
    m, t = scenes.stop_sign_candidates(c, 1)
    for name, kw in PM.PROBES:
        a, b = PM.u8_stage(c, m, t, range(0, 100, 5), kw)
        d = np.abs(a - b)
        assert d.max() <= 1, name                                  # the north star's 1/255 for warped pixels
        assert 0 < (d == 1).mean() < 0.08, (name, (d == 1).mean()) # the probes do bite (0.5 % fma / nudge+1, 4-5 % nudge-1)


def test_hit_counts_are_robust_to_the_probes_on_config_1(oracle):
    """Config 1 (hull mask, 100 transforms), both weight sets: no label moves although up to 7 % of the net-input
    values do (by up to 4 levels behind a steep gamma table)."""
    lines = []
    for sd, target in ((synth.gtsrb_state_dict(), 1), (synth.gtsrb_state_dict(readout="toy3"), 27)):
        tl = weights.gtsrb_tensor_list(sd)
        c = gtsrb_case(target_id=target, num_xforms=100)
        m, t = scenes.stop_sign_candidates(c, 1)
        rows = PM.margin(c, m, t, lambda x: oracle.gtsrb_forward(tl, x), target, "config 1", lines.append)
        for r in rows:
            assert r["values_moved"] > 0                           # the probe changed the classifier's input ...
            assert r["labels_flipped"] <= 1 and r["max_hit_change"] <= 1, r    # ... and at most one label of 100 (measured: 0)

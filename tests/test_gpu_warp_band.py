"""The opt-in LDS-band warp kernel (GRAPHITE_WARP_BAND=1, pixel.hip k_warp_band) against the oracle.  The switch is read
once per process, so the check runs in a child process; it covers a GTSRB table (15 bands of the 244x244 composite, all
four blur slots, the 2x2-area and bilinear resize branches) and a CIFAR table (one band, `same` crops), fused and CHW."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import numpy as np
import oracle_lib as O
from cases import gtsrb_case, gtsrb_candidates, cifar_case
from graphite_amd import _hip, engine, models, synth, weights
case = gtsrb_case(num_xforms=200)
masks, thetas = gtsrb_candidates(case, 3)
got = _hip.warp_batch(case["vic"], masks, thetas, case["tab"], 1.0, True)
ref = O.warp_batch(case["vic"], masks, thetas, case["tab"])
assert np.array_equal(got, ref), "GTSRB pixels differ"
sd = synth.gtsrb_state_dict()
m = models.GTSRBNet(); m.load_state_dict(sd)
lg = O.gtsrb_forward(weights.gtsrb_tensor_list(sd), ref.reshape(-1, 3, 32, 32))
hits, preds = engine.tr_hits(m, case["vic"], masks, thetas, case["tab"], 1, return_preds=True)
assert np.array_equal(preds.cpu().numpy().reshape(-1), lg.argmax(1)), "labels differ (fused NHWC4 + conv1_1 output)"
c = cifar_case(num_xforms=150)
cm = np.concatenate([c["hull"][None], synth.synth_patch_masks("band-cf", 1, c["hull"])])
ct = ((c["tar_s"] - c["vic_s"])[None] * cm).astype(np.float32)
assert np.array_equal(_hip.warp_batch(c["vic"], cm, ct, c["tab"], 1.0, False), O.warp_batch(c["vic"], cm, ct, c["tab"], normalize=False))
print("band kernel ok")
'''


def test_band_kernel_bit_exact_in_child_process():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GRAPHITE_WARP_BAND="1", PYTHONPATH=os.pathsep.join([os.path.dirname(here), here]))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "band kernel ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

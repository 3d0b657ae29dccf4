"""CPU pin of the plain-PyTorch restatements that the floating-point GPU tests lean on: their fp32 forward must agree
with the C oracle (which is pinned against the reference's own PyTorch forward, tests/test_oracle_golden.py), and the
bf16-operand variants must differ from fp32 by about the bfloat16 rounding error and no more.  No GPU."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import gtsrb_torch_reference as G                                           # noqa: E402
import wrn_torch_reference as W                                             # noqa: E402
import oracle_lib as O                                                      # noqa: E402
from graphite_amd import synth, weights                                     # noqa: E402


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_gtsrb_restatement_matches_c_oracle_and_bf16_variant_is_close():
    sd = synth.gtsrb_state_dict()
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
    x = torch.from_numpy(synth.synth_images("restate-gtsrb", 6))
    ref = O.gtsrb_forward(weights.gtsrb_tensor_list(sd), x.numpy())
    f32 = G.forward(tsd, x).numpy()
    assert _rel(f32, ref) <= 1e-4
    b16 = G.forward_bf16(tsd, x).numpy()
    assert 1e-4 < _rel(b16, ref) <= 5e-2
    # no layer selected == the fp32 forward
    assert np.array_equal(G.forward_bf16(tsd, x, layers=()).numpy(), f32)


def test_wrn_restatement_matches_c_oracle_and_bf16_variant_is_close():
    depth, widen = 10, 10                       # one block per stage, every conv shape of WRN-34-10
    sd = synth.wrn_state_dict(depth=depth, widen=widen)
    tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
    x = torch.from_numpy(synth.synth_images("restate-wrn", 3, 0.0, 1.0))
    ref = O.wrn_forward(weights.wrn_tensor_list(sd, depth, widen), depth, widen, 10, x.numpy())
    f32 = W.forward(tsd, x, depth, widen).numpy()
    assert _rel(f32, ref) <= 1e-4
    b16 = W.forward(tsd, x, depth, widen, bf16=True).numpy()
    assert 1e-5 < _rel(b16, ref) <= 5e-2

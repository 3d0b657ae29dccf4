"""a13, CPU side: the C oracle's PatchGuard restatement (oracle/graphite_oracle.c: or_pg_preprocess, or_bagnet_forward,
or_masking_defense) against plain PyTorch / numpy (oracle/patchguard_reference.py).

  preprocess       reference code (patchguard_predict.py:42-44, :30-32): torch's own F.interpolate(mode='bicubic') is the
                   oracle here -- pinned, tolerance 1e-5
  BagNet           third-party architecture restated twice (C with folded BatchNorm, torch modules): the two must agree to
                   1e-4 relative -- consistency of the restatements, PARITY UNPINNED against the absent PatchGuard source
  masking_defense  third-party, restated twice (C, numpy): same predictions, same masked evidence -- PARITY UNPINNED"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import patchguard_reference as PR                           # noqa: E402
from graphite_amd import synth, weights                     # noqa: E402

SMALL = dict(layers=(1, 1, 1, 1), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0))
B33 = dict(layers=(2, 2, 2, 2), strides=(2, 2, 1, 1), kernel3=(1, 1, 1, 1))       # bagnet33-style: every stage has a 3x3


def _torch_net(sd, arch):
    net = PR.BagNet(arch["layers"], arch["strides"], arch["kernel3"]).eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net


@pytest.mark.parametrize("size", [192, 64, 50])
def test_preprocess_equals_the_reference_lines(oracle, size):
    x = synth.synth_images("pg-pre", 3)
    x[0, :, :2, :] = -0.5; x[0, :, -1, :] = 0.5              # flat borders: the clamped taps
    got = oracle.pg_preprocess(x, size)
    ref = PR.preprocess(torch.from_numpy(x), size).numpy()
    assert got.shape == (3, 3, size, size)
    assert np.abs(got - ref).max() <= 1e-5, np.abs(got - ref).max()        # measured 7e-6 on values up to 3.7
    # bicubic overshoot exists (A = -0.75 has negative lobes): not a clamped / bilinear operator
    lo = (0.0 - 0.4914) / 0.2023
    assert got[:, 0].min() < lo - 1e-3


@pytest.mark.parametrize("arch,size", [(SMALL, 64), (B33, 48), (synth.BAGNET17, 192)])
def test_bagnet_restatements_agree(oracle, arch, size):
    sd = synth.bagnet_state_dict("bagnet-t", arch["layers"], arch["kernel3"], calibrated=False)
    x = synth.synth_images("pg-net", 2)
    z = oracle.pg_preprocess(x, size)
    lo = oracle.bagnet_forward(weights.bagnet_tensor_list(sd, arch["layers"]), z, **arch)
    with torch.no_grad():
        ref = _torch_net(sd, arch)(torch.from_numpy(z)).numpy()
    assert lo.shape == ref.shape and lo.shape[1] == oracle.bagnet_out_size(size, **arch)
    assert np.abs(lo - ref).max() <= 1e-4 * np.abs(ref).max()
    if arch is synth.BAGNET17:
        assert lo.shape == (2, 22, 22, 10)                   # rf 17, stride 8 at 192 (patchguard_predict.py:33-34)


def test_masking_defense_restatements_agree(oracle):
    rng = np.random.RandomState(5)
    local = rng.randn(40, 22, 22, 10).astype(np.float32) * 3
    local[0] = -np.abs(local[0])                              # no positive evidence at all: every class 0 -> last index wins
    local[1, 3:9, 4:10, 7] += 40.0                            # a "patch": huge class-7 evidence inside one 6x6 window
    local[2] = 0.0
    local[3, :, :, 4] = local[3, :, :, 2]                     # an exact tie between classes 2 and 4
    scores, preds = oracle.masking_defense(local, 6, 0.0)
    for i in range(len(local)):
        assert preds[i] == PR.masking_defense(local[i], 0.0, (6, 6)), i
    assert preds[0] == 9 and preds[2] == 9 and np.all(scores[0] == 0)
    # the patch is masked out: class 7 would win on the plain clipped sum, it does not after masking
    plain = np.clip(local[1], 0, None).sum((0, 1))
    assert plain.argmax() == 7 and preds[1] != 7
    assert scores[3, 4] == scores[3, 2]
    # masked evidence against float64 numpy
    u = np.clip(local.astype(np.float64), 0, None)
    tot = u.sum((1, 2))
    win = np.max([[u[:, a:a + 6, b:b + 6].sum((1, 2)) for b in range(17)] for a in range(17)], axis=(0, 1))
    assert np.allclose(scores, tot - win, rtol=1e-5, atol=1e-3)
    # a threshold of 1.0 can never be exceeded: nothing is masked
    s1, _ = oracle.masking_defense(local, 6, 1.0)
    assert np.allclose(s1, tot, rtol=1e-5, atol=1e-3)


def test_whole_patchguard_prediction_of_the_two_restatements(oracle):
    sd = synth.bagnet_state_dict()
    x = synth.synth_images("pg-e2e", 3)
    lo = oracle.bagnet_forward(weights.bagnet_tensor_list(sd), oracle.pg_preprocess(x))
    _, preds = oracle.masking_defense(lo, PR.WINDOW_SIZE, 0.0)
    ref = PR.PatchGuard(_torch_net(sd, synth.BAGNET17))
    with torch.no_grad():
        assert preds.tolist() == ref.predict(torch.from_numpy(x)).tolist()
    nd = PR.PatchGuard(PR.BagNet(aggregation='mean').eval())
    nd.net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    with torch.no_grad():
        assert oracle.pg_mean_logits(lo).argmax(1).tolist() == nd.predict(torch.from_numpy(x)).tolist()


def test_host_class_needs_weights_and_never_falls_back():
    from graphite_amd.patchguard.patchguard_predict import PatchGuard, PatchGuardNoDefense, WINDOW_SIZE
    assert WINDOW_SIZE == 6
    with pytest.raises(AttributeError):
        PatchGuard().hip_model
    assert not hasattr(PatchGuard(net=lambda z: z), "hip_model")      # a duck-typed middle never takes the fused path
    assert PatchGuardNoDefense._defense is False

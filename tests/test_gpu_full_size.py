"""BASELINE configs at their stated sizes on the GPU.

config 4  512 x 1000 and 4096 x 1000 on GTSRBNet: size-independent properties (slice additivity over 8 ranks' slices,
          permutation equivariance over candidates, every image gets exactly one label, hits == count of that label)
          plus the oracle on the first 64 candidates.
config 2  the whole stop -> pedestrians attack at -m 100 -b 100 (binary coarse reduction, seed 42): GPU trajectory ==
          oracle trajectory; the same in the 'Victim' heatmap mode.
config 3  tr_hits through WideResNet-34-10 against the oracle (C = 2, T = 8).
fuzz      8 randomised cases of tests/fuzz_parity.py.
"""
import os
import random

import numpy as np
import pytest
import torch

import oracle_lib as O
from cases import INP, gtsrb_case, cifar_case
from graphite_amd import _hip, engine, main as gmain, models, parsearguments, synth, weights
from oracle_evaluator import OracleEvaluator

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gtsrb_model():
    m = models.GTSRBNet()
    m.load_state_dict(synth.gtsrb_state_dict())
    return m


@pytest.mark.parametrize("C", [512, 4096])
def test_config4_at_full_size(gtsrb_model, C):
    T = 1000
    case = gtsrb_case(num_xforms=T)
    masks = synth.synth_patch_masks("bench-sweep", C, case["hull"])          # bench.py's candidates
    thetas = ((case["tar_s"] - case["vic_s"])[None] * masks).astype(np.float32)
    img, md, td = _hip.to_dev(case["vic"]), _hip.to_dev(masks), _hip.to_dev(thetas)
    xf = _hip.xform_table(case["tab"])
    hm = gtsrb_model.hip_model
    target = 1
    _hip.set_chunk(32000)
    try:
        hits, preds, _ = _hip.tr_hits_dev(hm, img, md, td, xf, 1.0, True, target, want_preds=True)
        h = hits.cpu().numpy()
        p = preds.cpu().numpy().reshape(C, T)
        assert ((p >= 0) & (p < 43)).all()                                    # every image got exactly one label
        assert np.array_equal((p == target).sum(1), h)
        assert 0 < h.sum() < C * T and len(np.unique(h)) > 10                 # not a constant classifier
        # the oracle on the first 64 candidates (what bench.py reports as parity_vs_oracle_on_sample)
        px = O.warp_batch(case["vic"], masks[:64], thetas[:64], case["tab"], nthreads=16)
        lg = O.gtsrb_forward(weights.gtsrb_tensor_list(synth.gtsrb_state_dict()), px.reshape(-1, 3, 32, 32), nthreads=16)
        pr, hr = O.argmax_hits(lg, T, target)
        assert np.array_equal(p[:64].reshape(-1), pr) and np.array_equal(h[:64], hr)
        # slices of 8 ranks sum to the whole
        acc = torch.zeros(C, dtype=torch.int64, device=hits.device)
        for r in range(8):
            b, c = engine.shard_range(C * T, r, 8)
            acc += _hip.tr_hits_dev(hm, img, md, td, xf, 1.0, True, target, b, c)[0]
        assert torch.equal(acc, hits.long())
        # permutation of the candidates permutes the counts
        perm = torch.from_numpy(np.random.RandomState(0).permutation(C)).to(hits.device)
        h2 = _hip.tr_hits_dev(hm, img, md[perm].contiguous(), td[perm].contiguous(), xf, 1.0, True, target)[0]
        assert torch.equal(hits[perm], h2)
    finally:
        _hip.set_chunk(8192)


def _attack(evaluator, T, target, sd, heatmap="Target"):
    args = parsearguments.getarguments(["-v", "14", "-t", str(target), "--tr_lo", "0.65", "--tr_hi", "0.85", "-m", str(T),
                                        "-b", str(T), "--coarse_mode", "binary", "--seed", "42", "--heatmap", heatmap])
    model = models.GTSRBNet()
    model.load_state_dict(sd)
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    return gmain.attack_network(
        model, os.path.join(INP, "GTSRB/images/14.png"), os.path.join(INP, f"GTSRB/images/{target}.png"),
        os.path.join(INP, "GTSRB/Hulls/14.png"), 14, target, os.path.join(INP, "GTSRB/Points/14.csv"), "score.py",
        heatmap, args.coarse_error, args.reduce_error, 1, T, T, model_type='GTSRB', args=args, write_outputs=False,
        _evaluator=evaluator)


def _same_trajectory(gpu, cpu):
    assert gpu["queries"] == cpu["queries"] and gpu["pixels"] == cpu["pixels"]
    assert gpu["mask_queries"] == cpu["mask_queries"] and gpu["boost_queries"] == cpu["boost_queries"]
    assert gpu["transform_robustness"] == cpu["transform_robustness"]
    for k in ("mask", "theta", "adversarial", "perturb"):
        assert torch.equal(gpu[k], cpu[k]), k


def test_config2_attack_m100_b100_matches_oracle():
    """BASELINE config 2 as stated: stop -> pedestrians, -m 100 -b 100, binary coarse reduction, seed 42."""
    sd = synth.gtsrb_state_dict(readout="toy3")
    gpu = _attack(None, 100, 27, sd)
    cpu = _attack(OracleEvaluator.factory(sd), 100, 27, sd)
    assert 0 < gpu["pixels"] < 702 and gpu["queries"] > 100000
    _same_trajectory(gpu, cpu)


def test_victim_heatmap_attack_matches_oracle():
    """--heatmap Victim: the heatmap is surveyed on the TARGET image with the labels swapped (generate_mask.py:262-263)."""
    sd = synth.gtsrb_state_dict(readout="toy3")
    gpu = _attack(None, 20, 27, sd, heatmap="Victim")
    cpu = _attack(OracleEvaluator.factory(sd), 20, 27, sd, heatmap="Victim")
    assert gpu["queries"] > 1000
    _same_trajectory(gpu, cpu)


def test_tr_hits_wrn34_10_vs_oracle():
    """Config 3's classifier at full size: C = 2 candidates x T = 8 CIFAR transforms through WideResNet-34-10."""
    c = cifar_case(num_xforms=8)
    sd = synth.wrn_state_dict()
    m = models.WideResNet()
    m.load_state_dict(sd)
    masks = np.concatenate([c["hull"][None], synth.synth_patch_masks("wrn34", 1, c["hull"])])
    thetas = ((c["tar_s"] - c["vic_s"])[None] * masks).astype(np.float32)
    imgs = O.warp_batch(c["vic"], masks, thetas, c["tab"], normalize=False).reshape(-1, 3, 32, 32)
    lg = O.wrn_forward(weights.wrn_tensor_list(sd), 34, 10, 10, imgs, nthreads=16)
    target = int(np.bincount(lg.argmax(1)).argmax())
    _, hits_ref = O.argmax_hits(lg, 8, target)
    hits, preds, logits = _hip.tr_hits_dev(m.hip_model, c["vic"], masks, thetas, c["tab"], 1.0, False, target,
                                           want_preds=True, want_logits=True)
    assert np.array_equal(logits.cpu().numpy(), lg), f"max |hip-oracle| = {np.abs(logits.cpu().numpy() - lg).max()}"
    assert np.array_equal(preds.cpu().numpy(), lg.argmax(1))
    assert np.array_equal(hits.cpu().numpy(), hits_ref)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_parity_bounded(seed):
    import fuzz_parity
    ok_px, ok_h, msg = fuzz_parity.run_case(seed)
    assert ok_px and ok_h, msg

"""Whole-attack parity on the GPU: generate_mask -> boost with the HIP evaluator must follow exactly the
trajectory the CPU oracle dictates (same masks, thetas, transform robustness, query counts), because
every hit count along the way is bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

from cases import INP
from graphite_amd import main as gmain
from graphite_amd import models, parsearguments, synth
from oracle_evaluator import OracleEvaluator

pytestmark = pytest.mark.gpu


def _run(evaluator, T, target, sd):
    args = parsearguments.getarguments(["-v", "14", "-t", str(target), "--tr_lo", "0.65", "--tr_hi", "0.85"])
    model = models.GTSRBNet()
    model.load_state_dict(sd)
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    return gmain.attack_network(
        model, os.path.join(INP, "GTSRB/images/14.png"), os.path.join(INP, f"GTSRB/images/{target}.png"),
        os.path.join(INP, "GTSRB/Hulls/14.png"), 14, target, os.path.join(INP, "GTSRB/Points/14.csv"), "score.py",
        "Target", args.coarse_error, args.reduce_error, 1, T, T, model_type='GTSRB', args=args, write_outputs=False,
        _evaluator=evaluator)


def test_attack_trajectory_matches_oracle():
    # stop sign -> pedestrians with the fitted 3-sign read-out: the coarse / fine reductions and the boosting take
    # real accept / reject decisions, every one of which hinges on a hit count
    T = 20
    target = 27
    sd = synth.gtsrb_state_dict(readout="toy3")
    gpu = _run(None, T, target, sd)
    cpu = _run(OracleEvaluator.factory(sd), T, target, sd)
    assert 0 < gpu["pixels"] < 702 and gpu["queries"] > 1000          # the mask was really reduced
    assert gpu["queries"] == cpu["queries"] and gpu["pixels"] == cpu["pixels"]
    assert gpu["transform_robustness"] == cpu["transform_robustness"]
    assert torch.equal(gpu["mask"], cpu["mask"])
    assert torch.equal(gpu["theta"], cpu["theta"])
    assert torch.equal(gpu["adversarial"], cpu["adversarial"])
    assert torch.equal(gpu["perturb"], cpu["perturb"])

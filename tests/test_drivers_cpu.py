"""Search drivers (boost / generate_mask / coarse_reduction) vs the REFERENCE'S OWN drivers.

tests/golden/drivers.json was produced by running the reference's generate_mask.py / boost.py /
coarse_reduction.py with the transform+classify pair replaced by tests/fake_tr.FakeTR (see
make_golden.py::golden_drivers).  Here our batched / speculative drivers run with the same fake
evaluator; decisions, masks, thetas, reported query counts and RNG consumption must be identical,
while the number of LAUNCHES must be far smaller (that is the point of batching).
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from cases import GOLD, INP
from fake_tr import FakeEvaluator, FakeTR
from graphite_amd import boost as gboost
from graphite_amd import coarse_reduction as gcoarse
from graphite_amd import generate_mask as ggm
from graphite_amd import parsearguments
from graphite_amd.imageio import load_image_chw

ROWS = json.load(open(os.path.join(GOLD, "drivers.json")))
PT = os.path.join(INP, "GTSRB/Points/14.csv")


def _digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def scene(oracle):
    vic = torch.from_numpy(load_image_chw(os.path.join(INP, "GTSRB/images/14.png")))
    tar = torch.from_numpy(load_image_chw(os.path.join(INP, "GTSRB/images/1.png")))
    hull = torch.from_numpy((load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32))
    vic_s = torch.from_numpy(oracle.resize_linear(vic.numpy(), 32, 32))
    tar_s = torch.from_numpy(oracle.resize_linear(tar.numpy(), 32, 32))
    return dict(vic=vic, tar=tar, hull=hull, vic_s=vic_s, tar_s=tar_s, obj=float(hull.sum() / 3),
                ref_dir=(tar_s - vic_s) * hull)


@pytest.mark.parametrize("speculate", [True, False], ids=["speculative", "sequential"])
@pytest.mark.parametrize("row", [r for r in ROWS if r["kind"] == "generate_mask"], ids=lambda r: r["name"])
def test_generate_mask_matches_reference_driver(row, scene, oracle, speculate):
    fake = FakeTR(row["T"], scene["obj"], scene["ref_dir"], row["sharp"], row["mid"])
    ev = FakeEvaluator(fake, oracle, speculate)
    args = parsearguments.getarguments(["--coarse_mode", row["coarse_mode"]])
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    attacked, qc, best_mask, nbits, best_tr = ggm.generate_mask(
        None, scene["vic_s"], scene["vic"], 14, scene["hull"].clone(), scene["tar_s"], scene["tar"], 1, PT, "score.py",
        row["heatmap"], 1 - row["tr_hi"], 1 - row["tr_lo"], num_xforms=row["T"], net_size=32, model_type='GTSRB',
        patch_size=4, max_mask_size=row["max_mask_size"], _evaluator=ev, _args=args)
    assert int(qc) == row["query_count"]
    assert float(nbits) == row["nbits"] and float(best_tr) == row["best_tr"]
    assert best_mask.numpy()[0].astype(np.uint8).tolist() == row["best_mask"]
    assert torch.equal(best_mask[0], best_mask[1]) and torch.equal(best_mask[0], best_mask[2])
    assert _digest(attacked) == row["attacked_sha"]
    assert float(np.random.uniform()) == row["np_random_next"] and random.random() == row["py_random_next"]
    # batching: the reference made `fake_calls` one-candidate round trips; we made far fewer launches,
    # and speculation evaluated at most a bounded number of extra candidates
    if speculate:
        assert ev.launches < row["fake_calls"] / 3
        assert ev.evaluated >= row["fake_calls"]
    else:           # a duck-typed classifier sees exactly the reference's evaluations (no hidden queries)
        assert ev.evaluated == row["fake_calls"]
    print(row["name"], "reference evals", row["fake_calls"], "our launches", ev.launches, "our evals", ev.evaluated)


@pytest.mark.parametrize("speculate", [True, False], ids=["fused", "sequential"])
@pytest.mark.parametrize("row", [r for r in ROWS if r["kind"] == "boost"], ids=lambda r: r["name"])
def test_boost_matches_reference_driver(row, scene, oracle, speculate):
    fake = FakeTR(row["T"], scene["obj"], scene["ref_dir"], 6.0, 0.1)
    ev = FakeEvaluator(fake, oracle, speculate)
    small_mask = torch.zeros(3, 32, 32)
    small_mask[:, 10:22, 8:24] = 1
    small_mask = small_mask * scene["hull"]
    init_theta = (scene["tar_s"] - scene["vic_s"]) * 0.5 if row["use_init"] else None
    np.random.seed(42); torch.manual_seed(42)
    adv, tr, perturb, queries, best_theta, adv_small = gboost.boost(
        None, scene["vic"], 14, small_mask, scene["tar_s"], 1, beta=1, iterations=row["iterations"], pt_file=PT,
        num_xforms=row["T"], net_size=32, model_type=row["model_type"], bt=row["bt"], budget_factor=row["budget_factor"],
        init_theta=init_theta, _evaluator=ev)
    assert float(tr) == row["tr"] and int(queries) == row["queries"]
    assert _digest(best_theta) == row["best_theta_sha"]
    assert _digest(adv) == row["adv_sha"] and _digest(perturb) == row["perturb_sha"]
    assert _digest(adv_small) == row["adv_small_sha"]
    assert float(torch.rand(1)) == row["torch_random_next"] and float(np.random.uniform()) == row["np_random_next"]
    assert ev.evaluated == row["fake_calls"]              # nothing is wasted in boost: exactly the reference's evaluations
    assert ev.launches < row["fake_calls"]                # 10 directions per launch
    if speculate and not row["bt"]:                       # ... and the step rides with the next directions: 11 per launch
        assert ev.launches <= 3 + (row["fake_calls"] - 1) // 11


def test_binary_search_replay_table(oracle):
    """perform_binary_coarse_reduction vs the reference on synthetic monotone / non-monotone TR tables."""
    rows = json.load(open(os.path.join(GOLD, "binary_search.json")))
    for r in rows:
        n, table = r["n"], r["table"]
        patches = []
        for i in range(n):
            p = torch.zeros(3, 32, 32)
            p[:, (i // 8) * 4:(i // 8) * 4 + 4, (i % 8) * 4:(i % 8) * 4 + 4] = 1
            patches.append(p)
        calls = []

        class Ev:
            def error(self, mask, theta):
                piv = int(round(float(mask.sum()) / 3 / 16))
                calls.append(piv)
                return 1.0 - table[piv], 100

        z = torch.zeros(3, 32, 32)
        pivot, best_mask, best_tr, q = gcoarse.perform_binary_coarse_reduction(
            patches, r["err_threshold"], 14, 1, z, z, torch.zeros(3, 244, 244), None, [None] * 100, torch.ones(3, 32, 32),
            None, 32, _evaluator=Ev())
        assert (int(pivot), float(best_tr), int(q), float(best_mask.sum() / 3)) == (r["pivot"], r["best_tr"], r["queries"],
                                                                                    r["mask_bits"])
        assert calls == r["calls"]


def test_parsearguments_defaults_and_derived():
    a = parsearguments.getarguments([])
    assert (a.victim_id, a.target_id, a.coarse_mode, a.heatmap, a.seed) == ('14', '1', 'binary', 'Target', 42)
    assert (a.boost_transforms, a.mask_transforms, a.lbl_v, a.lbl_t) == (100, 100, 14, 1)
    assert a.coarse_error == 0.5 and a.reduce_error == 0.75
    a = parsearguments.getarguments("-v 14 -t 27 --tr_lo 0.65 --tr_hi 0.85 -b 100 -m 100 --coarse_mode=binary --junk 1".split())
    assert a.lbl_t == 27 and abs(a.coarse_error - 0.15) < 1e-12 and abs(a.reduce_error - 0.35) < 1e-12
    assert a.img_v == 'inputs/GTSRB/images/14.png' and a.mask == 'inputs/GTSRB/Hulls/14.png'


def test_heatmap_cache_round_trip_and_restricted_loader(scene, oracle, tmp_path):
    """A heatmap cache written through heatmap_out_path (the reference's pickle layout, generate_mask.py:72-79) reloads
    through heatmap_file with identical downstream decisions and no heatmap evaluations; the loader resolves only the
    constructors such a file needs, so a pickle that names anything else is refused instead of executed."""
    import io
    import pickle
    row = [r for r in ROWS if r["kind"] == "generate_mask"][0]
    args = parsearguments.getarguments(["--coarse_mode", row["coarse_mode"]])

    def run(**kw):
        fake = FakeTR(row["T"], scene["obj"], scene["ref_dir"], row["sharp"], row["mid"])
        ev = FakeEvaluator(fake, oracle)
        np.random.seed(42); torch.manual_seed(42); random.seed(42)
        out = ggm.generate_mask(None, scene["vic_s"], scene["vic"], 14, scene["hull"].clone(), scene["tar_s"], scene["tar"], 1, PT,
                                "score.py", row["heatmap"], 1 - row["tr_hi"], 1 - row["tr_lo"], num_xforms=row["T"], net_size=32,
                                model_type='GTSRB', patch_size=4, max_mask_size=row["max_mask_size"], _evaluator=ev, _args=args, **kw)
        return out, ev

    path = str(tmp_path / "heatmap.pkl")
    (att1, q1, mask1, nbits1, tr1), ev1 = run(heatmap_out_path=path)
    with open(path, "rb") as f:
        plain = pickle.load(f)                              # our own file: the reference's keys and types
    assert set(plain) == {"heatmap_query_ct", "patches", "indices", "order"} and isinstance(plain["patches"][0], torch.Tensor)
    (att2, q2, mask2, nbits2, tr2), ev2 = run(heatmap_file=path)
    assert int(q1) == int(q2) == row["query_count"] and nbits1 == nbits2 and tr1 == tr2
    assert torch.equal(mask1, mask2) and torch.equal(att1, att2)
    assert ev2.evaluated == ev1.evaluated - len(plain["patches"])          # the heatmap itself was not re-evaluated

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    for payload in (pickle.dumps({"heatmap_query_ct": 0, "patches": [Evil()], "indices": [], "order": []}),
                    pickle.dumps(Evil())):
        bad = tmp_path / "evil.pkl"
        bad.write_bytes(payload)
        with pytest.raises(pickle.UnpicklingError, match="never contains"):
            ggm.load_heatmap(str(bad))
    # a tensor whose storage bytes are themselves a hostile pickle: the storage loader is weights-only
    inner = io.BytesIO()
    pickle.dump(Evil(), inner)
    with pytest.raises(Exception):
        ggm._storage_from_bytes(inner.getvalue())
    assert not (tmp_path / "pwned").exists()
    (tmp_path / "notaheatmap.pkl").write_bytes(pickle.dumps({"a": 1}))
    with pytest.raises(ValueError, match="not a heatmap"):
        ggm.load_heatmap(str(tmp_path / "notaheatmap.pkl"))


# ---- the PatchGuard twins (patchguard/pg_generate_mask.py, pg_boost.py) against the reference's own, same method ----
PG_ROWS = json.load(open(os.path.join(GOLD, "drivers_pg.json")))


@pytest.fixture(scope="module")
def pg_scene():
    from graphite_amd import synth
    vic = torch.from_numpy(synth.synth_images("pg-victim", 1, 0.0, 1.0)[0])
    tar = torch.from_numpy(synth.synth_images("pg-target", 1, 0.0, 1.0)[0])
    mask = torch.from_numpy((load_image_chw(os.path.join(INP, "cifar/masks/mask.png")) * 255 > 128).astype(np.float32))
    return dict(vic=vic, tar=tar, mask=mask, obj=float(mask.sum() / 3), ref_dir=(tar - vic) * mask)


@pytest.mark.parametrize("speculate", [True, False], ids=["speculative", "sequential"])
@pytest.mark.parametrize("row", [r for r in PG_ROWS if r["kind"] == "generate_mask"], ids=lambda r: r["name"])
def test_pg_generate_mask_matches_reference_driver(row, pg_scene, oracle, speculate):
    from graphite_amd.patchguard import pg_generate_mask
    sc = pg_scene
    fake = FakeTR(row["T"], sc["obj"], sc["ref_dir"], row["sharp"], row["mid"])
    ev = FakeEvaluator(fake, oracle, speculate)
    args = parsearguments.getarguments(["--coarse_mode", row["coarse_mode"]])
    np.random.seed(42); torch.manual_seed(42); random.seed(42)
    attacked, qc, best_mask, nbits, best_tr = pg_generate_mask.generate_mask(
        None, sc["vic"], sc["vic"], 3, sc["mask"].clone(), sc["tar"], sc["tar"], 5, None, "score.py", row["heatmap"],
        1 - row["tr_hi"], 1 - row["tr_lo"], num_xforms=row["T"], net_size=32, model_type='CIFAR', patch_size=4,
        max_mask_size=row["max_mask_size"], _evaluator=ev, _args=args)
    assert int(qc) == row["query_count"]
    assert float(nbits) == row["nbits"] and float(best_tr) == row["best_tr"]
    assert best_mask.numpy()[0].astype(np.uint8).tolist() == row["best_mask"]
    assert _digest(attacked) == row["attacked_sha"]
    # the baseline pg parameter table was drawn (same np.random consumption as the reference's driver)
    assert float(np.random.uniform()) == row["np_random_next"] and random.random() == row["py_random_next"]
    if speculate:
        assert ev.launches < row["fake_calls"] / 3 and ev.evaluated >= row["fake_calls"]
    else:
        assert ev.evaluated == row["fake_calls"]


@pytest.mark.parametrize("speculate", [True, False], ids=["fused", "sequential"])
@pytest.mark.parametrize("row", [r for r in PG_ROWS if r["kind"] == "boost"], ids=lambda r: r["name"])
def test_pg_boost_matches_reference_driver(row, pg_scene, oracle, speculate):
    from graphite_amd.patchguard import pg_boost
    sc = pg_scene
    fake = FakeTR(row["T"], sc["obj"], sc["ref_dir"], 6.0, 0.1)
    ev = FakeEvaluator(fake, oracle, speculate)
    small_mask = torch.zeros(3, 32, 32)
    small_mask[:, 8:24, 6:26] = 1
    np.random.seed(42); torch.manual_seed(42)
    out = pg_boost.boost(None, sc["vic"], 3, small_mask, sc["tar"], 5, beta=1, iterations=row["iterations"], pt_file=None,
                         num_xforms=row["T"], net_size=32, model_type='CIFAR', bt=row["bt"],
                         budget_factor=row["budget_factor"], _evaluator=ev)
    assert len(out) == 5                                   # pg_boost.py:215-216: no best_theta in the CIFAR tuple
    adv, tr, perturb, queries, adv_small = out
    assert float(tr) == row["tr"] and int(queries) == row["queries"]
    assert _digest(adv) == row["adv_sha"] and _digest(perturb) == row["perturb_sha"]
    assert _digest(adv_small) == row["adv_small_sha"]
    assert float(torch.rand(1)) == row["torch_random_next"] and float(np.random.uniform()) == row["np_random_next"]
    assert ev.evaluated == row["fake_calls"] and ev.launches < row["fake_calls"]


def test_pg_transforms_interface():
    """pg_transforms.py's names, defaults and parameter table (the table itself is pinned by test_host_golden)."""
    import inspect
    from graphite_amd.patchguard import pg_transforms as P, patchguard_predict
    assert list(inspect.signature(P.get_transform_params).parameters) == ["num_xforms", "model_type", "nps", "baseline"]
    assert inspect.signature(P.get_transform_params).parameters["model_type"].default == 'CIFAR'
    assert list(inspect.signature(P.get_transformed_images).parameters) == ["image", "mask", "transforms", "lbd", "theta",
                                                                           "pt_file", "net_size"]
    assert list(inspect.signature(P.convert2Network).parameters) == ["img", "is_torch", "net_size"]
    np.random.seed(3)
    a = P.get_transform_params(5, baseline=True)
    np.random.seed(3)
    from graphite_amd import transforms as T
    assert a == T.get_transform_params(5, 'CIFAR', baseline=True, patchguard=True) == [tuple(t) for t in a]
    assert all(t[1] == 3.0 and t[2] == 1.0 and t[3] in (0, 3) and abs(t[0]) <= 10 for t in a)
    with pytest.raises(AssertionError):
        P.get_transform_params(1, model_type='GTSRB')
    assert patchguard_predict.WINDOW_SIZE == 6
    with pytest.raises(AttributeError, match="BagNet"):      # no weights, no fused path -- and no fallback
        patchguard_predict.PatchGuard().hip_model

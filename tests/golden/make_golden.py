#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference does not exist on the GPU box):

    cd /root/repo && python3 -B tests/golden/make_golden.py

What runs here is the reference's own Python (transforms.get_transform_params,
perspective_transform.get_perspective_transform / get_offset_and_crop_size,
score.score_fn, GTSRB.GTSRBNet, cifar.wideresnet, utils.run_predictions*,
coarse_reduction.perform_binary_coarse_reduction).  The reference imports cv2,
torchvision and kornia at module top; none is installed, so IMPORT-ONLY
placeholders are put in sys.modules: they make `import cv2` succeed and nothing
else.  The only callable among them is a recorder for `cv2.warpPerspective`
that stores the matrix and dsize the reference computed and returns zeros --
no OpenCV arithmetic is imitated, which is why pixel-level parity with OpenCV
stays "unpinned" (see DESIGN.md).

Outputs are data only (numbers); no reference source text is stored.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np
import torch

# ----------------------------------------------------------------- placeholders
_rec = {}


def _record_warp(img, M, dsize, borderMode=None, **kw):
    _rec["M"] = np.array(M, dtype=np.float64).copy()
    _rec["dsize"] = tuple(int(v) for v in dsize)
    _rec["border"] = borderMode
    return np.zeros((dsize[1], dsize[0], 3), np.float32)


def _absent(name):
    def f(*a, **k):
        raise RuntimeError(f"{name} is not available (import-only placeholder)")
    return f


cv2 = types.ModuleType("cv2")
cv2.BORDER_REPLICATE = 1
cv2.warpPerspective = _record_warp
for n in ("resize", "GaussianBlur", "LUT", "imread", "imwrite", "cvtColor"):
    setattr(cv2, n, _absent("cv2." + n))
sys.modules["cv2"] = cv2

tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvt.Normalize = _absent("torchvision.transforms.Normalize")
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt

k = types.ModuleType("kornia")
kg = types.ModuleType("kornia.geometry")
kgt = types.ModuleType("kornia.geometry.transform")
kgt.resize = _absent("kornia resize")
kgt.warp_perspective = _absent("kornia warp_perspective")
ke = types.ModuleType("kornia.enhance")
ke.adjust_gamma = _absent("kornia adjust_gamma")
k.geometry = kg
kg.transform = kgt
k.enhance = ke
for name, mod in (("kornia", k), ("kornia.geometry", kg), ("kornia.geometry.transform", kgt),
                  ("kornia.enhance", ke)):
    sys.modules[name] = mod

# ----------------------------------------------------------------- reference
import transforms as ref_transforms                      # noqa: E402
import perspective_transform as ref_pt                   # noqa: E402
import score as ref_score                                # noqa: E402
import score_border as ref_score_border                  # noqa: E402
import utils as ref_utils                                # noqa: E402
import coarse_reduction as ref_coarse                    # noqa: E402
from GTSRB.GTSRBNet import GTSRBNet                      # noqa: E402
from cifar.wideresnet import WideResNet                  # noqa: E402

sys.path.insert(0, os.path.join(REF, "patchguard"))
import pg_transforms as ref_pg                           # noqa: E402

from graphite_amd import synth                           # noqa: E402

INP = os.path.join(os.path.dirname(os.path.dirname(HERE)), "graphite_amd", "data", "inputs")


def xf_array(xf):
    return np.array([[float(v) for v in t[:9]] for t in xf], dtype=np.float64)


# ----------------------------------------------------------------- (1) transform parameter streams
def golden_xforms():
    out = {}
    for name, fn in (
        ("gtsrb_1000", lambda: ref_transforms.get_transform_params(1000, "GTSRB")),
        ("gtsrb_baseline_100", lambda: ref_transforms.get_transform_params(100, "GTSRB", baseline=True)),
        ("cifar_1000", lambda: ref_transforms.get_transform_params(1000, "CIFAR")),
        ("alpr_100", lambda: ref_transforms.get_transform_params(100, "OpenALPRBorder")),
        ("pg_100", lambda: ref_pg.get_transform_params(100, "CIFAR")),
        ("pg_baseline_100", lambda: ref_pg.get_transform_params(100, "CIFAR", baseline=True)),
    ):
        np.random.seed(42)
        out[name] = xf_array(fn())
    # a second seed, and state continuation (two consecutive calls share the global stream)
    np.random.seed(7)
    a = ref_transforms.get_transform_params(10, "GTSRB")
    b = ref_transforms.get_transform_params(10, "CIFAR")
    out["seed7_gtsrb10_then_cifar10"] = np.concatenate([xf_array(a), xf_array(b)])
    np.savez_compressed(os.path.join(HERE, "xforms.npz"), **out)
    return out


# ----------------------------------------------------------------- (2) homography + crop
def golden_homography(xf):
    out = {}
    dummy244 = np.zeros((244, 244, 3), np.float32)
    dummy32 = np.zeros((32, 32, 3), np.float32)
    cases = (
        ("gtsrb14", xf["gtsrb_1000"], 244, os.path.join(INP, "GTSRB/Points/14.csv"), dummy244),
        ("gtsrb1", xf["gtsrb_1000"][:100], 244, os.path.join(INP, "GTSRB/Points/1.csv"), dummy244),
        ("cifar", xf["cifar_1000"], 32, None, dummy32),
        ("pg", xf["pg_100"], 32, None, dummy32),
        ("gtsrb14_baseline", xf["gtsrb_baseline_100"], 244, os.path.join(INP, "GTSRB/Points/14.csv"), dummy244),
    )
    for name, params, w, pt_file, img in cases:
        Ms, crops = [], []
        for p in params:
            angle, dist, gamma, blur, cp, cox, coy, obj_w, focal = p
            d = ref_transforms.dist2pixels(dist, w, obj_w)
            f = ref_transforms.dist2pixels(focal, w, obj_w)
            ref_pt.get_perspective_transform(img, angle, w, w, f, d, cp, cox, coy, pt_file)
            Ms.append(_rec["M"].reshape(9))
            crops.append(_rec["dsize"])
            assert _rec["border"] == cv2.BORDER_REPLICATE
        out[name + "_M"] = np.array(Ms)
        out[name + "_dsize"] = np.array(crops, dtype=np.int32)
    # direct get_offset_and_crop_size on a non-square frame (tall and wide branches)
    extra = []
    for (w, h) in ((244, 244), (300, 200), (200, 300)):
        for angle in (-40.0, 0.0, 25.0):
            a = np.radians(angle)
            f, d = 3 * 12 * w / 30.0, 7.5 * 12 * w / 30.0
            for pt_file in (os.path.join(INP, "GTSRB/Points/14.csv"), None):
                ref_pt.get_perspective_transform(np.zeros((h, w, 3), np.float32), angle, w, h, f, d,
                                                 0.011, -0.02, 0.03, pt_file)
                extra.append([w, h, angle, 1.0 if pt_file else 0.0, *_rec["M"].reshape(9), *_rec["dsize"]])
    out["nonsquare"] = np.array(extra)
    np.savez_compressed(os.path.join(HERE, "homography.npz"), **out)


# ----------------------------------------------------------------- (3) score_fn
def golden_score():
    rows = []
    masks = {"hull14": None, "ones": torch.ones(3, 32, 32), "half": torch.zeros(3, 32, 32)}
    from PIL import Image
    hull = np.array(Image.open(os.path.join(INP, "GTSRB/Hulls/14.png")).convert("RGB"))
    hull = (hull > 128).astype(np.float32).transpose(2, 0, 1)
    masks["hull14"] = torch.from_numpy(hull)
    masks["half"][:, :16, :] = 1
    for mname, m in masks.items():
        obj = float(masks["hull14"].sum() / 3)
        for tr_err in (0.0, 0.1, 0.35, 0.75, 0.7500001, 0.9, 1.0):
            for thr in (0.35, 0.75):
                for lbd in (5, 10):
                    v = ref_score.score_fn(None, m, tr_err, obj, None, threshold=thr, lbd=lbd)
                    vb = ref_score_border.score_fn(None, m, tr_err, obj, None, threshold=thr, lbd=lbd)
                    rows.append({"mask": mname, "tr_err": tr_err, "threshold": thr, "lbd": lbd,
                                 "object_size": obj, "score": float(v), "score_border": float(vb)})
    with open(os.path.join(HERE, "score.json"), "w") as f:
        json.dump(rows, f, indent=0)


# ----------------------------------------------------------------- (4) classifier logits
def golden_nets():
    out = {}
    torch.set_num_threads(8)
    # GTSRBNet
    sd = synth.gtsrb_state_dict("gtsrb-v1")
    net = GTSRBNet()
    net.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()})
    net.eval()
    x = torch.from_numpy(synth.synth_images("gtsrb-golden", 64))
    with torch.no_grad():
        logits = net(x)
    preds = net.predict(x)
    assert torch.equal(preds, logits.argmax(1))
    out["gtsrb_logits"] = logits.numpy()
    out["gtsrb_preds"] = preds.numpy()
    out["gtsrb_single_pred"] = np.array(net.predict(x[3]))
    out["gtsrb_checksum"] = np.array(synth.state_dict_checksum(sd))
    # WideResNet-34-10 (all 252 entries incl. unused sub_block1 and BN buffers)
    sdw = synth.wrn_state_dict("wrn-v1")
    wrn = WideResNet(num_classes=10)
    missing = wrn.load_state_dict({k_: torch.from_numpy(np.asarray(v)) for k_, v in sdw.items()}, strict=True)
    wrn.eval()
    xw = torch.from_numpy(synth.synth_images("wrn-golden", 8, 0.0, 1.0))
    with torch.no_grad():
        lw = wrn(xw)
    pw = wrn.predict(xw)
    out["wrn_logits"] = lw.numpy()
    out["wrn_preds"] = pw.numpy()
    out["wrn_checksum"] = np.array(synth.state_dict_checksum(sdw))
    out["wrn_num_entries"] = np.array(len(sdw))
    np.savez_compressed(os.path.join(HERE, "nets.npz"), **out)
    print("gtsrb preds", preds[:16].tolist(), "wrn preds", pw.tolist())
    print("gtsrb logit range", float(logits.min()), float(logits.max()),
          "wrn", float(lw.min()), float(lw.max()))


# ----------------------------------------------------------------- (5) run_predictions
class _FakeModel:
    """predict() returns prescribed labels, in call order, as the reference's models do."""

    def __init__(self, labels):
        self.labels = torch.as_tensor(labels, dtype=torch.long)
        self.pos = 0
        self.batches = []

    def predict(self, x):
        n = x.shape[0]
        self.batches.append(n)
        r = self.labels[self.pos:self.pos + n]
        self.pos += n
        return r


def golden_run_predictions():
    rows = []
    rng = np.random.RandomState(123)
    for n in (1, 3, 100, 256, 257, 512, 600, 1000):
        labels = rng.randint(0, 4, size=n)
        imgs = [torch.zeros(3, 32, 32) for _ in range(n)]
        for target in (1, None):
            m = _FakeModel(labels)
            err, cnt = ref_utils.run_predictions(m, imgs, 2, target)
            rows.append({"n": n, "labels": labels.tolist(), "label": 2, "target": target,
                         "err": err, "count": cnt, "batches": m.batches})
    with open(os.path.join(HERE, "run_predictions.json"), "w") as f:
        json.dump(rows, f)


# ----------------------------------------------------------------- (6) binary coarse-reduction replay
def golden_binary_search():
    """Run the reference's perform_binary_coarse_reduction with the TR evaluation
    replaced by a table lookup tr[pivot] (number of 4x4 patches in the union mask)."""
    rows = []
    rng = np.random.RandomState(5)

    def run(n, table, err_threshold):
        patches = []
        for i in range(n):
            p = torch.zeros(3, 32, 32)
            p[:, (i // 8) * 4:(i // 8) * 4 + 4, (i % 8) * 4:(i % 8) * 4 + 4] = 1   # disjoint cells
            patches.append(p)
        start_mask = torch.ones(3, 32, 32)
        calls = []

        def fake_gti(img_v, mask, xforms, lbd, theta, pt_file, net_size=32, model_type="GTSRB"):
            piv = int(round(float(mask.sum()) / 3 / 16))
            calls.append(piv)
            return piv

        def fake_rp(model, imgs, lbl_v, lbl_t):
            return 1.0 - table[imgs], 100

        ref_coarse.get_transformed_images = fake_gti
        ref_coarse.run_predictions = fake_rp
        z = torch.zeros(3, 32, 32)
        pivot, best_mask, best_tr, q = ref_coarse.perform_binary_coarse_reduction(
            patches, err_threshold, 14, 1, z, z, torch.zeros(3, 244, 244), None, [None] * 100, start_mask,
            None, 32)
        return {"n": n, "table": [float(t) for t in table], "err_threshold": err_threshold,
                "pivot": int(pivot), "best_tr": float(best_tr), "queries": int(q),
                "mask_bits": float(best_mask.sum() / 3), "calls": calls}

    for n in (1, 2, 3, 7, 16, 33, 64):
        mono = np.sort(rng.randint(0, 101, size=n + 1)) / 100.0
        rows.append(run(n, mono, 0.15))
        rows.append(run(n, mono, 0.5))
        rows.append(run(n, rng.randint(0, 101, size=n + 1) / 100.0, 0.3))       # non-monotone
        rows.append(run(n, np.zeros(n + 1), 0.15))                               # never reached
        rows.append(run(n, np.ones(n + 1), 0.15))                                # reached at 0
    with open(os.path.join(HERE, "binary_search.json"), "w") as f:
        json.dump(rows, f)


if __name__ == "__main__" and "--drivers" not in sys.argv and "--drivers-pg" not in sys.argv:
    xf = golden_xforms()
    golden_homography(xf)
    golden_score()
    golden_run_predictions()
    golden_binary_search()
    golden_nets()
    print("golden fixtures written to", HERE)


# ----------------------------------------------------------------- (7) search-driver control flow
def golden_drivers():
    """Run the reference's generate_mask / boost / get_coarse_reduced_mask with the transform+classify
    pair replaced by tests/fake_tr.FakeTR (a deterministic function of the exact mask/theta bytes).
    cv2.resize -- which these drivers call on 32x32 / 244x244 float images -- is served by the CPU
    oracle's resize (the same one the tests give our drivers), so what is pinned is the reference's
    HOST LOGIC: decisions, masks, thetas, query counts, RNG consumption."""
    import hashlib
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    from fake_tr import FakeTR
    from graphite_amd.imageio import load_image_chw
    import generate_mask as ref_gm
    import boost as ref_boost

    def cv_resize(a, dsize):
        chw = np.ascontiguousarray(np.transpose(np.asarray(a, np.float32), (2, 0, 1)))
        return np.ascontiguousarray(np.transpose(O.resize_linear(chw, dsize[1], dsize[0]), (1, 2, 0)))

    cv2.resize = cv_resize
    cv2.cvtColor = lambda a, code: a
    cv2.COLOR_RGB2BGR = 0

    vic = torch.from_numpy(load_image_chw(os.path.join(INP, "GTSRB/images/14.png")))
    tar = torch.from_numpy(load_image_chw(os.path.join(INP, "GTSRB/images/1.png")))
    hull = torch.from_numpy((load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32))
    vic_s = torch.from_numpy(O.resize_linear(vic.numpy(), 32, 32))
    tar_s = torch.from_numpy(O.resize_linear(tar.numpy(), 32, 32))
    obj = float(hull.sum() / 3)
    ref_dir = (tar_s - vic_s) * hull

    def install(mod, fake):
        mod.get_transformed_images = lambda img, mask, xf, lbd, theta, pt_file, net_size=32, model_type='GTSRB': (mask.clone(), theta.clone())
        mod.run_predictions = lambda model, imgs, lbl_v, lbl_t=None: fake.err(imgs[0], imgs[1])

    def digest(t):
        return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()

    class _M:
        def predict(self, x):
            return 0

    rows = []
    pt = os.path.join(INP, "GTSRB/Points/14.csv")
    for name, T, heat, coarse_mode, tr_lo, tr_hi, max_mask, sharp, mid in (
        ("target_binary", 100, "Target", "binary", 0.65, 0.85, -1, 6.0, 0.35),
        ("target_linear", 50, "Target", "linear", 0.5, 0.8, -1, 6.0, 0.35),
        ("random_binary", 100, "Random", "binary", 0.65, 0.85, -1, 10.0, 0.5),
        ("target_none_maxmask", 40, "Target", "none", 0.4, 0.6, 300, 6.0, 0.2),
        ("victim_binary", 30, "Victim", "binary", 0.3, 0.6, -1, 4.0, 0.3),
    ):
        fake = FakeTR(T, obj, ref_dir, sharp, mid)
        install(ref_gm, fake)
        install(ref_coarse, fake)
        ref_gm.get_coarse_reduced_mask = ref_coarse.get_coarse_reduced_mask
        sys.argv = ["main.py", "--coarse_mode", coarse_mode]
        np.random.seed(42); torch.manual_seed(42)
        import random as _r
        _r.seed(42)
        attacked, qc, best_mask, nbits, best_tr = ref_gm.generate_mask(
            _M(), vic_s, vic, 14, hull.clone(), tar_s, tar, 1, pt, "score.py", heat, 1 - tr_hi, 1 - tr_lo,
            num_xforms=T, net_size=32, model_type='GTSRB', patch_size=4, heatmap_file=None, heatmap_out_path=None,
            max_mask_size=max_mask, init_theta=None)
        rows.append({"kind": "generate_mask", "name": name, "T": T, "heatmap": heat, "coarse_mode": coarse_mode,
                     "tr_lo": tr_lo, "tr_hi": tr_hi, "max_mask_size": max_mask, "sharp": sharp, "mid": mid,
                     "query_count": int(qc), "nbits": float(nbits), "best_tr": float(best_tr),
                     "best_mask": best_mask.numpy()[0].astype(np.uint8).tolist(), "attacked_sha": digest(attacked),
                     "fake_calls": fake.calls, "np_random_next": float(np.random.uniform()),
                     "py_random_next": _r.random()})
        print(name, "queries", qc, "bits", nbits, "tr", best_tr, "calls", fake.calls)

    small_mask = torch.zeros(3, 32, 32)
    small_mask[:, 10:22, 8:24] = 1
    small_mask = small_mask * hull
    for name, T, bt, iters, budget, use_init, model_type in (
        ("rgf_plain", 100, False, 1000, 200, False, "GTSRB"),
        ("rgf_bt", 50, True, 6, 200, False, "GTSRB"),
        ("rgf_init_theta_cifar", 20, False, 3, 200, True, "CIFAR"),
    ):
        fake = FakeTR(T, obj, ref_dir, 6.0, 0.1)
        install(ref_boost, fake)
        ref_boost.add_noise = ref_transforms.add_noise
        ref_boost.convert2Network = lambda *a, **k: torch.zeros(3, 32, 32)
        np.random.seed(42); torch.manual_seed(42)
        init_theta = (tar_s - vic_s) * 0.5 if use_init else None
        adv, tr, perturb, queries, best_theta, adv_small = ref_boost.boost(
            _M(), vic, 14, small_mask, tar_s, 1, beta=1, iterations=iters, pt_file=pt, num_xforms=T, net_size=32,
            model_type=model_type, bt=bt, budget_factor=budget, init_theta=init_theta)
        rows.append({"kind": "boost", "name": name, "T": T, "bt": bt, "iterations": iters, "budget_factor": budget,
                     "use_init": use_init, "model_type": model_type, "tr": float(tr), "queries": int(queries),
                     "best_theta_sha": digest(best_theta), "adv_sha": digest(adv), "perturb_sha": digest(perturb),
                     "adv_small_sha": digest(adv_small), "fake_calls": fake.calls,
                     "torch_random_next": float(torch.rand(1)), "np_random_next": float(np.random.uniform())})
        print(name, "tr", tr, "queries", queries, "calls", fake.calls)
    with open(os.path.join(HERE, "drivers.json"), "w") as f:
        json.dump(rows, f)


# ----------------------------------------------------------------- (8) PatchGuard search drivers
class _PgModel:                                            # module level: pg_boost pickles its arguments, the model included
    def predict(self, x):
        return 0


def golden_drivers_pg():
    """The reference's patchguard/pg_generate_mask.generate_mask and pg_boost.boost, run the same way as golden_drivers():
    transform + classify replaced by FakeTR, cv2.resize served by the CPU oracle.  32x32 frames (the CIFAR path), the
    reference's own plain 32x32 mask, hash-generated victim / target images.  pg_boost writes ./dumped_data.pkl (a
    debugging leftover of the reference): the drivers run inside a temporary directory."""
    import hashlib
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    from fake_tr import FakeTR
    from graphite_amd.imageio import load_image_chw
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = _absent("torchvision.utils.save_image")
    tv.utils = tvu
    sys.modules["torchvision.utils"] = tvu

    def cv_resize(a, dsize):
        chw = np.ascontiguousarray(np.transpose(np.asarray(a, np.float32), (2, 0, 1)))
        return np.ascontiguousarray(np.transpose(O.resize_linear(chw, dsize[1], dsize[0]), (1, 2, 0)))

    cv2.resize = cv_resize
    cv2.cvtColor = lambda a, code: a
    cv2.COLOR_RGB2BGR = 0
    import pg_generate_mask as ref_pgm
    import pg_boost as ref_pgb

    vic = torch.from_numpy(synth.synth_images("pg-victim", 1, 0.0, 1.0)[0])
    tar = torch.from_numpy(synth.synth_images("pg-target", 1, 0.0, 1.0)[0])
    mask = torch.from_numpy((load_image_chw(os.path.join(INP, "cifar/masks/mask.png")) * 255 > 128).astype(np.float32))
    obj = float(mask.sum() / 3)
    ref_dir = (tar - vic) * mask

    def install(mod, fake):
        mod.get_transformed_images = lambda img, m, xf, lbd, theta, pt_file, net_size=32, model_type='GTSRB': (m.clone(), theta.clone())
        mod.run_predictions = lambda model, imgs, lbl_v, lbl_t=None: fake.err(imgs[0], imgs[1])

    def digest(t):
        return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()

    rows = []
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for name, T, heat, coarse_mode, tr_lo, tr_hi, max_mask, sharp, mid in (
                ("pg_target_binary", 100, "Target", "binary", 0.65, 0.85, -1, 6.0, 0.35),
                ("pg_target_linear", 40, "Target", "linear", 0.5, 0.8, -1, 6.0, 0.35),
                ("pg_target_none_maxmask", 40, "Target", "none", 0.4, 0.6, 500, 6.0, 0.2),
                ("pg_victim_binary", 30, "Victim", "binary", 0.3, 0.6, -1, 4.0, 0.3),
            ):
                fake = FakeTR(T, obj, ref_dir, sharp, mid)
                install(ref_pgm, fake)
                install(ref_coarse, fake)
                ref_pgm.get_coarse_reduced_mask = ref_coarse.get_coarse_reduced_mask
                sys.argv = ["main_patchguard.py", "--coarse_mode", coarse_mode]
                np.random.seed(42); torch.manual_seed(42)
                import random as _r
                _r.seed(42)
                attacked, qc, best_mask, nbits, best_tr = ref_pgm.generate_mask(
                    _PgModel(), vic, vic, 3, mask.clone(), tar, tar, 5, None, "score.py", heat, 1 - tr_hi, 1 - tr_lo,
                    num_xforms=T, net_size=32, model_type='CIFAR', patch_size=4, max_mask_size=max_mask)
                rows.append({"kind": "generate_mask", "name": name, "T": T, "heatmap": heat, "coarse_mode": coarse_mode,
                             "tr_lo": tr_lo, "tr_hi": tr_hi, "max_mask_size": max_mask, "sharp": sharp, "mid": mid,
                             "query_count": int(qc), "nbits": float(nbits), "best_tr": float(best_tr),
                             "best_mask": best_mask.numpy()[0].astype(np.uint8).tolist(), "attacked_sha": digest(attacked),
                             "fake_calls": fake.calls, "np_random_next": float(np.random.uniform()),
                             "py_random_next": _r.random()})
                print(name, "queries", qc, "bits", nbits, "tr", best_tr, "calls", fake.calls)

            small_mask = torch.zeros(3, 32, 32)
            small_mask[:, 8:24, 6:26] = 1
            for name, T, bt, iters, budget in (("pg_rgf_plain", 100, False, 1000, 200), ("pg_rgf_bt", 40, True, 5, 200)):
                fake = FakeTR(T, obj, ref_dir, 6.0, 0.1)
                install(ref_pgb, fake)
                ref_pgb.add_noise = ref_pg.add_noise
                np.random.seed(42); torch.manual_seed(42)
                adv, tr, perturb, queries, adv_small = ref_pgb.boost(
                    _PgModel(), vic, 3, small_mask, tar, 5, beta=1, iterations=iters, pt_file=None, num_xforms=T, net_size=32,
                    model_type='CIFAR', bt=bt, budget_factor=budget)
                rows.append({"kind": "boost", "name": name, "T": T, "bt": bt, "iterations": iters, "budget_factor": budget,
                             "tr": float(tr), "queries": int(queries), "adv_sha": digest(adv), "perturb_sha": digest(perturb),
                             "adv_small_sha": digest(adv_small), "fake_calls": fake.calls,
                             "torch_random_next": float(torch.rand(1)), "np_random_next": float(np.random.uniform())})
                print(name, "tr", tr, "queries", queries, "calls", fake.calls)
        finally:
            os.chdir(cwd)
    with open(os.path.join(HERE, "drivers_pg.json"), "w") as f:
        json.dump(rows, f)


if __name__ == "__main__" and "--drivers" in sys.argv:
    golden_drivers()
if __name__ == "__main__" and "--drivers-pg" in sys.argv:
    golden_drivers_pg()

#!/usr/bin/env python3
"""A GTSRBNet that can tell three signs apart, without a dataset: keep the seeded hash weights
(graphite_amd.synth, tag gtsrb-v1) for every layer but fc7 and FIT fc7 by ridge regression on the fc6
features of transformed views of the three sign images the reference ships (14 stop, 1 speed limit 30,
27 pedestrians; random conv features + a trained linear read-out).  The 40 classes without data get a
constant -1 logit.  Purpose: whole-attack runs (bench.py config 2, tests) whose accept / reject decisions
are real; the reference's own checkpoint (GTSRB/checkpoint_us.tar) is not available offline.

Output: graphite_amd/data/gtsrb-v1_toy3_fc7.npz (weight [43,1024], bias [43], float32, 176 KB), loaded by
graphite_amd.synth.gtsrb_state_dict(readout="toy3").

    cd /root/repo && python3 -B tests/golden/make_toy_readout.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import oracle_lib as O                                   # noqa: E402
from graphite_amd import synth, transforms as T, weights  # noqa: E402
from graphite_amd.imageio import load_image_chw           # noqa: E402

INP = os.path.join(os.path.dirname(os.path.dirname(HERE)), "graphite_amd", "data", "inputs")
OUT = os.path.join(REPO, "graphite_amd", "data")
PT14 = os.path.join(INP, "GTSRB/Points/14.csv")
CLASSES = (14, 1, 27)


def views(label, n, seed):
    """n transformed 32x32 network inputs ([-0.5, 0.5]) of sign `label`, through the oracle's TR path."""
    img = load_image_chw(os.path.join(INP, f"GTSRB/images/{label}.png"))
    np.random.seed(seed)
    tab = T.build_xform_table(T.get_transform_params(n, "GTSRB"), 244, 244, PT14)
    z = np.zeros((1, 3, 32, 32), np.float32)
    return O.warp_batch(img, z, z, tab).reshape(-1, 3, 32, 32)


def mixes(n, seed):
    """stop sign with random hull patches of a target pasted in: labelled as the sign covering more of the hull"""
    vic = load_image_chw(os.path.join(INP, "GTSRB/images/14.png"))
    hull = (load_image_chw(os.path.join(INP, "GTSRB/Hulls/14.png")) * 255 > 128).astype(np.float32)
    vic_s = O.resize_linear(vic, 32, 32)
    xs, ys = [], []
    for k, t in enumerate((1, 27)):
        tar_s = O.resize_linear(load_image_chw(os.path.join(INP, f"GTSRB/images/{t}.png")), 32, 32)
        for j, keep in enumerate((0.25, 0.5, 0.75, 0.9)):
            masks = synth.synth_patch_masks(f"toy-{t}-{keep}", 6, hull, p_keep=keep)
            thetas = ((tar_s - vic_s)[None] * masks).astype(np.float32)
            np.random.seed(seed + 10 * k + j)
            tab = T.build_xform_table(T.get_transform_params(n, "GTSRB"), 244, 244, PT14)
            px = O.warp_batch(vic, masks, thetas, tab)                       # [6, n, 3, 32, 32]
            frac = masks[:, 0].sum((1, 2)) / hull[0].sum()
            for c in range(6):
                xs.append(px[c])
                ys.append(np.full(n, t if frac[c] > 0.5 else 14))
    return np.concatenate(xs), np.concatenate(ys)


def features(sd, x):
    """fc6 activations with torch on the CPU (fitting only; evaluation below goes through the oracle)"""
    t = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        h = torch.from_numpy(x)
        h = F.relu(F.conv2d(h, t["conv1_1.weight"], t["conv1_1.bias"]))
        for a, b in (("conv2_1", "conv2_2"), ("conv3_1", "conv3_2"), ("conv4_1", "conv4_2")):
            h = F.relu(F.conv2d(h, t[a + ".weight"], t[a + ".bias"], padding=2))
            h = F.relu(F.conv2d(h, t[b + ".weight"], t[b + ".bias"], padding=2))
            h = F.max_pool2d(h, 2)
        h = h.reshape(-1, 2048)
        h = F.relu(F.linear(h, t["fc5_1.weight"], t["fc5_1.bias"]))
        h = F.relu(F.linear(h, t["fc6_1.weight"], t["fc6_1.bias"]))
    return h.double().numpy()


def main():
    torch.set_num_threads(8)
    sd = synth.gtsrb_state_dict("gtsrb-v1", calibrated=False)
    xs = [views(c, 600, 1000 + i) for i, c in enumerate(CLASSES)]
    ys = [np.full(600, c) for c in CLASSES]
    xm, ym = mixes(40, 2000)
    X = np.concatenate(xs + [xm]).astype(np.float32)
    y = np.concatenate(ys + [ym])
    Fm = features(sd, X)
    Fm1 = np.concatenate([Fm, np.ones((len(Fm), 1))], 1)
    Y = -np.ones((len(y), 43))
    Y[np.arange(len(y)), y] = 1.0
    lam = 1e-3 * np.trace(Fm1.T @ Fm1) / Fm1.shape[1]
    Wb = np.linalg.solve(Fm1.T @ Fm1 + lam * np.eye(Fm1.shape[1]), Fm1.T @ Y)       # [1025, 43]
    W = np.ascontiguousarray(Wb[:-1].T.astype(np.float32))
    b = Wb[-1].astype(np.float32)
    np.savez(os.path.join(OUT, "gtsrb-v1_toy3_fc7.npz"), weight=W, bias=b)
    # held-out check through the oracle
    sd["fc7_1.weight"], sd["fc7_1.bias"] = W, b
    tl = weights.gtsrb_tensor_list(sd)
    for i, c in enumerate(CLASSES):
        p = O.gtsrb_forward(tl, views(c, 300, 3000 + i)).argmax(1)
        print(f"class {c}: held-out accuracy {np.mean(p == c):.3f}; predicted {np.unique(p).tolist()}")
    p = O.gtsrb_forward(tl, X).argmax(1)
    print("train accuracy", np.mean(p == y))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off (hand-run on a GPU box): WideResNet-34-10 logits of N seeded images, HIP vs the C oracle, bit for bit,
and CIFAR-path hit counts for a few candidates.   python tests/fuzz_wrn.py [N]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np, torch
import oracle_lib as O
from cases import cifar_case
from graphite_amd import _hip, engine, models, synth, weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sd = synth.wrn_state_dict()
m = models.WideResNet(); m.load_state_dict(sd)
tl = weights.wrn_tensor_list(sd)
x = synth.synth_images("wrn-fuzz", N, 0.0, 1.0)
t0 = time.time(); ref = O.wrn_forward(tl, 34, 10, 10, x); t1 = time.time()
got = m.forward(torch.from_numpy(x)).cpu().numpy()
print("WRN-34-10 logits of %d images: bit-exact %s (oracle %.1f s); max |logit| %.3f" % (N, np.array_equal(got, ref), t1 - t0, np.abs(ref).max()))
c = cifar_case(num_xforms=12)
masks = np.concatenate([c["hull"][None], synth.synth_patch_masks("wrn-fz", 2, c["hull"])])
thetas = ((c["tar_s"] - c["vic_s"])[None] * masks).astype(np.float32)
px = O.warp_batch(c["vic"], masks, thetas, c["tab"], normalize=False)
lg = O.wrn_forward(tl, 34, 10, 10, px.reshape(-1, 3, 32, 32))
tgt = int(np.bincount(lg.argmax(1)).argmax())
_, h_ref = O.argmax_hits(lg, 12, tgt)
hits, preds = engine.tr_hits(m, c["vic"], masks, thetas, c["tab"], tgt, pt_file=None, model_type='CIFAR', return_preds=True)
print("CIFAR tr_hits 3x12: hits %s ref %s labels equal %s" % (hits.cpu().numpy().tolist(), h_ref.tolist(),
      np.array_equal(preds.cpu().numpy().reshape(-1), lg.argmax(1))))

"""End-to-end attack driver -- the GTSRB / CIFAR path of the reference's `main.py`
(`attack_network` :20-136, `attack_GTSRB` :139-152, `attack_CIFAR` :155-168, `__main__` :171-202)
on top of the batched search drivers.  I/O is Pillow + numpy instead of OpenCV.

    python -m graphite_amd.main -v 14 -t 27 --tr_lo 0.65 --tr_hi 0.85 -s score.py -n GTSRB \\
        --heatmap=Target --coarse_mode=binary -b 100 -m 100 [--checkpoint GTSRB/checkpoint_us.tar]

Without --checkpoint the seeded synthetic weights of graphite_amd.synth are used (the reference's
checkpoints are not redistributable / not available offline).
"""
import getpass
import os
import random
import time

import numpy as np
import torch

from . import parsearguments
from .boost import boost
from .evaluator import TREvaluator
from .generate_mask import generate_mask
from .imageio import load_image_u8_hwc, resize_u8_linear, save_image_chw
from .models import GTSRBNet, WideResNet
from .transforms import add_noise, draw_transforms


def _chw01(u8_hwc):
    return torch.from_numpy(np.ascontiguousarray(u8_hwc.astype(np.float32).transpose(2, 0, 1) / 255.0))


def attack_network(model, img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror, beta=1,
                   num_xforms_mask=100, num_xforms_boost=1000, net_size=32, noise_size=32, model_type='GTSRB',
                   joint_iters=1, image_id='', args=None, write_outputs=True,
                   _evaluator=None):
    """Paths in (victim image, target image, hull mask), PNGs out; returns a dict with what the
    reference prints (transform robustness, mask pixels, queries) plus the tensors."""
    args = args or parsearguments.getarguments()
    issued = []                                            # evaluators created on the way, for the issued-image count
    if _evaluator is None:
        def _evaluator(*a, **k):
            ev = TREvaluator(*a, **k)
            issued.append(ev)
            return ev
    user = getpass.getuser()
    tag = (user + "_" + (image_id + "_" if image_id != '' else '') + str(lbl_v) + '_' + str(lbl_t) + '_' + str(reduceerror) +
           '_' + str(coarseerror) + '_' + str(args.coarse_mode) + '_' + scorefile + "_")
    # main.py:31-33: outputs/<model_type>/ except for GTSRB; main_patchguard.py:29-31 writes straight into its base directory
    out_base = args.out_path + (model_type + '/' if model_type not in ('GTSRB', 'PATCHGUARD') else '')

    img_u8 = load_image_u8_hwc(img_v)
    img_torch = _chw01(img_u8)
    img_small_torch = TREvaluator.resize(img_torch, noise_size, noise_size)          # float resize, main.py:38
    mask_u8 = load_image_u8_hwc(mask)
    if mask_u8.shape[0] != noise_size or mask_u8.shape[1] != noise_size:
        mask_u8 = resize_u8_linear(mask_u8, noise_size, noise_size)
    mask_t = torch.from_numpy(np.where(mask_u8 > 128, 255, 0).transpose(2, 0, 1).copy()) / 255.0
    tar_large_u8 = load_image_u8_hwc(img_t)
    tar = _chw01(resize_u8_linear(tar_large_u8, noise_size, noise_size))              # 8-bit resize, main.py:59
    tar_large = _chw01(tar_large_u8)

    prior_mask, next_theta = mask_t.clone().float(), None
    prior_attack, prior_attack_large = tar.clone(), tar_large.clone()
    total_mask_q = total_boost_q = 0
    res = {}
    for i in range(joint_iters):
        if args.square_x is not None:
            assert joint_iters == 1 and args.square_y is not None
            sx, sy, ss = int(args.square_x), int(args.square_y), int(args.square_size)
            mask_out = torch.zeros(3, noise_size, noise_size)
            mask_out[:, sy:sy + ss, sx:sx + ss] = 1
            best_theta = (tar - img_small_torch) * mask_out
            init = add_noise(img_torch, mask_out, 1.0, best_theta)
            nbits = ss ** 2
            xforms = draw_transforms(num_xforms_mask, model_type)
            ev = TREvaluator(model, img_torch, xforms, pt_file, model_type, lbl_v, lbl_t)
            success_rate, _ = ev.error(mask_out, best_theta)
            tr_score = 1 - success_rate
        else:
            init, q, mask_out, nbits, tr_score = generate_mask(
                model, img_small_torch, img_torch, lbl_v, prior_mask, prior_attack, prior_attack_large, lbl_t, pt_file,
                scorefile, heatmap, coarseerror, reduceerror, num_xforms=num_xforms_mask, net_size=net_size,
                model_type=model_type, patch_size=noise_size // 8, heatmap_file=args.heatmap_file,
                heatmap_out_path=(out_base + 'heatmaps/' + user + "_" + str(lbl_v) + '_' + str(lbl_t) + '_heatmap.pkl')
                if write_outputs else None, max_mask_size=args.max_mask_size, init_theta=next_theta, _args=args, _evaluator=_evaluator)
            total_mask_q += q
        nbits = int(nbits)
        suffix = '_' + str(nbits) + '_bits_' + str(tr_score) + '_tr_' + str(i) + '_joint_iter.png'
        if write_outputs:
            save_image_chw(out_base + 'masks/' + tag + suffix, mask_out)
            save_image_chw(out_base + 'inits/' + tag + suffix, init)
        adversarial, tr_score, perturb, bq, next_theta, adv_small = boost(
            model, img_torch, lbl_v, mask_out, prior_attack, lbl_t, beta=beta, iterations=1000, pt_file=pt_file,
            num_xforms=num_xforms_boost, net_size=net_size, model_type=model_type, bt=False,
            square_mask=args.square_x is not None, early_boost_exit=args.early_boost_exit, init_theta=next_theta,
            _evaluator=_evaluator)
        total_boost_q += bq
        prior_mask = mask_out.clone()
        prior_attack_large = adversarial.clone()
        prior_attack = TREvaluator.resize(prior_attack_large, noise_size, noise_size)
        if write_outputs:
            save_image_chw(out_base + 'boosted/' + tag + suffix, adversarial)
            save_image_chw(out_base + 'perturbations/' + tag + suffix.replace('_bits_', 'bits_', 1), perturb)
        res = dict(adversarial=adversarial, mask=mask_out, theta=next_theta, init=init, perturb=perturb)

    print("--------------------------------")
    print("Attack Completed.")
    if args.num_test_xforms > -1:
        xforms = draw_transforms(args.num_test_xforms, model_type)
        ev = TREvaluator(model, res["adversarial"], xforms, pt_file, model_type, lbl_v, lbl_t, lbd=0.0)
        success_rate, _ = ev.error(res["mask"], torch.zeros(res["mask"].size()))
        res["final_test_tr"] = 1.0 - success_rate
        print("Final transform_robustness (" + str(args.num_test_xforms) + " transforms):", 1.0 - success_rate)
    print("Final transform_robustness:", tr_score)
    print("Final number of pixels:", nbits)
    print("Final number of queries:", total_mask_q + total_boost_q)
    res.update(transform_robustness=tr_score, pixels=nbits, queries=total_mask_q + total_boost_q,
               mask_queries=total_mask_q, boost_queries=total_boost_q)
    if issued:      # images really classified, discarded speculation included (`queries` is the reference-equivalent count)
        res["queries_issued"] = int(sum(ev.issued for ev in issued))
        print("Images classified including discarded speculation:", res["queries_issued"])
    return res


def _load_weights(model, checkpoint, key, synth_fn):
    if checkpoint:
        sd = torch.load(checkpoint, map_location="cpu", weights_only=True)
        model.load_state_dict(sd[key] if key and key in sd else sd)
    else:
        print("no --checkpoint given: using graphite_amd.synth seeded weights")
        model.load_state_dict(synth_fn())
    return model


def attack_GTSRB(img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror, beta=1,
                 num_xforms_mask=100, num_xforms_boost=1000, net_size=32, noise_size=32, joint_iters=1, checkpoint=None,
                 **kw):
    from . import synth
    model = _load_weights(GTSRBNet(), checkpoint, 'model_state_dict', synth.gtsrb_state_dict)
    return attack_network(model, img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror,
                          beta, num_xforms_mask, num_xforms_boost, net_size, noise_size, model_type='GTSRB',
                          joint_iters=joint_iters, **kw)


def attack_CIFAR(img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror, beta=1,
                 num_xforms_mask=100, num_xforms_boost=1000, net_size=32, noise_size=32, joint_iters=1, image_id='',
                 checkpoint=None, **kw):
    from . import synth
    model = _load_weights(WideResNet(num_classes=10), checkpoint, None, synth.wrn_state_dict)
    return attack_network(model, img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror,
                          beta, num_xforms_mask, num_xforms_boost, net_size, noise_size, model_type='CIFAR',
                          joint_iters=joint_iters, image_id=image_id, **kw)


def main(argv=None):
    import argparse
    extra = argparse.ArgumentParser(add_help=False)
    extra.add_argument('--checkpoint')
    extra.add_argument('--no_outputs', action='store_true')
    ex, rest = extra.parse_known_args(argv)
    args = parsearguments.getarguments(rest)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    if not ex.no_outputs:
        base = args.out_path + (args.network + '/' if args.network != 'GTSRB' else '')
        for d in ('masks', 'inits', 'boosted', 'perturbations', 'heatmaps'):
            os.makedirs(base + d, exist_ok=True)
    t0 = time.time()
    fn = attack_GTSRB if args.network == 'GTSRB' else attack_CIFAR
    kw = dict(image_id=args.image_id) if args.network == 'CIFAR' else {}
    res = fn(args.img_v, args.img_t, args.mask, args.lbl_v, args.lbl_t, args.pt_file, args.scorefile, args.heatmap,
             args.coarse_error, args.reduce_error, 1, args.mask_transforms, args.boost_transforms, net_size=32,
             noise_size=32, joint_iters=args.joint_iters, checkpoint=ex.checkpoint, args=args,
             write_outputs=not ex.no_outputs, **kw)
    print("\n\nTotal running time: %.4f seconds\n" % (time.time() - t0))
    return res


if __name__ == '__main__':
    main()

"""Drop-in for the reference's `patchguard/main_patchguard.py`: `attack_network` (:19-105) with the same positional
arguments.  Paths in, PNGs out (under `out_path`, default the reference's 'outputs_reproduce/'); returns the dict
graphite_amd.main.attack_network returns.  `attack_CIFAR` (:108-119) builds the PatchGuard classifier from the checkpoint
path the reference hard-codes (patchguard_predict.py:28); that file is a download (README.md:117), so without it pass a
model (e.g. PatchGuard(synth.bagnet_state_dict())) to `attack_network`."""
from .. import main as _M
from .. import parsearguments


def attack_network(model, img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror, beta=1,
                   num_xforms_mask=100, num_xforms_boost=1000, net_size=32, noise_size=32, model_type='CIFAR',
                   out_path='outputs_reproduce/', args=None, write_outputs=True, _evaluator=None):
    assert model_type == 'CIFAR'
    args = args or parsearguments.getarguments()
    # the PatchGuard driver takes only coarse_mode from the command line (:23-24): no heatmap file, no mask-size cap,
    # no square mask, no extra test transforms
    pg_args = parsearguments.getarguments(["--coarse_mode", str(args.coarse_mode)])
    pg_args.out_path = out_path
    return _M.attack_network(model, img_v, img_t, mask, lbl_v, lbl_t, pt_file, scorefile, heatmap, coarseerror, reduceerror,
                             beta, num_xforms_mask, num_xforms_boost, net_size, noise_size, model_type='PATCHGUARD',
                             joint_iters=1, args=pg_args, write_outputs=write_outputs, _evaluator=_evaluator)


def attack_CIFAR(*a, checkpoint='PatchGuard/checkpoints/bagnet17_192_cifar.pth', **k):
    import os
    import torch
    from .patchguard_predict import PatchGuard
    if not os.path.exists(checkpoint):
        raise FileNotFoundError(checkpoint + ": the BagNet-17 checkpoint of inspire-group/PatchGuard is a separate download "
                                "(reference README.md:117); pass a model to attack_network instead")
    return attack_network(PatchGuard(torch.load(checkpoint, map_location='cpu', weights_only=True)), *a, **k)

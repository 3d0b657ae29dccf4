"""Drop-in for the reference's `patchguard/pg_generate_mask.py` (same signatures and return tuples).

The PatchGuard driver is an older twin of generate_mask.py: it draws the BASELINE pg parameter table
(pg_generate_mask.py:43), has no init_theta, and its max_mask_size loop shares one patch list between passes, raises lbd by
2.5 and reports the last pass's queries only (:119-131).  All of that is selected inside graphite_amd.generate_mask by
model_type 'PATCHGUARD'; candidates are evaluated in batches exactly as for the other flavours.
"""
from .. import generate_mask as _G

_PG = 'PATCHGUARD'


def generate_mask(model, img_v_small, img_v, lbl_v, mask, img_t_small, img_t, lbl_t, pt_file, scorefile, HEATMAP_MODE,
                  COARSE_ACCEPT_THRESHOLD, FINE_REJECT_THRESHOLD, num_xforms=100, net_size=32, model_type='CIFAR',
                  patch_size=4, stride_factor=4, heatmap_file=None, heatmap_out_path=None, max_mask_size=-1,
                  _evaluator=None, _args=None):
    assert model_type == 'CIFAR'
    return _G.generate_mask(model, img_v_small, img_v, lbl_v, mask, img_t_small, img_t, lbl_t, pt_file, scorefile,
                            HEATMAP_MODE, COARSE_ACCEPT_THRESHOLD, FINE_REJECT_THRESHOLD, num_xforms, net_size, _PG,
                            patch_size, stride_factor, heatmap_file, heatmap_out_path, max_mask_size, None,
                            _evaluator=_evaluator, _args=_args)


def get_heatmap(start_mask, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file, net_size, init_tr=0.5,
                plot=False, _evaluator=None):
    return _G.get_heatmap(start_mask, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file, net_size,
                          init_tr, plot, model_type=_PG, _evaluator=_evaluator)


def get_fine_reduced_mask(start_mask, object_size, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms, pt_file,
                          net_size, err_threshold=0.75, HEATMAP_MODE=None, scoremod=None, lbd=5, max_mask_size=-1,
                          _evaluator=None):
    return _G.get_fine_reduced_mask(start_mask, object_size, patches, indices, img_v, img_t, lbl_v, lbl_t, model, xforms,
                                    pt_file, net_size, err_threshold, HEATMAP_MODE, scoremod, lbd, max_mask_size,
                                    model_type=_PG, _evaluator=_evaluator)


def survey_heatmap(mask, patches, indices, img_v, img_t, img_v_small, img_t_small, lbl_t, lbl_v, model, xforms, pt_file,
                   net_size, HEATMAP_MODE, plot=False, _evaluator=None):
    return _G.survey_heatmap(mask, patches, indices, img_v, img_t, img_v_small, img_t_small, lbl_t, lbl_v, model, xforms,
                             pt_file, net_size, HEATMAP_MODE, plot, model_type=_PG, _evaluator=_evaluator)

"""graphite_amd -- GRAPHITE's transform-robustness scoring path on MI355X (gfx950).

Drop-in names (reference file:line in each module's docstring):
    transforms.get_transform_params / get_transformed_images / add_noise / convert2Network /
    apply_transformation, perspective_transform.get_perspective_transform,
    utils.run_predictions, models.GTSRBNet / WideResNet (.predict), score.score_fn
    transforms.transform_wb / convert2NetworkWB (differentiable twin), generate_mask / coarse_reduction / boost / main
    (batched search drivers), whitebox_attack (EOT loop)
Added: `tr_hits` (fused, batched evaluation that never leaves the GPU), `engine.eot_gradient`.
"""
from .engine import tr_hits, tr_error_rates  # noqa: F401

__all__ = ["tr_hits", "tr_error_rates"]

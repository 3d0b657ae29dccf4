"""Victim classifiers with the reference's duck-typed interface (`.predict`, `.load_state_dict`).

  GTSRBNet    reference GTSRB/GTSRBNet.py:7-70
  WideResNet  reference cifar/wideresnet.py:52-111

Weights live in the HIP library (repacked for the implicit-GEMM kernels); `predict` keeps the
reference's pre-processing quirks (re-normalise only when the input is out of range).
"""
import numpy as np
import torch

from . import _hip, weights


class _HipClassifier:
    _kind = None

    def __init__(self):
        self._model = None

    # the reference calls these on nn.Module; they are no-ops for an inference-only model
    def eval(self):
        return self

    def cuda(self):
        return self

    def _require(self):
        if self._model is None:
            raise RuntimeError("weights not loaded: call load_state_dict() first")
        return self._model

    def __call__(self, x):
        return self.forward(x)

    def forward(self, x):
        """logits as a CUDA tensor [N, num_classes]; x = [N,3,32,32] already in the net's input range."""
        return self._require().forward(x)

    def _predict(self, image):
        single = image.dim() < 4
        batch = image.unsqueeze(0) if single else image
        preds = self._require().predict_dev(batch)
        return int(preds[0].item()) if single else preds

    @property
    def hip_model(self):
        return self._require()

    def set_precision(self, precision):
        """'f32' (default, the bit-exact parity path) or 'bf16' (throughput mode, bf16 operands with f32 accumulation): see gr_model_set_precision."""
        self._require().set_precision(precision)
        return self


class GTSRBNet(_HipClassifier):
    num_classes = 43

    def load_state_dict(self, state_dict, strict=True):
        self._model = _hip.Model("gtsrb", weights.gtsrb_tensor_list(state_dict))
        return self

    def predict(self, image):
        """GTSRBNet.py:54-70: inputs outside [-0.5, 0.5] are clamped to [0,1] and shifted by -0.5."""
        image = torch.as_tensor(image, dtype=torch.float32)
        if torch.max(image) > 0.5 or torch.min(image) < -0.5:
            image = torch.clamp(image, 0.0, 1.0) - 0.5
        return self._predict(image)


class WideResNet(_HipClassifier):
    def __init__(self, depth=34, num_classes=10, widen_factor=10, dropRate=0.0):
        super().__init__()
        assert (depth - 4) % 6 == 0
        self.depth, self.num_classes, self.widen = depth, num_classes, widen_factor

    def load_state_dict(self, state_dict, strict=True):
        self._model = _hip.Model("wrn", weights.wrn_tensor_list(state_dict, self.depth, self.widen),
                                 depth=self.depth, widen=self.widen, num_classes=self.num_classes)
        return self

    def predict(self, image):
        """wideresnet.py:96-111: inputs outside [0,1] are clamped."""
        image = torch.as_tensor(image, dtype=torch.float32)
        if torch.max(image) > 1.0 or torch.min(image) < 0.0:
            image = torch.clamp(image, 0.0, 1.0)
        return self._predict(image)

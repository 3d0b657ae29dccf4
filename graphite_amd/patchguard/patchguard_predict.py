"""Drop-in for the reference's `patchguard/patchguard_predict.py` (PatchGuard / PatchGuardNoDefense, :21-89) on the HIP
library.

  forward (:38-50)   z = (x + 0.5) * 255; z = F.interpolate(z, 192, mode='bicubic') / 255; z = Normalize(...)(z)
                     local = bagnet17(aggregation='none')(z); pred_i = masking_defense(local[i], thres=0.0, window [6, 6])

Lines 42-44 are reference code and run as one HIP kernel (`k_pg_pre`, fused with BagNet's first 1x1 conv).  BagNet-17 and
`masking_defense` come from inspire-group/PatchGuard (README.md:113-117), third-party source that GRAPHITE neither vendors
nor can fetch here, with a checkpoint (`bagnet17_192_cifar.pth`) that is absent: both are RESTATED from the publications
(include/graphite_hip.h says from which) -- "parity unpinned" -- and run on the fp32 MFMA implicit-GEMM kernels
(`gr_model_load_bagnet`) and `k_pg_masking`.  Weights: any BagNet state_dict (the authors' checkpoint if it is ever
supplied, DataParallel 'module.' prefixes accepted); without one, `synth.bagnet_state_dict()` gives the seeded stand-in the
tests and bench.py use.

A different local-logit network can be plugged in: `PatchGuard(net=callable)` keeps the HIP pre-processing and the HIP
masking epilogue and calls `net(z)` -> [B,h,w,num_classes] for the middle.
"""
from math import ceil

import torch

from .. import _hip, weights

RF_SIZE, RF_STRIDE, PATCH_SIZE = 17, 8, 30
WINDOW_SIZE = ceil((PATCH_SIZE + RF_SIZE - 1) / RF_STRIDE)          # patchguard_predict.py:33-36 -> 6
BAGNET17 = dict(layers=(3, 4, 6, 3), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0))


class PatchGuard:
    """model.predict(x): x [B,3,32,32] (or [3,32,32]) in [-0.5, 0.5] -> int64 labels [B] on the GPU (:38-53)."""
    _defense = True
    num_classes = 10

    def __init__(self, state_dict=None, net=None, size=192, layers=BAGNET17["layers"], strides=BAGNET17["strides"],
                 kernel3=BAGNET17["kernel3"], num_classes=10, window_size=None):
        self.size, self.num_classes = int(size), int(num_classes)
        self._arch = dict(layers=tuple(layers), strides=tuple(strides), kernel3=tuple(kernel3))
        self._net = net                      # duck-typed local-logit network (normalised [B,3,size,size] -> [B,h,w,ncls])
        self._model = None
        self._window = window_size
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        sd = state_dict.get("net", state_dict) if isinstance(state_dict, dict) else state_dict     # :28-29 checkpoint['net']
        self._model = _hip.Model("bagnet", weights.bagnet_tensor_list(sd, self._arch["layers"]), num_classes=self.num_classes,
                                 upsample=self.size, **self._arch)
        w = self._window if self._window is not None else min(WINDOW_SIZE, self._model.pg_out_size)
        self._model.pg_configure(self._defense, w, 0.0)
        return self

    def eval(self):
        return self

    def cuda(self):
        return self

    @property
    def hip_model(self):
        if self._model is None:             # AttributeError: hasattr(model, "hip_model") is how the evaluator asks for the fused path
            raise AttributeError("PatchGuard: no BagNet weights loaded (load_state_dict) -- with a duck-typed `net` only "
                               "predict()/forward() are available, not the fused tr_hits path")
        return self._model

    def set_precision(self, precision):
        """'f32' (default, bit-exact against the oracle) or 'bf16': BagNet's stride-1 1x1 convs (74 % of bagnet17's flops) on
        bf16 operands with f32 accumulation -- a throughput mode, as for the other classifiers (gr_model_set_precision)."""
        self.hip_model.set_precision(precision)
        return self

    def local_logits(self, x):
        x = torch.as_tensor(x, dtype=torch.float32)
        if self._net is not None:
            return _hip.to_dev(self._net(_hip.pg_preprocess(x, self.size)))
        return self.hip_model.pg_local_logits(x)

    def forward(self, *x):
        if type(x) == tuple:                                  # :39-41
            assert len(x) == 1
            x = x[0]
        x = torch.as_tensor(x, dtype=torch.float32)
        single = x.dim() < 4
        if single:
            x = x.unsqueeze(0)
        if self._net is None:
            preds = self.hip_model.predict_dev(x)
        else:
            local = self.local_logits(x)
            if self._defense:
                w = self._window if self._window is not None else min(WINDOW_SIZE, local.shape[1])
                preds = _hip.pg_masking_defense(local, w, 0.0)[1]
            else:
                preds = local.mean(dim=(1, 2)).argmax(1)
        return int(preds[0].item()) if single else preds

    __call__ = forward

    def predict(self, x):
        return self.forward(x)


class PatchGuardNoDefense(PatchGuard):
    """:56-89: aggregation='mean' and a plain argmax."""
    _defense = False

"""Plain-PyTorch / numpy restatement of the PatchGuard robust classifier the reference wraps
(patchguard/patchguard_predict.py:21-89).  TEST INFRASTRUCTURE: imported by tests/ and bench.py's checker leg only.

What is what:
  preprocess()        patchguard_predict.py:42-44 + :30-32 -- REFERENCE CODE, reproduced with the same torch calls
                      (F.interpolate(mode='bicubic'); torchvision's Normalize is (z - mean) / std).  torch is importable
                      here, so this function IS the oracle for that stage (tolerance 1e-5, tests/test_patchguard_cpu.py).
  BagNet              THIRD PARTY: `PatchGuard.nets.bagnet.bagnet17(clip_range=None, aggregation=...)` from
                      inspire-group/PatchGuard @ ae23629644d8628ef376d3121c35de261472daf5 (README.md:113-117), itself
                      Brendel & Bethge's BagNet (bagnets/pytorchnet.py) with the final fc applied per position.  Its source
                      is not under /root/reference and cannot be fetched: the ARCHITECTURE is restated from the publications
                      (ICLR 2019 "Approximating CNNs with bag-of-local-features models", PatchGuard USENIX Sec. 2021 3.2);
                      the authors' weights (bagnet17_192_cifar.pth) are absent.  PARITY UNPINNED.
  masking_defense     THIRD PARTY: `PatchGuard.utils.defense_utils.masking_defense`; restated from PatchGuard's Algorithm 1
                      (robust masking): clip local logits to [0, inf), per class detect the window with the largest class
                      evidence, mask it when its share of the class evidence exceeds the threshold, argmax of the masked
                      class evidence.  Recalled details marked [recalled].  PARITY UNPINNED.
"""
from math import ceil

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CIFAR_MEAN = (0.4914, 0.4822, 0.4465)            # patchguard_predict.py:31
CIFAR_STD = (0.2023, 0.1994, 0.2010)
RF_SIZE, RF_STRIDE, PATCH_SIZE = 17, 8, 30       # :33-35
WINDOW_SIZE = ceil((PATCH_SIZE + RF_SIZE - 1) / RF_STRIDE)      # :36 -> 6


def preprocess(x, size=192):
    """patchguard_predict.py:42-44: x [B,3,32,32] in [-0.5, 0.5] -> normalised [B,3,size,size]."""
    z = (x + 0.5) * 255
    z = F.interpolate(z, size, mode='bicubic') / 255
    mean = torch.tensor(CIFAR_MEAN, dtype=z.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CIFAR_STD, dtype=z.dtype).view(1, 3, 1, 1)
    return (z - mean) / std


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, kernel_size=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=kernel_size, stride=stride, padding=0, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        if residual.size(-1) != out.size(-1):                 # unpadded 3x3: crop the residual at the bottom / right
            diff = residual.size(-1) - out.size(-1)
            residual = residual[:, :, :-diff, :-diff]
        return self.relu(out + residual)


class BagNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0), num_classes=10, aggregation='none'):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=0.001)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = self._make_layer(64, layers[0], strides[0], kernel3[0])
        self.layer2 = self._make_layer(128, layers[1], strides[1], kernel3[1])
        self.layer3 = self._make_layer(256, layers[2], strides[2], kernel3[2])
        self.layer4 = self._make_layer(512, layers[3], strides[3], kernel3[3])
        self.fc = nn.Linear(512 * 4, num_classes)
        self.aggregation = aggregation

    def _make_layer(self, planes, blocks, stride, kernel3):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, kernel_size=1 if kernel3 == 0 else 3)]
        self.inplanes = planes * 4
        for i in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, kernel_size=1 if kernel3 <= i else 3))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv2(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.fc(x.permute(0, 2, 3, 1))                    # local logits [B, h, w, num_classes]
        if self.aggregation == 'mean':
            x = torch.mean(x, dim=(1, 2))
        return x


def bagnet17(**kw):
    return BagNet((3, 4, 6, 3), (2, 2, 2, 1), (1, 1, 1, 0), **kw)


def masking_defense(local_feature, thres=0.0, window_shape=(6, 6)):
    """local_feature: numpy [h, w, num_cls] -> int robust prediction (PatchGuard Algorithm 1)."""
    h, w, ncls = local_feature.shape
    wx, wy = window_shape
    u = np.clip(local_feature, 0, np.inf)                     # clipping < 0 means [0, inf)  [recalled]
    total = np.sum(u, axis=(0, 1))
    win = np.zeros([h - wx + 1, w - wy + 1, ncls], dtype=u.dtype)
    for x in range(h - wx + 1):
        for y in range(w - wy + 1):
            win[x, y] = np.sum(u[x:x + wx, y:y + wy], axis=(0, 1))
    max_win = np.max(win, axis=(0, 1))                        # per class: the window holding the most evidence
    with np.errstate(divide='ignore', invalid='ignore'):
        detected = max_win / total > thres
    masked = np.where(detected, total - max_win, total)
    return int(np.argsort(masked, kind='stable')[-1])         # [recalled] stable argsort: the LAST of equal maxima


class PatchGuard(nn.Module):
    """patchguard_predict.py:21-53 with a BagNet given by the caller (seeded weights: the checkpoint is absent)."""

    def __init__(self, net, size=192, window=WINDOW_SIZE):
        super().__init__()
        self.net, self.size, self.window = net, size, window

    def local_logits(self, x):
        return self.net(preprocess(x, self.size))

    def forward(self, x):
        z = self.local_logits(x).detach().cpu().numpy()
        if z.ndim == 2:                                       # aggregation='mean': PatchGuardNoDefense, :83
            return torch.tensor([int(np.argmax(z[i])) for i in range(z.shape[0])])
        return torch.tensor([masking_defense(z[i], thres=0.0, window_shape=[self.window, self.window])
                             for i in range(z.shape[0])])

    predict = forward


def bagnet_local_logits_folded(sd, z, layers=(3, 4, 6, 3), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0), bf16=False):
    """BagNet forward on a state dict with every conv + eval BatchNorm FOLDED as the HIP library and the C oracle fold them
    (w' = w * scale[co] in float32, bias' = shift[co]); with bf16=True the stride-1 1x1 convs take activations and folded
    weights rounded to bfloat16 (nearest even) and accumulate in float64 -- the tolerance reference of the library's
    GR_PREC_BF16 mode for this net (the matrix cores accumulate in float32; 3x3 convs, strided shortcuts and the fc stay fp32)."""
    t = {k: torch.as_tensor(v) for k, v in sd.items()}

    def fold(conv, bn):
        w = t[conv + ".weight"].float()
        inv = 1.0 / torch.sqrt(t[bn + ".running_var"].float() + 1e-5)
        sc = t[bn + ".weight"].float() * inv
        return w * sc.view(-1, 1, 1, 1), t[bn + ".bias"].float() - t[bn + ".running_mean"].float() * sc

    def conv(x, w, b, stride=1):
        if bf16 and w.shape[-1] == 1 and stride == 1 and w.shape[1] % 32 == 0 and w.shape[0] % 64 == 0:
            y = F.conv2d(x.bfloat16().double(), w.bfloat16().double(), None, 1).float()
        else:
            y = F.conv2d(x, w, None, stride)
        return y + b.view(1, -1, 1, 1)

    x = F.conv2d(z, t["conv1.weight"].float())
    w, b = fold("conv2", "bn1")
    x = F.relu(conv(x, w, b))
    for g, nblk in enumerate(layers):
        for i in range(nblk):
            p = f"layer{g + 1}.{i}"
            s = strides[g] if i == 0 else 1
            w, b = fold(p + ".conv1", p + ".bn1"); o = F.relu(conv(x, w, b))
            w, b = fold(p + ".conv2", p + ".bn2"); o = F.relu(conv(o, w, b, s))
            w, b = fold(p + ".conv3", p + ".bn3"); o = conv(o, w, b)
            r = x
            if i == 0:
                w, b = fold(p + ".downsample.0", p + ".downsample.1"); r = conv(x, w, b, s)
            if r.size(-1) != o.size(-1):
                d = r.size(-1) - o.size(-1)
                r = r[:, :, :-d, :-d]
            x = F.relu(o + r)
    return F.linear(x.permute(0, 2, 3, 1), t["fc.weight"].float(), t["fc.bias"].float())

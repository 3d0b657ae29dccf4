"""A deterministic stand-in for "transform + classify" used to pin the SEARCH DRIVERS' control flow
(boost / generate_mask / coarse_reduction) against the reference's own drivers: both sides are run
with this same evaluator, so any difference in decisions, masks or query counts is host logic.

err(mask, theta) depends on the mask area, on the direction of theta and on a hash of the exact
bytes of both tensors -- a driver whose tensors differ by one ulp from the reference's diverges.
"""
import zlib

import numpy as np
import torch


class FakeTR:
    def __init__(self, T, object_size, ref_dir, sharp=6.0, mid=0.35):
        self.T, self.object_size = T, float(object_size)
        self.ref = ref_dir / (torch.norm(ref_dir) + 1e-12)
        self.sharp, self.mid = sharp, mid
        self.calls = 0

    def hits(self, mask, theta):
        self.calls += 1
        frac = float(mask.sum()) / 3.0 / self.object_size
        nt = float(torch.norm(theta))
        cos = float((theta * self.ref).sum()) / nt if nt > 0 else 0.0
        base = 1.0 / (1.0 + np.exp(-self.sharp * (frac - self.mid)))
        tr = base * (0.55 + 0.45 * cos)
        h = zlib.crc32(mask.numpy().tobytes() + theta.numpy().tobytes())
        tr += ((h % 7) - 3) / 100.0
        return int(round(self.T * min(max(tr, 0.0), 1.0)))

    def err(self, mask, theta):
        return (1.0 - self.hits(mask, theta) * 1.0 / self.T), self.T


class FakeEvaluator:
    """Same seam as graphite_amd.evaluator.TREvaluator, backed by FakeTR + the CPU oracle's image ops."""

    def __init__(self, fake, oracle, speculate=True):
        self.fake, self.oracle = fake, oracle
        self.speculate = speculate        # may the drivers evaluate candidates the sequential algorithm might not ask for?
        self.launches = 0
        self.evaluated = 0
        self.order = []                   # crc of every (mask, theta) evaluated, in order

    def errors(self, masks, thetas):
        self.launches += 1
        self.evaluated += len(masks)
        self.order += [zlib.crc32(m.numpy().tobytes() + t.numpy().tobytes()) for m, t in zip(masks, thetas)]
        return [self.fake.err(m, t) for m, t in zip(masks, thetas)]

    def error(self, mask, theta):
        return self.errors([mask], [theta])[0]

    def resize(self, t, h, w):
        return torch.from_numpy(self.oracle.resize_linear(np.ascontiguousarray(t.numpy(), dtype=np.float32), h, w))

    def add_noise(self, image, mask, lbd, theta, clip=True):
        comb, _, _ = self.oracle.add_noise(image.numpy(), mask.numpy(), theta.numpy(), lbd, clip)
        return torch.from_numpy(comb)

    def swapped(self, img, label, target):
        return self

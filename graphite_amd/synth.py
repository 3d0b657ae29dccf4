"""Version-independent synthetic tensors (weights, images, masks).

The reference ships no checkpoints (`GTSRB/checkpoint_us.tar` and
`cifar/epoch-39.ckpt` are absent, reference main.py:149,165), so tests, goldens
and the benchmark use seeded stand-ins.  They must be identical in this
container (where the goldens are made) and on the GPU box, whatever numpy /
torch version generates them, so nothing here touches `torch.manual_seed` or
`np.random`: every element is a pure integer hash of (tensor name, flat index).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """float32 array, element i = lo + (hi-lo) * u24(hash(name, i)); u24 in [0,1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(_fnv1a64(name))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    bits = _splitmix64(idx) >> np.uint64(40)          # 24 random bits
    u = bits.astype(np.float64) / 16777216.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------
# GTSRBNet (reference GTSRB/GTSRBNet.py:8-33): state_dict key -> shape
# --------------------------------------------------------------------------
GTSRB_SHAPES = [
    ("conv1_1", (3, 3, 1, 1)),
    ("conv2_1", (32, 3, 5, 5)),
    ("conv2_2", (32, 32, 5, 5)),
    ("conv3_1", (64, 32, 5, 5)),
    ("conv3_2", (64, 64, 5, 5)),
    ("conv4_1", (128, 64, 5, 5)),
    ("conv4_2", (128, 128, 5, 5)),
    ("fc5_1", (1024, 2048)),
    ("fc6_1", (1024, 1024)),
    ("fc7_1", (43, 1024)),
]


def _calibrated_bias(tag):
    """Last-layer bias centred on a calibration set (tests/golden/make_calibration.py): random-init
    nets otherwise predict one constant class, which would make hit-count parity trivially true."""
    path = os.path.join(_DATA, f"{tag}_last_bias.npy")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} missing: run tests/golden/make_calibration.py")
    return np.load(path, allow_pickle=False).astype(np.float32)


def gtsrb_state_dict(tag: str = "gtsrb-v1", calibrated: bool = True, readout: str = None):
    """dict[str, np.ndarray float32] with the reference's state_dict keys.

    He-uniform weights (bound sqrt(6/fan_in)) keep the activations O(1) through
    the net so the 43 logits are well separated for most inputs; biases are
    small uniform values.

    readout="toy3": fc7 is replaced by the linear read-out fitted by tests/golden/make_toy_readout.py on
    transformed views of the three sign images the reference ships (classes 14, 1, 27): a classifier whose
    decisions mean something, for whole-attack runs.
    """
    sd = {}
    for name, shape in GTSRB_SHAPES:
        fan_in = int(np.prod(shape[1:]))
        bound = float(np.sqrt(6.0 / fan_in))
        sd[name + ".weight"] = hash_uniform(f"{tag}/{name}.weight", shape, -bound, bound)
        sd[name + ".bias"] = hash_uniform(f"{tag}/{name}.bias", (shape[0],), -0.05, 0.05)
    if readout is not None:
        import os
        f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"{tag}_{readout}_fc7.npz")
        z = np.load(f)
        sd["fc7_1.weight"] = np.ascontiguousarray(z["weight"], dtype=np.float32)
        sd["fc7_1.bias"] = np.ascontiguousarray(z["bias"], dtype=np.float32)
    elif calibrated:
        sd["fc7_1.bias"] = _calibrated_bias(tag)
    return sd


# --------------------------------------------------------------------------
# WideResNet-34-10 (reference cifar/wideresnet.py:52-84)
# --------------------------------------------------------------------------
def wrn_layout(depth: int = 34, widen: int = 10, num_classes: int = 10):
    """List of (state_dict key, shape, kind) in the reference module order."""
    n = (depth - 4) // 6
    ch = [16, 16 * widen, 32 * widen, 64 * widen]
    out = [("conv1.weight", (ch[0], 3, 3, 3), "conv")]

    def bn(prefix, c):
        return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                (prefix + ".running_mean", (c,), "bn_m"), (prefix + ".running_var", (c,), "bn_v"),
                (prefix + ".num_batches_tracked", (), "bn_n")]

    def block(prefix, cin, cout):
        r = []
        for i in range(n):
            ci = cin if i == 0 else cout
            p = f"{prefix}.layer.{i}"
            r += bn(p + ".bn1", ci)
            r += [(p + ".conv1.weight", (cout, ci, 3, 3), "conv")]
            r += bn(p + ".bn2", cout)
            r += [(p + ".conv2.weight", (cout, cout, 3, 3), "conv")]
            if ci != cout:
                r += [(p + ".convShortcut.weight", (cout, ci, 1, 1), "conv")]
        return r

    out += block("block1", ch[0], ch[1])
    out += block("sub_block1", ch[0], ch[1])      # present in the state_dict, unused in forward (:65)
    out += block("block2", ch[1], ch[2])
    out += block("block3", ch[2], ch[3])
    out += bn("bn1", ch[3])
    out += [("fc.weight", (num_classes, ch[3]), "fc_w"), ("fc.bias", (num_classes,), "fc_b")]
    return out


def wrn_state_dict(tag: str = "wrn-v1", depth: int = 34, widen: int = 10, num_classes: int = 10,
                   include_unused: bool = True, calibrated: bool = True):
    sd = {}
    for key, shape, kind in wrn_layout(depth, widen, num_classes):
        if not include_unused and key.startswith("sub_block1"):
            continue
        name = f"{tag}/{key}"
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            b = float(np.sqrt(6.0 / fan_in))
            sd[key] = hash_uniform(name, shape, -b, b)
        elif kind == "bn_w":
            sd[key] = hash_uniform(name, shape, 0.8, 1.2)
        elif kind == "bn_b":
            sd[key] = hash_uniform(name, shape, -0.1, 0.1)
        elif kind == "bn_m":
            sd[key] = hash_uniform(name, shape, -0.1, 0.1)
        elif kind == "bn_v":
            sd[key] = hash_uniform(name, shape, 0.8, 1.2)
        elif kind == "bn_n":
            sd[key] = np.array(1, dtype=np.int64)
        elif kind == "fc_w":
            b = float(np.sqrt(6.0 / shape[1]))
            sd[key] = hash_uniform(name, shape, -b, b)
        elif kind == "fc_b":
            sd[key] = hash_uniform(name, shape, -0.05, 0.05)
    if calibrated and (depth, widen, num_classes) == (34, 10, 10):
        sd["fc.bias"] = _calibrated_bias(tag)
    return sd


# --------------------------------------------------------------------------
# BagNet (PatchGuard's nets/bagnet.py; bagnet17 = layers [3,4,6,3], strides [2,2,2,1], kernel3 [1,1,1,0])
# --------------------------------------------------------------------------
BAGNET17 = dict(layers=(3, 4, 6, 3), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0))


def bagnet_layout(layers=(3, 4, 6, 3), num_classes: int = 10):
    """List of (state_dict key, shape, kind) in module order (the keys of a torch BagNet's state_dict)."""
    def bn(prefix, c):
        return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                (prefix + ".running_mean", (c,), "bn_m"), (prefix + ".running_var", (c,), "bn_v"),
                (prefix + ".num_batches_tracked", (), "bn_n")]

    out = [("conv1.weight", (64, 3, 1, 1), "conv"), ("conv2.weight", (64, 64, 3, 3), "conv")] + bn("bn1", 64)
    inpl = 64
    for g, nblk in enumerate(layers):
        planes = 64 << g
        for i in range(nblk):
            p = f"layer{g + 1}.{i}"
            out += [(p + ".conv1.weight", (planes, inpl, 1, 1), "conv")] + bn(p + ".bn1", planes)
            out += [(p + ".conv2.weight", (planes, planes, None, None), "conv_k")] + bn(p + ".bn2", planes)
            out += [(p + ".conv3.weight", (4 * planes, planes, 1, 1), "conv")] + bn(p + ".bn3", 4 * planes)
            if i == 0:
                out += [(p + ".downsample.0.weight", (4 * planes, inpl, 1, 1), "conv")] + bn(p + ".downsample.1", 4 * planes)
            inpl = 4 * planes
    out += [("fc.weight", (num_classes, inpl), "fc_w"), ("fc.bias", (num_classes,), "fc_b")]
    return out


def bagnet_state_dict(tag: str = "bagnet-v1", layers=(3, 4, 6, 3), kernel3=(1, 1, 1, 0), num_classes: int = 10,
                      calibrated: bool = True):
    """Seeded stand-in for bagnet17_192_cifar.pth (absent, README.md:117).  He-uniform convs; the BatchNorm of every
    block's conv3 (and downsample) is scaled so that the residual sum keeps O(1) activations through 16 blocks."""
    sd = {}
    for key, shape, kind in bagnet_layout(layers, num_classes):
        name = f"{tag}/{key}"
        if kind == "conv_k":
            g, i = int(key[5]) - 1, int(key.split(".")[1])
            k = 3 if kernel3[g] > i else 1
            shape, kind = (shape[0], shape[1], k, k), "conv"
        if kind == "conv":
            b = float(np.sqrt(6.0 / int(np.prod(shape[1:]))))
            sd[key] = hash_uniform(name, shape, -b, b)
        elif kind == "bn_w":
            lo, hi = (0.5, 0.8) if (".bn3." in key or ".downsample.1." in key) else (0.8, 1.2)
            sd[key] = hash_uniform(name, shape, lo, hi)
        elif kind in ("bn_b", "bn_m"):
            sd[key] = hash_uniform(name, shape, -0.1, 0.1)
        elif kind == "bn_v":
            sd[key] = hash_uniform(name, shape, 0.8, 1.2)
        elif kind == "bn_n":
            sd[key] = np.array(1, dtype=np.int64)
        elif kind == "fc_w":
            b = float(np.sqrt(6.0 / shape[1]))
            sd[key] = hash_uniform(name, shape, -b, b)
        elif kind == "fc_b":
            sd[key] = hash_uniform(name, shape, -0.05, 0.05)
    if calibrated and (tuple(layers), tuple(kernel3), num_classes) == ((3, 4, 6, 3), (1, 1, 1, 0), 10):
        sd["fc.bias"] = _calibrated_bias(tag)          # local logits centred per class (tests/golden/make_calibration.py)
    return sd


def state_dict_checksum(sd) -> str:
    """Order-independent 64-bit checksum of a state dict (hex)."""
    acc = 0
    for k in sorted(sd):
        a = np.ascontiguousarray(sd[k])
        h = _fnv1a64(k)
        b = a.view(np.uint8).reshape(-1)
        # fold bytes 8 at a time
        pad = (-b.size) % 8
        if pad:
            b = np.concatenate([b, np.zeros(pad, np.uint8)])
        w = b.view(np.uint64)
        with np.errstate(over="ignore"):
            mix = _splitmix64(w + np.arange(w.size, dtype=np.uint64) + np.uint64(h))
        acc ^= int(np.bitwise_xor.reduce(mix)) if w.size else 0
        acc = (acc * 0x100000001B3 + h) & 0xFFFFFFFFFFFFFFFF
    return f"{acc:016x}"


# --------------------------------------------------------------------------
# Synthetic images / masks
# --------------------------------------------------------------------------
def synth_images(tag: str, n: int, lo: float = -0.5, hi: float = 0.5, size: int = 32) -> np.ndarray:
    """[n,3,size,size] float32 classifier inputs."""
    return hash_uniform(f"{tag}/images", (n, 3, size, size), lo, hi)


def synth_victim(tag: str, size: int) -> np.ndarray:
    """[3,size,size] float32 image in [0,1] quantised to k/255 (as a decoded PNG is)."""
    u = hash_uniform(f"{tag}/victim", (3, size, size), 0.0, 256.0)
    return (np.floor(u).clip(0, 255) / 255.0).astype(np.float32)


def synth_patch_masks(tag: str, n: int, hull: np.ndarray, patch: int = 4, p_keep: float = 0.5) -> np.ndarray:
    """[n,3,S,S] 0/1 masks = hull AND Bernoulli(p_keep) over patch x patch cells
    (SURVEY 8d, config 4)."""
    s = hull.shape[-1]
    g = s // patch
    keep = hash_uniform(f"{tag}/cells", (n, g, g)) < p_keep
    m = np.repeat(np.repeat(keep, patch, axis=1), patch, axis=2).astype(np.float32)
    m = m[:, None, :, :] * hull[None, :1, :, :]
    return np.ascontiguousarray(np.repeat(m, 3, axis=1))

"""ctypes binding of the CPU oracle (oracle/libgraphite_oracle.so).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.  The product package (graphite_amd/) never imports this.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "libgraphite_oracle.so")

XFORM_DTYPE = np.dtype([("M", np.float64, (9,)), ("crop_x", np.int32), ("crop_y", np.int32),
                        ("blur", np.int32), ("pad_", np.int32), ("lut", np.uint8, (256,))], align=True)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


@contextlib.contextmanager
def variant(fma=False, nudge=0):
    """Margin probes (tests/pixel_margin.py): inside the block every call goes to the FMA-contracted build of the same
    source (-ffp-contract=fast) and / or moves each warped value `nudge` ulps before the uint8 truncation."""
    global _lib
    base = lib()
    if fma:
        path = os.path.join(_ORACLE_DIR, "libgraphite_oracle_fma.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        assert L.or_contracted_build() == 1 and base.or_contracted_build() == 0
    else:
        L = base
    L.or_set_warp_nudge(int(nudge))
    _lib = L
    try:
        yield L
    finally:
        L.or_set_warp_nudge(0)
        _lib = base


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def resize_linear(src, dh, dw):
    src, ps = _f(src)
    c, sh, sw = src.shape
    dst = np.empty((c, dh, dw), np.float32)
    _check(lib().or_resize_linear(ps, c, sh, sw, dst.ctypes.data_as(C.c_void_p), dh, dw), "resize")
    return dst


def add_noise(img, mask, theta, lbd=1.0, clip=True):
    img, pi = _f(img); mask, pm = _f(mask); theta, pt = _f(theta)
    _, H, W = img.shape
    _, nh, nw = mask.shape
    comb = np.empty_like(img); pert = np.empty_like(img); ml = np.empty_like(img)
    _check(lib().or_add_noise(pi, H, W, pm, pt, nh, nw, C.c_float(lbd), int(clip),
                              comb.ctypes.data_as(C.c_void_p), pert.ctypes.data_as(C.c_void_p),
                              ml.ctypes.data_as(C.c_void_p)), "add_noise")
    return comb, pert, ml


def composite(img, pert, mask_lg, blur):
    img, pi = _f(img); pert, pp = _f(pert); mask_lg, pm = _f(mask_lg)
    _, H, W = img.shape
    att = np.empty_like(img)
    _check(lib().or_composite(pi, pp, pm, H, W, int(blur), att.ctypes.data_as(C.c_void_p)), "composite")
    return att


def warp_one(att, xf, normalize=True, net=32, want_warped=False):
    att, pa = _f(att)
    _, H, W = att.shape
    xf = np.ascontiguousarray(xf, dtype=XFORM_DTYPE).reshape(1)
    out = np.empty((3, net, net), np.float32)
    warped = np.empty((3, int(xf[0]["crop_y"]), int(xf[0]["crop_x"])), np.float32) if want_warped else None
    _check(lib().or_warp_one(pa, H, W, xf.ctypes.data_as(C.c_void_p), int(normalize), net,
                             out.ctypes.data_as(C.c_void_p),
                             warped.ctypes.data_as(C.c_void_p) if want_warped else None), "warp_one")
    return (out, warped) if want_warped else out


def warp_batch(img, masks, thetas, xf, lbd=1.0, normalize=True, net=32, nthreads=8):
    img, pi = _f(img); masks, pm = _f(masks); thetas, pt = _f(thetas)
    _, H, W = img.shape
    Cn, _, nh, nw = masks.shape
    xf = np.ascontiguousarray(xf, dtype=XFORM_DTYPE)
    T = xf.shape[0]
    out = np.empty((Cn, T, 3, net, net), np.float32)
    _check(lib().or_warp_batch(pi, H, W, pm, pt, Cn, nh, nw, xf.ctypes.data_as(C.c_void_p), T, C.c_float(lbd),
                               int(normalize), net, out.ctypes.data_as(C.c_void_p), nthreads), "warp_batch")
    return out


def net_size(crop_x, crop_y, net):
    nx, ny = C.c_int(), C.c_int()
    lib().or_net_size(int(crop_x), int(crop_y), int(net), C.byref(nx), C.byref(ny))
    return nx.value, ny.value


def warp_batch_var(img, masks, thetas, xf, net, lbd=1.0, nthreads=8):
    """net_size != 32 branch: list (c-major) of [3, ny_t, nx_t] arrays."""
    img, pi = _f(img); masks, pm = _f(masks); thetas, pt = _f(thetas)
    _, H, W = img.shape
    Cn, _, nh, nw = masks.shape
    xf = np.ascontiguousarray(xf, dtype=XFORM_DTYPE)
    T = xf.shape[0]
    sizes = [net_size(int(t["crop_x"]), int(t["crop_y"]), net) for t in xf]
    stride = max(3 * a * b for a, b in sizes)
    out = np.zeros((Cn * T, stride), np.float32)
    _check(lib().or_warp_batch_var(pi, H, W, pm, pt, Cn, nh, nw, xf.ctypes.data_as(C.c_void_p), T, C.c_float(lbd), int(net),
                                   out.ctypes.data_as(C.c_void_p), C.c_longlong(stride), nthreads), "warp_batch_var")
    return [out[i, :3 * sizes[i % T][0] * sizes[i % T][1]].reshape(3, sizes[i % T][1], sizes[i % T][0]).copy()
            for i in range(Cn * T)]


def _tensor_array(tensors):
    keep = [np.ascontiguousarray(t, dtype=np.float32) for t in tensors]
    arr = (C.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
    return keep, arr


def gtsrb_forward(tensors, x, nthreads=8):
    keep, arr = _tensor_array(tensors)
    x, px = _f(x)
    N = x.shape[0]
    logits = np.empty((N, 43), np.float32)
    _check(lib().or_gtsrb_forward(arr, px, N, logits.ctypes.data_as(C.c_void_p), nthreads), "gtsrb_forward")
    return logits


def wrn_forward(tensors, depth, widen, num_classes, x, nthreads=8):
    keep, arr = _tensor_array(tensors)
    x, px = _f(x)
    N = x.shape[0]
    logits = np.empty((N, num_classes), np.float32)
    _check(lib().or_wrn_forward(arr, len(keep), depth, widen, num_classes, px, N,
                                logits.ctypes.data_as(C.c_void_p), nthreads), "wrn_forward")
    return logits


def wrn_forward_masks(tensors, depth, widen, num_classes, x, nthreads=8):
    """(logits, masks uint8 [N, or_wrn_mask_bytes]) -- every ReLU decision of the forward pass."""
    keep, arr = _tensor_array(tensors)
    x, px = _f(x)
    N = x.shape[0]
    lib().or_wrn_mask_bytes.restype = C.c_longlong
    nbytes = int(lib().or_wrn_mask_bytes(depth, widen))
    logits = np.empty((N, num_classes), np.float32)
    masks = np.empty((N, nbytes), np.uint8)
    _check(lib().or_wrn_forward_masks(arr, len(keep), depth, widen, num_classes, px, N, logits.ctypes.data_as(C.c_void_p),
                                      nthreads, masks.ctypes.data_as(C.c_void_p)), "wrn_forward_masks")
    return logits, masks


def argmax_hits(logits, T, target):
    logits, pl = _f(logits)
    N, ncls = logits.shape
    preds = np.empty(N, np.int64)
    hits = np.zeros((N + T - 1) // T, np.int32)
    _check(lib().or_argmax_hits(pl, N, ncls, T, int(target), preds.ctypes.data_as(C.c_void_p),
                                hits.ctypes.data_as(C.c_void_p)), "argmax_hits")
    return preds, hits


# ---- PatchGuard robust classifier (patchguard/patchguard_predict.py:21-53) ----
def pg_preprocess(x, size=192, nthreads=8):
    x, px = _f(x)
    N, _, S, _ = x.shape
    out = np.empty((N, 3, size, size), np.float32)
    _check(lib().or_pg_preprocess(px, N, S, size, out.ctypes.data_as(C.c_void_p), nthreads), "pg_preprocess")
    return out


def _i4(v):
    return (C.c_int * 4)(*[int(a) for a in v])


def bagnet_out_size(D, layers, strides, kernel3):
    return int(lib().or_bagnet_out_size(int(D), _i4(layers), _i4(strides), _i4(kernel3)))


def bagnet_forward(tensors, z, layers=(3, 4, 6, 3), strides=(2, 2, 2, 1), kernel3=(1, 1, 1, 0), num_classes=10, nthreads=8):
    """z [N,3,D,D] (normalised) -> local logits [N,h,h,num_classes]."""
    keep, arr = _tensor_array(tensors)
    z, pz = _f(z)
    N, _, D, _ = z.shape
    h = bagnet_out_size(D, layers, strides, kernel3)
    local = np.empty((N, h, h, num_classes), np.float32)
    _check(lib().or_bagnet_forward(arr, len(keep), _i4(layers), _i4(strides), _i4(kernel3), num_classes, pz, N, D,
                                   local.ctypes.data_as(C.c_void_p), nthreads), "bagnet_forward")
    return local


def masking_defense(local, window=6, thres=0.0):
    local, pl = _f(local)
    N, h, w, ncls = local.shape
    scores = np.empty((N, ncls), np.float32)
    preds = np.empty(N, np.int64)
    _check(lib().or_masking_defense(pl, N, h, w, ncls, window, window, C.c_float(thres), scores.ctypes.data_as(C.c_void_p),
                                    preds.ctypes.data_as(C.c_void_p)), "masking_defense")
    return scores, preds


def pg_mean_logits(local):
    local, pl = _f(local)
    N, h, w, ncls = local.shape
    scores = np.empty((N, ncls), np.float32)
    _check(lib().or_pg_mean_logits(pl, N, h, w, ncls, scores.ctypes.data_as(C.c_void_p)), "pg_mean_logits")
    return scores

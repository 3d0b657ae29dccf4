"""ctypes binding of libgraphite_hip.so (include/graphite_hip.h).

PyTorch-ROCm is used for device memory and streams only: every tensor handed to
the library is a contiguous CUDA(ROCm) tensor whose `data_ptr()` crosses the C
ABI together with `torch.cuda.current_stream().cuda_stream`.

There is NO CPU fallback: if the shared library is missing, or no gfx950 device
is visible, the first call raises.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgraphite_hip.so")

XFORM_DTYPE = np.dtype([("M", np.float64, (9,)), ("crop_x", np.int32), ("crop_y", np.int32),
                        ("blur", np.int32), ("pad_", np.int32), ("lut", np.uint8, (256,))], align=True)

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); also the list checked by tests/test_abi.py against the header
SIGNATURES = {
    "gr_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "gr_ctx_destroy": (None, [_vp]),
    "gr_last_error": (C.c_char_p, [_vp]),
    "gr_ctx_set_chunk": (_i, [_vp, _i]),
    "gr_ctx_workspace_bytes": (_i64, [_vp]),
    "gr_ctx_set_workspace_limit": (_i, [_vp, _i64]),
    "gr_profile_enable": (_i, [_vp, _i]),
    "gr_profile_reset": (_i, [_vp]),
    "gr_profile_read": (_i, [_vp, _i, _vp]),
    "gr_xforms_create": (_i, [_vp, _vp, _i, C.POINTER(_vp)]),
    "gr_xforms_destroy": (None, [_vp]),
    "gr_resize_linear": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "gr_convert_to_network": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "gr_add_noise": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "gr_warp_perspective": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "gr_transform_composited": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "gr_net_size": (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "gr_transform_composited_var": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i64, _vp]),
    "gr_warp_batch_var": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _f, _i, _vp, _i64, _vp]),
    "gr_transform_composited_static": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "gr_warp_batch": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _f, _i, _i, _vp, _vp]),
    "gr_model_load_gtsrb": (_i, [_vp, _vp, _vp, _i, C.POINTER(_vp)]),
    "gr_model_load_wrn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gr_model_destroy": (None, [_vp]),
    "gr_model_num_classes": (_i, [_vp]),
    "gr_model_set_precision": (_i, [_vp, _i]),
    "gr_model_get_precision": (_i, [_vp]),
    "gr_model_set_bf16_layers": (_i, [_vp, C.c_uint]),
    "gr_model_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "gr_model_predict": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "gr_pg_preprocess": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "gr_model_load_bagnet": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, C.POINTER(_vp)]),
    "gr_pg_configure": (_i, [_vp, _i, _i, _f]),
    "gr_pg_out_size": (_i, [_vp]),
    "gr_pg_local_logits": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "gr_pg_masking_defense": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "gr_model_input_grad": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gr_tr_hits": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _f, _i, _i, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gr_wb_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "gr_wb_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
}

WB_XFORM_DTYPE = np.dtype([("A", np.float32, (9,)), ("crop_x", np.int32), ("crop_y", np.int32),
                           ("gamma", np.float32), ("blur", np.int32)], align=True)

_lib = None
_lock = threading.Lock()
_ctxs = {}


class GraphiteHipError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree library and declare every entry point.  Needs no GPU."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GraphiteHipError(
                    f"{LIB_PATH} not found: build it with `make -C graphite_amd/csrc` "
                    "(or __graft_entry__.build()); there is no CPU fallback")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def _check(ctx, rc, what):
    if rc != 0:
        msg = load_library().gr_last_error(ctx)
        raise GraphiteHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def device_index():
    if not torch.cuda.is_available():
        raise GraphiteHipError("no ROCm device visible to PyTorch; graphite_amd has no CPU path")
    return torch.cuda.current_device()


def ctx():
    """One gr_ctx per device, created on first use."""
    lib = load_library()
    dev = device_index()
    with _lock:
        c = _ctxs.get(dev)
        if c is None:
            h = _vp()
            rc = lib.gr_ctx_create(dev, C.byref(h))
            if rc != 0:
                raise GraphiteHipError(f"gr_ctx_create({dev}) failed ({rc}): {lib.gr_last_error(None).decode()}")
            c = _ctxs[dev] = h
    return c


def set_chunk(images_per_chunk):
    _check(ctx(), load_library().gr_ctx_set_chunk(ctx(), int(images_per_chunk)), "gr_ctx_set_chunk")


def set_workspace_limit(nbytes):
    _check(ctx(), load_library().gr_ctx_set_workspace_limit(ctx(), int(nbytes)), "gr_ctx_set_workspace_limit")


def workspace_bytes():
    return int(load_library().gr_ctx_workspace_bytes(ctx()))


class _ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


def profile_enable(on=True):
    _check(ctx(), load_library().gr_profile_enable(ctx(), int(on)), "gr_profile_enable")


def profile_reset():
    _check(ctx(), load_library().gr_profile_reset(ctx()), "gr_profile_reset")


def profile_read():
    """{kernel name: dict(launches, ms, flops, bytes)} accumulated since the last reset."""
    buf = (_ProfEntry * 64)()
    n = load_library().gr_profile_read(ctx(), 64, C.cast(buf, _vp))
    if n < 0:
        _check(ctx(), n, "gr_profile_read")
    return {buf[i].name.decode(): dict(launches=int(buf[i].launches), ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
            for i in range(n)}


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def to_dev(a):
    """float32 contiguous CUDA tensor from numpy / CPU tensor / CUDA tensor."""
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    elif a.is_cuda and a.dtype == torch.float32 and a.is_contiguous() and a.device.index == torch.cuda.current_device():
        return a                                    # already in place: the evaluator's calls pass through here twice
    return a.to(device=f"cuda:{device_index()}", dtype=torch.float32).contiguous()


def _p(t):
    return _vp(t.data_ptr()) if t is not None else None


class XformTable:
    """Device copy of a transform table (gr_xforms)."""

    def __init__(self, tab):
        tab = np.ascontiguousarray(tab, dtype=XFORM_DTYPE)
        self.T = int(tab.shape[0])
        self.host = tab
        self._h = _vp()
        self._ctx = ctx()
        _check(self._ctx, load_library().gr_xforms_create(self._ctx, tab.ctypes.data_as(_vp), self.T, C.byref(self._h)),
               "gr_xforms_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gr_xforms_destroy(h)


_xf_cache = {}


def xform_table(tab):
    """Cached XformTable: an attack evaluates ~1,700 candidates against one transform list."""
    if isinstance(tab, XformTable):
        return tab
    tab = np.ascontiguousarray(tab, dtype=XFORM_DTYPE)
    key = (device_index(), hash(tab.tobytes()))
    x = _xf_cache.get(key)
    if x is None:
        if len(_xf_cache) > 16:
            _xf_cache.clear()
        x = _xf_cache[key] = XformTable(tab)
    return x


# ------------------------------------------------------------------------------------------------
# pixel operators (numpy in, numpy out -- the reference's functions take and return CPU data)
# ------------------------------------------------------------------------------------------------
def resize_linear(src, dh, dw):
    s = to_dev(src)
    c, sh, sw = s.shape
    d = torch.empty((c, dh, dw), device=s.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_resize_linear(ctx(), _p(s), c, sh, sw, _p(d), dh, dw, _stream()), "gr_resize_linear")
    return d.cpu().numpy()


def net_size(crop_x, crop_y, net):
    """(nx, ny) of convert2Network's output for a crop_x x crop_y image (gr_net_size)."""
    nx, ny = _i(), _i()
    rc = load_library().gr_net_size(int(crop_x), int(crop_y), int(net), C.byref(nx), C.byref(ny))
    if rc != 0:
        raise GraphiteHipError("gr_net_size: bad arguments")
    return nx.value, ny.value


def _split_var(out, tab, net):
    """[N, stride] device buffer -> list of CPU arrays [3, ny_t, nx_t] (one per image, transform = index % T)."""
    sizes = [net_size(int(t["crop_x"]), int(t["crop_y"]), net) for t in tab]
    host = out.cpu().numpy()
    T = len(tab)
    return [host[i, :3 * sizes[i % T][0] * sizes[i % T][1]].reshape(3, sizes[i % T][1], sizes[i % T][0]).copy()
            for i in range(host.shape[0])]


def _var_stride(tab, net):
    return max(3 * nx * ny for nx, ny in (net_size(int(t["crop_x"]), int(t["crop_y"]), net) for t in tab))


def transform_composited_var(img, mask_lg, pert, tab, net):
    """apply_transformation's net_size != 32 branch for every transform of `tab`: list of [3, ny_t, nx_t] arrays."""
    i, m, p = to_dev(img), to_dev(mask_lg), to_dev(pert)
    _, H, W = i.shape
    xf = xform_table(tab)
    stride = _var_stride(xf.host, net)
    out = torch.zeros((xf.T, stride), device=i.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_transform_composited_var(ctx(), _p(i), _p(m), _p(p), H, W, xf._h, net, _p(out), stride,
                                                             _stream()), "gr_transform_composited_var")
    return _split_var(out, xf.host, net)


def warp_batch_var(img, masks, thetas, tab, lbd, net):
    """get_transformed_images' net_size != 32 branch for C candidates: list (c-major) of [3, ny_t, nx_t] arrays."""
    i, m, t = to_dev(img), to_dev(masks), to_dev(thetas)
    _, H, W = i.shape
    Cn, _, nh, nw = m.shape
    xf = xform_table(tab)
    stride = _var_stride(xf.host, net)
    out = torch.zeros((Cn * xf.T, stride), device=i.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_warp_batch_var(ctx(), _p(i), H, W, _p(m), _p(t), Cn, nh, nw, xf._h, lbd, net, _p(out),
                                                   stride, _stream()), "gr_warp_batch_var")
    return _split_var(out, xf.host, net)


def convert_to_network(src, net, normalize):
    s = to_dev(src)
    _, sh, sw = s.shape
    nx, ny = net_size(sw, sh, net)
    d = torch.empty((3, ny, nx), device=s.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_convert_to_network(ctx(), _p(s), sh, sw, _p(d), net, int(normalize), _stream()),
           "gr_convert_to_network")
    return d.cpu().numpy()


def add_noise(img, mask, theta, lbd, clip):
    i, m, t = to_dev(img), to_dev(mask), to_dev(theta)
    _, H, W = i.shape
    _, nh, nw = m.shape
    comb, pert, ml = torch.empty_like(i), torch.empty_like(i), torch.empty_like(i)
    _check(ctx(), load_library().gr_add_noise(ctx(), _p(i), H, W, _p(m), _p(t), nh, nw, lbd, int(clip), _p(comb), _p(pert),
                                              _p(ml), _stream()), "gr_add_noise")
    return comb.cpu().numpy(), pert.cpu().numpy(), ml.cpu().numpy()


def warp_perspective_hwc(img_hwc, M, crop_x, crop_y):
    s = to_dev(np.ascontiguousarray(np.transpose(img_hwc, (2, 0, 1))))
    _, H, W = s.shape
    Mh = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
    d = torch.empty((3, crop_y, crop_x), device=s.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_warp_perspective(ctx(), _p(s), H, W, Mh.ctypes.data_as(_vp), crop_x, crop_y, _p(d),
                                                     _stream()), "gr_warp_perspective")
    return np.ascontiguousarray(np.transpose(d.cpu().numpy(), (1, 2, 0)))


def transform_composited(img, mask_lg, pert, tab, net, normalize):
    i, m, p = to_dev(img), to_dev(mask_lg), to_dev(pert)
    _, H, W = i.shape
    xf = xform_table(tab)
    out = torch.empty((xf.T, 3, net, net), device=i.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_transform_composited(ctx(), _p(i), _p(m), _p(p), H, W, xf._h, int(normalize), net,
                                                         _p(out), _stream()), "gr_transform_composited")
    return out.cpu().numpy()


def transform_composited_static(img_d, mask_lg_d, pert_d, xf, out_d, normalize=True, net=32):
    """gr_transform_composited_static: device tensors in, `out_d` [T,3,net,net] preallocated by the caller; the library
    allocates nothing (composites live in its 4 MiB static device array)."""
    _, H, W = img_d.shape
    _check(ctx(), load_library().gr_transform_composited_static(ctx(), _p(img_d), _p(mask_lg_d), _p(pert_d), H, W, xf._h,
                                                                int(normalize), net, _p(out_d), _stream()),
           "gr_transform_composited_static")
    return out_d


def warp_batch_dev(img, masks, thetas, tab, lbd, normalize, net=32):
    """Device-resident get_transformed_images for C candidates: returns a CUDA tensor [C,T,3,net,net]."""
    i, m, t = to_dev(img), to_dev(masks), to_dev(thetas)
    _, H, W = i.shape
    Cn, _, nh, nw = m.shape
    assert t.shape == m.shape
    xf = xform_table(tab)
    out = torch.empty((Cn, xf.T, 3, net, net), device=i.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_warp_batch(ctx(), _p(i), H, W, _p(m), _p(t), Cn, nh, nw, xf._h, lbd, int(normalize), net,
                                               _p(out), _stream()), "gr_warp_batch")
    return out


def warp_batch(img, masks, thetas, tab, lbd, normalize, net=32):
    return warp_batch_dev(img, masks, thetas, tab, lbd, normalize, net).cpu().numpy()


# ------------------------------------------------------------------------------------------------
# classifiers
# ------------------------------------------------------------------------------------------------
class Model:
    """gr_model handle.  `kind` = 'gtsrb' (20 tensors), 'wrn' (see weights.wrn_tensor_list) or 'bagnet' (the PatchGuard
    classifier: weights.bagnet_tensor_list; layers / strides / kernel3 / upsample as gr_model_load_bagnet)."""

    def __init__(self, kind, tensors, depth=34, widen=10, num_classes=10, layers=(3, 4, 6, 3), strides=(2, 2, 2, 1),
                 kernel3=(1, 1, 1, 0), upsample=192):
        lib = load_library()
        self._ctx = ctx()
        keep = [np.ascontiguousarray(t, dtype=np.float32) for t in tensors]
        ptrs = (_vp * len(keep))(*[t.ctypes.data for t in keep])
        numel = (C.c_int64 * len(keep))(*[t.size for t in keep])
        self._h = _vp()
        self.kind = kind
        if kind == "gtsrb":
            rc = lib.gr_model_load_gtsrb(self._ctx, ptrs, numel, len(keep), C.byref(self._h))
        elif kind == "wrn":
            rc = lib.gr_model_load_wrn(self._ctx, ptrs, numel, len(keep), depth, widen, num_classes, C.byref(self._h))
        elif kind == "bagnet":
            i4 = [(C.c_int * 4)(*[int(v) for v in a]) for a in (layers, strides, kernel3)]
            rc = lib.gr_model_load_bagnet(self._ctx, ptrs, numel, len(keep), i4[0], i4[1], i4[2], num_classes, int(upsample),
                                          C.byref(self._h))
        else:
            raise ValueError(kind)
        _check(self._ctx, rc, f"gr_model_load_{kind}")
        self.num_classes = int(lib.gr_model_num_classes(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gr_model_destroy(h)

    PRECISIONS = {"f32": 0, "bf16": 1}                    # GR_PREC_F32, GR_PREC_BF16

    def set_precision(self, precision):
        """'f32' (default; the parity path) or 'bf16' (conv / linear layers on bf16 operands, f32 accumulation; see gr_model_set_precision)."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        _check(self._ctx, load_library().gr_model_set_precision(self._h, self.PRECISIONS[precision]), "gr_model_set_precision")

    def set_bf16_layers(self, mask):
        """bit 0 = conv2_1 .. 5 = conv4_2, 6 = fc5 + fc6: the layers that 'bf16' precision runs in bf16 (default 0x7f)."""
        _check(self._ctx, load_library().gr_model_set_bf16_layers(self._h, int(mask)), "gr_model_set_bf16_layers")

    @property
    def precision(self):
        v = int(load_library().gr_model_get_precision(self._h))
        return {0: "f32", 1: "bf16"}[v]

    def input_grad(self, x, target):
        """(logits [N,ncls], d CE(logits_n, target) / d x_n [N,3,32,32]) as CUDA tensors (GTSRBNet and WideResNet)."""
        xd = to_dev(x)
        if xd.dim() == 3:
            xd = xd[None]
        n = xd.shape[0]
        logits = torch.empty((n, self.num_classes), device=xd.device, dtype=torch.float32)
        gx = torch.empty_like(xd)
        _check(self._ctx, load_library().gr_model_input_grad(self._ctx, self._h, _p(xd), n, int(target), _p(logits), _p(gx),
                                                             _stream()), "gr_model_input_grad")
        return logits, gx

    def forward(self, x):
        """x: [N,3,32,32] (any device) -> logits CUDA tensor [N,ncls]."""
        xd = to_dev(x)
        n = xd.shape[0]
        out = torch.empty((n, self.num_classes), device=xd.device, dtype=torch.float32)
        _check(self._ctx, load_library().gr_model_forward(self._ctx, self._h, _p(xd), n, _p(out), _stream()),
               "gr_model_forward")
        return out

    # ---- PatchGuard models only ----
    def pg_configure(self, defense=True, window=6, thres=0.0):
        _check(self._ctx, load_library().gr_pg_configure(self._h, int(bool(defense)), int(window), float(thres)), "gr_pg_configure")

    @property
    def pg_out_size(self):
        return int(load_library().gr_pg_out_size(self._h))

    def pg_local_logits(self, x):
        """x [N,3,32,32] in [-0.5,0.5] -> local logits CUDA tensor [N,h,h,ncls] (patchguard_predict.py:42-44)."""
        xd = to_dev(x)
        n, h = xd.shape[0], self.pg_out_size
        out = torch.empty((n, h, h, self.num_classes), device=xd.device, dtype=torch.float32)
        _check(self._ctx, load_library().gr_pg_local_logits(self._ctx, self._h, _p(xd), n, _p(out), _stream()), "gr_pg_local_logits")
        return out

    def predict_dev(self, x):
        xd = to_dev(x)
        n = xd.shape[0]
        out = torch.empty((n,), device=xd.device, dtype=torch.int64)
        _check(self._ctx, load_library().gr_model_predict(self._ctx, self._h, _p(xd), n, _p(out), _stream()),
               "gr_model_predict")
        return out


def pg_preprocess(x, size=192):
    """patchguard_predict.py:42-44 + :30-32: x [N,3,S,S] in [-0.5,0.5] -> normalised CUDA tensor [N,3,size,size]."""
    xd = to_dev(x)
    n, _, s, s2 = xd.shape
    assert s == s2
    out = torch.empty((n, 3, size, size), device=xd.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_pg_preprocess(ctx(), _p(xd), n, s, int(size), _p(out), _stream()), "gr_pg_preprocess")
    return out


def pg_masking_defense(local, window=6, thres=0.0):
    """masking_defense (patchguard_predict.py:48) for a batch of local logits [N,h,w,ncls] -> (scores [N,ncls], preds [N])."""
    ld = to_dev(local)
    n, h, w, ncls = ld.shape
    scores = torch.empty((n, ncls), device=ld.device, dtype=torch.float32)
    preds = torch.empty((n,), device=ld.device, dtype=torch.int64)
    _check(ctx(), load_library().gr_pg_masking_defense(ctx(), _p(ld), n, h, w, ncls, int(window), float(thres), _p(scores),
                                                       _p(preds), _stream()), "gr_pg_masking_defense")
    return scores, preds


def tr_hits_dev(model, img, masks, thetas, tab, lbd, normalize, target, img_begin=0, img_count=None,
                want_preds=False, want_logits=False):
    """Fused TR evaluation; everything stays on the GPU.  Returns (hits int32[C] CUDA, preds|None, logits|None)."""
    i, m, t = to_dev(img), to_dev(masks), to_dev(thetas)
    _, H, W = i.shape
    Cn, _, nh, nw = m.shape
    assert t.shape == m.shape
    xf = xform_table(tab)
    total = Cn * xf.T
    if img_count is None:
        img_count = total - img_begin
    hits = torch.empty((Cn,), device=i.device, dtype=torch.int32)
    preds = torch.full((total,), -1, device=i.device, dtype=torch.int64) if want_preds else None
    logits = torch.zeros((total, model.num_classes), device=i.device, dtype=torch.float32) if want_logits else None
    _check(model._ctx, load_library().gr_tr_hits(model._ctx, model._h, _p(i), H, W, _p(m), _p(t), Cn, nh, nw, xf._h, lbd,
                                                 int(normalize), int(target), int(img_begin), int(img_count), _p(hits),
                                                 _p(preds), _p(logits), _stream()), "gr_tr_hits")
    return hits, preds, logits


# ------------------------------------------------------------------------------------------------
# differentiable ("whitebox") transform
# ------------------------------------------------------------------------------------------------
def wb_forward(orig, att, mask, tab, net=32):
    """[T,3,net,net] CUDA tensor: transform_wb of (orig, att, mask) under every descriptor of `tab` (WB_XFORM_DTYPE)."""
    o, a, m = to_dev(orig), to_dev(att), to_dev(mask)
    tab = np.ascontiguousarray(tab, dtype=WB_XFORM_DTYPE)
    _, H, W = o.shape
    assert a.shape == o.shape == m.shape
    out = torch.empty((len(tab), 3, net, net), device=o.device, dtype=torch.float32)
    _check(ctx(), load_library().gr_wb_forward(ctx(), _p(o), _p(a), _p(m), H, W, tab.ctypes.data_as(_vp), len(tab), net,
                                               _p(out), _stream()), "gr_wb_forward")
    return out


def wb_backward(orig, att, mask, tab, grad_out, net=32):
    """d loss / d att [3,H,W] (summed over the T transforms) given d loss / d out [T,3,net,net]."""
    o, a, m, g = to_dev(orig), to_dev(att), to_dev(mask), to_dev(grad_out)
    tab = np.ascontiguousarray(tab, dtype=WB_XFORM_DTYPE)
    _, H, W = o.shape
    assert g.shape == (len(tab), 3, net, net)
    ga = torch.empty_like(a)
    _check(ctx(), load_library().gr_wb_backward(ctx(), _p(o), _p(a), _p(m), H, W, tab.ctypes.data_as(_vp), len(tab), net,
                                                _p(g), _p(ga), _stream()), "gr_wb_backward")
    return ga

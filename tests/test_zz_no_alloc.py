"""gr_transform_composited_static (the entry point that allocates nothing, include/graphite_hip.h) against
gr_transform_composited: same outputs bit for bit, on two calls in a row, the second with a smaller input.
Collected last on purpose (file name): a failure here must not hide the other GPU tests under `pytest -x`."""
import numpy as np
import pytest
import torch

from cases import gtsrb_case, gtsrb_candidates
from graphite_amd import _hip

pytestmark = pytest.mark.gpu


def _triple(case, cand, H, W):
    """(img, mask_lg, pert) device tensors of one candidate, cropped to an HxW frame (add_noise through the C ABI)."""
    masks, thetas = gtsrb_candidates(case, cand + 1)
    _, pert, ml = _hip.add_noise(case["vic"], masks[cand], thetas[cand], 1.0, True)
    cut = lambda a: np.ascontiguousarray(a[:, :H, :W])
    return _hip.to_dev(cut(case["vic"])), _hip.to_dev(cut(ml)), _hip.to_dev(cut(pert))


def test_static_entry_matches_workspace_entry_twice():
    case = gtsrb_case()
    lib, ctx = _hip.load_library(), _hip.ctx()
    ws_before = None
    # first call: the full 244x244 frame, 100 transforms, all four blur kernels;
    # second call: a SMALLER frame and fewer transforms (stale composites of the first call stay in the array)
    for (cand, H, W, tab) in ((0, 244, 244, case["tab"]), (2, 120, 200, case["tab"][10:37])):
        img, ml, pert = _triple(case, cand, H, W)
        xf = _hip.xform_table(tab)
        want = torch.empty((xf.T, 3, 32, 32), device=img.device)
        rc = lib.gr_transform_composited(ctx, _hip._p(img), _hip._p(ml), _hip._p(pert), H, W, xf._h, 1, 32, _hip._p(want),
                                         _hip._stream())
        assert rc == 0, lib.gr_last_error(ctx)
        torch.cuda.synchronize()
        if ws_before is None:
            ws_before = _hip.workspace_bytes()
        got = torch.full((xf.T, 3, 32, 32), 7.0, device=img.device)
        _hip.transform_composited_static(img, ml, pert, xf, got)
        torch.cuda.synchronize()
        assert _hip.workspace_bytes() == ws_before                # the library's workspace did not grow
        assert torch.equal(got, want), f"{int((got != want).sum())} values differ"
        assert float(got.min()) >= -0.5 and float(got.max()) <= 0.5 and float(got.std()) > 0.01


def test_static_entry_rejects_oversize_and_profiling():
    case = gtsrb_case()
    lib, ctx = _hip.load_library(), _hip.ctx()
    xf = _hip.xform_table(case["tab"][:4])
    big = torch.zeros((3, 300, 300), device="cuda")
    out = torch.zeros((4, 3, 32, 32), device="cuda")
    with pytest.raises(_hip.GraphiteHipError, match="exceeds the static scratch"):
        _hip.transform_composited_static(big, big, big, xf, out)
    img, ml, pert = _triple(case, 0, 244, 244)
    _hip.profile_enable(True)
    try:
        with pytest.raises(_hip.GraphiteHipError, match="profiling"):
            _hip.transform_composited_static(img, ml, pert, xf, out)
    finally:
        _hip.profile_enable(False)
    _hip.transform_composited_static(img, ml, pert, xf, out)      # still works afterwards
    torch.cuda.synchronize()
    assert float(out.abs().max()) > 0

"""Pinhole-camera homography and crop geometry (host side, float64).

Drop-in for the reference's `perspective_transform.py`:
  * `get_offset_and_crop_size`  -- reference perspective_transform.py:54-114
  * `get_perspective_transform` -- reference perspective_transform.py:8-51
plus `perspective_matrix`, which stops before the pixel work and returns the
`(M, crop_x, crop_y)` that the reference hands to `cv2.warpPerspective`; the
batched HIP path uploads those instead of warping one image at a time.

The geometry is host logic in the reference too (numpy float64, a few dozen
flops per transform); it stays on the host.  Values are pinned bit-for-bit
against the reference in tests/golden/homography.npz, which is why the 4x4
products go through `np.matmul` exactly as the reference's do.
"""
import csv
import functools
import math

import numpy as np


@functools.lru_cache(maxsize=64)
def _read_points(pt_file):
    """Rows `x,y,scale` of the object outline (e.g. 8 octagon corners at scale 300).
    The reference re-reads the CSV on every transform (:57-60); it is cached here."""
    rows = []
    with open(pt_file) as f:
        for row in csv.reader(f):
            if not row:
                continue
            rows.append((float(row[0]), float(row[1]), float(row[2])))
    return tuple(rows)


def _object_points(w, h, pt_file):
    if pt_file is not None and pt_file != '':
        pts = []
        for (x, y, s) in _read_points(pt_file):
            pts.append(np.array([[x * (w / s)], [y * (h / s)], [s * (1.0 / s)]]))
        return pts, True
    return [np.array([[0], [0], [1.0]]), np.array([[0], [h], [1.0]]),
            np.array([[w], [0], [1.0]]), np.array([[w], [h], [1.0]])], False


def get_offset_and_crop_size(w, h, H, crop_percent, crop_off_x, crop_off_y, pt_file, ratio):
    """Bounding box of the projected object points -> (x_off, y_off, crop_size).

    With an outline file the box is padded to the frame's aspect ratio and the
    offsets are rounded with Python's round-half-even (:91-106); without one
    (CIFAR) the frame corners are used with float floor division (:108-114).
    """
    pts, has_file = _object_points(w, h, pt_file)
    min_x, min_y, max_x, max_y = w, h, 0, 0
    for pt in pts:
        q = np.matmul(H, pt)
        q /= q[2, 0]
        qx, qy = q[0, 0], q[1, 0]
        if qx < min_x:
            min_x = qx
        if qx > max_x:
            max_x = qx
        if qy < min_y:
            min_y = qy
        if qy > max_y:
            max_y = qy

    if has_file:
        bw, bh = max_x - min_x, max_y - min_y
        if bw / bh < w / h:                       # projected box is tall and narrow
            diff_in_size = bh / h * w - bw
            orig_size = bh if w > h else bh / h * w
            crop_size = int(round(orig_size * (1.0 - crop_percent)))
            y_off = -min_y - int(round(crop_percent / 2 * orig_size))
            x_off = -min_x + int(round(diff_in_size / 2 - crop_percent / 2 * orig_size))
        else:                                     # wide and short
            diff_in_size = bw / w * h - bh
            orig_size = bw if h > w else bw / w * h
            crop_size = int(round(orig_size * (1.0 - crop_percent)))
            x_off = -min_x - int(round(crop_percent / 2 * orig_size))
            y_off = -min_y + int(round(diff_in_size / 2 - crop_percent / 2 * orig_size))
        return x_off + crop_off_x * crop_size, y_off + crop_off_y * crop_size, crop_size

    min_x -= (w * ratio - (max_x - min_x)) // 2
    min_y -= (h * ratio - (max_y - min_y)) // 2
    crop_size = int(round((1.0 - crop_percent) * min(w, h) * ratio))
    return (-min_x - int(round(crop_percent / 2 * w * ratio)),
            -min_y - int(round(crop_percent / 2 * h * ratio)), crop_size)


def perspective_matrix(angle, w, h, f, d, crop_percent, crop_off_x, crop_off_y, pt_file=None):
    """(M float64[3,3] forward src->dst, crop_x, crop_y) -- reference :11-44."""
    a = math.radians(angle)
    sa, ca = math.sin(a), math.cos(a)
    x_cam, y_cam, z_cam = w / 2 - sa * d, h / 2, -ca * d
    R = np.array([[ca, 0, -sa, 0], [0, 1, 0, 0], [sa, 0, ca, 0], [0, 0, 0, 1]])
    C = np.array([[1, 0, 0, -x_cam], [0, 1, 0, -y_cam], [0, 0, 1, -z_cam], [0, 0, 0, 1]])
    RT = np.matmul(R, C)
    H = np.array([[f * RT[0, 0], f * RT[0, 1], f * RT[0, 3]],
                  [f * RT[1, 0], f * RT[1, 1], f * RT[1, 3]],
                  [RT[2, 0], RT[2, 1], RT[2, 3]]])
    x_off, y_off, crop_size = get_offset_and_crop_size(w, h, H, crop_percent, crop_off_x, crop_off_y,
                                                       pt_file, f / d)
    M = np.matmul(np.array([[1, 0, x_off], [0, 1, y_off], [0, 0, 1]]), H)
    if h > w:
        crop_x, crop_y = crop_size, int(round(crop_size / w * h))
    else:
        crop_y, crop_x = crop_size, int(round(crop_size / h * w))
    return M, crop_x, crop_y


def get_perspective_transform(img, angle, w, h, f, d, crop_percent, crop_off_x, crop_off_y,
                              pt_file='inputs/GTSRB/Points/14.csv', whitebox=False):
    """Warp an HWC float32 numpy image as the reference does (:8-51); returns HWC [crop_y,crop_x,3].

    The pixel work runs in the HIP library (gr_warp_perspective).  `whitebox=True` (the kornia branch, :49) is
    not offered as a stand-alone warp: the differentiable path is fused end to end in
    graphite_amd.transforms.transform_wb / transform_wb_batch (gr_wb_forward / gr_wb_backward), which never
    materialise the warped crop.
    """
    if whitebox:
        raise NotImplementedError("use graphite_amd.transforms.transform_wb: the differentiable warp is fused with "
                                  "gamma and resize and has no stand-alone form")
    from . import _hip
    M, crop_x, crop_y = perspective_matrix(angle, w, h, f, d, crop_percent, crop_off_x, crop_off_y, pt_file)
    return _hip.warp_perspective_hwc(np.ascontiguousarray(img, dtype=np.float32), M, crop_x, crop_y)

"""Image / mask / outline-file loading without OpenCV (reference main.py:35-67 uses cv2.imread).

PNG decode goes through Pillow; as `cv2.imread(path)` does, an alpha channel is
dropped (not composited) and the result is 3-channel 8-bit.  Returned arrays are
CHW float32 in [0,1] with RGB channel order (the reference converts BGR->RGB
right after imread, main.py:36).
"""
import numpy as np
from PIL import Image


def load_image_u8_hwc(path):
    im = Image.open(path)
    if im.mode == "RGBA":
        im = Image.fromarray(np.asarray(im)[:, :, :3])      # drop alpha, keep colour as stored
    elif im.mode != "RGB":
        im = im.convert("RGB")
    return np.asarray(im, dtype=np.uint8)


def load_image_chw(path):
    return np.ascontiguousarray(load_image_u8_hwc(path).astype(np.float32).transpose(2, 0, 1) / 255.0)


def resize_u8_linear(img_hwc, dh, dw):
    """cv2.resize(uint8 HWC, (dw, dh)) INTER_LINEAR -- the fixed-point 8-bit path the reference uses for
    the target image and a non-32x32 hull (main.py:49,59).  [OpenCV-recalled, unpinned]: coefficients
    are rounded to 11 fractional bits, the horizontal pass keeps 32-bit sums, the vertical pass is
    ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Host-side input preparation, as in the reference."""
    src = np.asarray(img_hwc, dtype=np.uint8)
    sh, sw = src.shape[:2]
    if (sh, sw) == (dh, dw):
        return src.copy()
    if (sh, sw) == (2 * dh, 2 * dw):
        # exact 2x2 reduction: cv::resize takes INTER_AREA's fast path, (S00 + S01 + S10 + S11 + 2) >> 2 for 8-bit pixels
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(ssize, dsize, horizontal):
        scale = 1.0 / (float(dsize) / float(ssize))
        d = np.arange(dsize)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if horizontal:
            f = np.where(s < 0, 0, f); s = np.maximum(s, 0)
            f = np.where(s >= ssize - 1, 0, f); s = np.minimum(s, ssize - 1)
            s1 = np.minimum(s + 1, ssize - 1)
        else:
            s1 = np.clip(s + 1, 0, ssize - 1)
            s = np.clip(s, 0, ssize - 1)
        w0 = np.clip(np.rint((1.0 - f) * 2048.0), -32768, 32767).astype(np.int64)
        w1 = np.clip(np.rint(f * 2048.0), -32768, 32767).astype(np.int64)
        return s, s1, w0, w1

    x0, x1, a0, a1 = taps(sw, dw, True)
    y0, y1, b0, b1 = taps(sh, dh, False)
    s = src.astype(np.int64)
    h = s[:, x0, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]          # [sh, dw, c]
    r0, r1 = h[y0], h[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def save_image_chw(path, chw, scale=255.0):
    """cv2.imwrite(path, hwc * 255) for an RGB CHW float tensor/array: values are rounded half-to-even
    and saturated to uint8, as cv::saturate_cast<uchar> does."""
    a = np.asarray(chw.detach().cpu().numpy() if hasattr(chw, "detach") else chw, dtype=np.float64)
    u = np.clip(np.rint(a * scale), 0, 255).astype(np.uint8)
    Image.fromarray(np.ascontiguousarray(np.transpose(u, (1, 2, 0)))).save(path)

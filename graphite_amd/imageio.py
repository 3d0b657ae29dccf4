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

"""TEST INFRASTRUCTURE (oracle): plain-PyTorch fp32 restatement of the reference's differentiable
("whitebox") transform, for the floating-point HIP kernels of graphite_amd/csrc/wb.hip.

Follows /root/reference/transforms.py:176-228 (transform_wb, convert2NetworkWB) and
perspective_transform.py:49 (the kornia branch).  The reference delegates to kornia==0.5.10
(requirements.txt:11), which is NOT installed here and not vendored: PARITY UNPINNED for kornia's
internals.  Restated from its published source:
  * warp_perspective(src, M, dsize, align_corners=True, padding_mode='border'):
        dst_norm_trans_src_norm = N_dst @ M @ inv(N_src),  N(h, w) = [[2/(w-1), 0, -1], [0, 2/(h-1), -1], [0, 0, 1]]
        grid = transform_points(inv(dst_norm_trans_src_norm), create_meshgrid(normalized))   (all float32;
        normalised coordinate of pixel x = (x / (w - 1) - 0.5) * 2)
        F.grid_sample(src, grid, mode='bilinear', padding_mode='border', align_corners=True)
    transform_points divides by z where |z| > 1e-8 (by z + 1e-8), else by 1.
  * adjust_gamma(x, gamma) = clamp(x ** gamma, 0, 1)
  * resize(x, 32, align_corners=False) = F.interpolate(x, (32, 32), mode='bilinear', align_corners=False) for square input
Everything is differentiable torch code, so autograd of this file is the reference for the backward kernels.
Only tests/ (and bench.py's checker legs) may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

GAUSS = {3: [0.25, 0.5, 0.25],
         5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
         7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}     # cv2.getGaussianKernel(k, 0)


def blur_kernel(k):
    """what transforms.py:181-186 builds with cv2.GaussianBlur on a float64 delta image: outer(g, g), double"""
    g = np.asarray(GAUSS[int(k)], np.float64)
    return np.outer(g, g)


def normal_transform_pixel(h, w):
    tr = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    tr[0, 0] = tr[0, 0] * 2.0 / (w - 1 if w > 1 else 1e-14)
    tr[1, 1] = tr[1, 1] * 2.0 / (h - 1 if h > 1 else 1e-14)
    return tr


def src_norm_trans_dst_norm(M, src_hw, dst_hw):
    """float32 3x3 that maps normalised dst coordinates to normalised src coordinates (kornia normalize_homography + inverse)"""
    M32 = torch.as_tensor(np.asarray(M, np.float64)).float()
    n_src = normal_transform_pixel(*src_hw)
    n_dst = normal_transform_pixel(*dst_hw)
    d_n_s = n_dst @ (M32 @ torch.inverse(n_src))
    return torch.inverse(d_n_s)


def warp_perspective(src, M, dsize):
    """src [1,C,H,W] float32, M forward src->dst pixel homography, dsize = (h_out, w_out)"""
    _, _, H, W = src.shape
    h_out, w_out = dsize
    A = src_norm_trans_dst_norm(M, (H, W), (h_out, w_out))
    xs = torch.linspace(0, w_out - 1, w_out, dtype=torch.float32)        # kornia create_meshgrid(normalized_coordinates=True)
    ys = torch.linspace(0, h_out - 1, h_out, dtype=torch.float32)
    xs = (xs / (w_out - 1) - 0.5) * 2
    ys = (ys / (h_out - 1) - 0.5) * 2
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")
    base = torch.stack([gx, gy], dim=-1).permute(1, 0, 2).unsqueeze(0)   # [1, h_out, w_out, 2] = (x, y)
    ph = torch.cat([base, torch.ones_like(base[..., :1])], dim=-1)       # homogeneous
    q = ph @ A.t()                                                       # [1,h,w,3]
    z = q[..., 2:]
    scale = torch.where(z.abs() > 1e-8, 1.0 / (z + 1e-8), torch.ones_like(z))
    grid = q[..., :2] * scale
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=True)


def transform_wb(orig, att, mask, M, crop_x, crop_y, gamma, blur, net_size=32):
    """orig, att [1,3,H,W]; mask [3,H,W] (0/1).  Returns [1,3,32,32] in [0,1] (transforms.py:176-207)."""
    att = torch.clamp(att, 0.0, 1.0)
    if blur != 0:
        k = torch.from_numpy(blur_kernel(blur)).to(att.dtype)[None, None].repeat(3, 1, 1, 1)
        pert = torch.where(mask > 0.5, att, torch.zeros_like(att))
        att = torch.where(mask > 0.5, F.conv2d(pert, k, padding=int(blur) // 2, groups=3), orig)
    else:
        att = torch.where(mask > 0.5, att, orig)
    att = torch.clamp(att, 0.0, 1.0)
    att = warp_perspective(att, M, (crop_y, crop_x))
    att = torch.clamp(torch.pow(att, gamma), 0.0, 1.0)                   # kornia adjust_gamma, gain 1
    assert net_size == 32 and crop_x == crop_y
    att = F.interpolate(att, size=(32, 32), mode="bilinear", align_corners=False)
    return torch.clamp(att, 0.0, 1.0)

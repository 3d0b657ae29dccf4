"""TEST INFRASTRUCTURE (oracle): WideResNet.forward (reference cifar/wideresnet.py:25-34, 86-94, eval mode) as plain-PyTorch
code on a state dict, with the operand rounding of the HIP library's GR_PREC_BF16 mode as an option: the 3x3 convs
(stride 1 and 2) whose Cout is a multiple of 160 and Cin of 32 take activations and weights rounded to bfloat16 (nearest even) and are summed in
float64 (order-independent; the matrix cores accumulate in float32); everything else -- first conv, block1's first conv (Cin 16),
1x1 shortcuts, BatchNorm, the linear layer -- stays fp32.  A tolerance reference, not a bit-exact one.
Only tests/ and bench.py's checker legs import it."""
import torch
import torch.nn.functional as F


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _conv(x, w, stride, bf16):
    pad = w.shape[-1] // 2
    if bf16 and w.shape[-1] == 3 and w.shape[0] % 160 == 0 and w.shape[1] % 32 == 0:
        return F.conv2d(x.bfloat16().double(), w.bfloat16().double(), None, stride, pad).float()
    return F.conv2d(x, w, None, stride, pad)


def forward(sd, x, depth=34, widen=10, bf16=False):
    n = (depth - 4) // 6
    ch = [16, 16 * widen, 32 * widen, 64 * widen]
    out = _conv(x, sd["conv1.weight"], 1, bf16)
    for bi, stride in ((1, 1), (2, 2), (3, 2)):
        for i in range(n):
            p = f"block{bi}.layer.{i}"
            cin = ch[bi - 1] if i == 0 else ch[bi]
            s = stride if i == 0 else 1
            o = F.relu(_bn(sd, p + ".bn1", out))                              # :27 / :29
            y = F.relu(_bn(sd, p + ".bn2", _conv(o, sd[p + ".conv1.weight"], s, bf16)))   # :30
            y = _conv(y, sd[p + ".conv2.weight"], 1, bf16)                    # :33
            out = (out if cin == ch[bi] else _conv(o, sd[p + ".convShortcut.weight"], s, bf16)) + y   # :34
    out = F.relu(_bn(sd, "bn1", out))
    out = F.avg_pool2d(out, 8).reshape(-1, ch[3])
    return F.linear(out, sd["fc.weight"], sd["fc.bias"])


def input_grad(sd, x, target, depth=34, widen=10):
    """(logits, d CE(logits_n, target) / d x_n) per image (reduction='sum'): torch autograd of forward() is the
    reference of gr_model_input_grad for WideResNet (the classifier half of whitebox_attack.py:155-158)."""
    x = x.clone().requires_grad_(True)
    logits = forward(sd, x, depth, widen)
    loss = F.cross_entropy(logits, torch.full((x.shape[0],), int(target), dtype=torch.long), reduction="sum")
    g, = torch.autograd.grad(loss, x)
    return logits.detach(), g


def input_grad_forced(sd, x, target, masks, depth=34, widen=10, dtype=torch.float64):
    """d CE / d x with every ReLU DECISION taken from `masks` (uint8 [N, or_wrn_mask_bytes]: the oracle's = the HIP
    library's forward decisions, oracle/graphite_oracle.c or_wrn_forward_masks) instead of from this pass's own
    pre-activations: relu(z) becomes z * mask.  In float64 this is a rounding-free gradient of the function the HIP
    kernels differentiate.  Returns (logits, grad, report) where report lists, per ReLU layer, how many units' own sign
    (z > 0 in this float64 pass) differs from the forced decision and the largest |z| / rms(z) among them."""
    n = (depth - 4) // 6
    ch = [16, 16 * widen, 32 * widen, 64 * widen]
    sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.to(dtype).clone().requires_grad_(True)
    mk = torch.from_numpy(masks) if not torch.is_tensor(masks) else masks
    pos = [0]
    report = []

    def relu_forced(z, name):
        N, Cc, H, W = z.shape
        cnt = H * W * Cc
        m = mk[:, pos[0]:pos[0] + cnt].reshape(N, H, W, Cc).permute(0, 3, 1, 2).to(dtype)
        pos[0] += cnt
        own = (z.detach() > 0).to(dtype)
        diff = own != m
        if bool(diff.any()):
            rms = float(z.detach().pow(2).mean().sqrt())
            report.append((name, int(diff.sum()), float(z.detach().abs()[diff].max()) / rms))
        return z * m

    out = F.conv2d(x, sdd["conv1.weight"], None, 1, 1)
    for bi, stride in ((1, 1), (2, 2), (3, 2)):
        for i in range(n):
            p = f"block{bi}.layer.{i}"
            cin = ch[bi - 1] if i == 0 else ch[bi]
            s = stride if i == 0 else 1
            o = relu_forced(_bn(sdd, p + ".bn1", out), p + ".relu1")
            y = relu_forced(_bn(sdd, p + ".bn2", F.conv2d(o, sdd[p + ".conv1.weight"], None, s, 1)), p + ".relu2")
            y = F.conv2d(y, sdd[p + ".conv2.weight"], None, 1, 1)
            out = (out if cin == ch[bi] else F.conv2d(o, sdd[p + ".convShortcut.weight"], None, s, 0)) + y
    out = relu_forced(_bn(sdd, "bn1", out), "relu_final")
    assert pos[0] == mk.shape[1]
    out = F.avg_pool2d(out, 8).reshape(-1, ch[3])
    logits = F.linear(out, sdd["fc.weight"], sdd["fc.bias"])
    loss = F.cross_entropy(logits, torch.full((x.shape[0],), int(target), dtype=torch.long), reduction="sum")
    g, = torch.autograd.grad(loss, x)
    return logits.detach(), g, report

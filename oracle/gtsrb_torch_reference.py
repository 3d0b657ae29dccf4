"""TEST INFRASTRUCTURE (oracle): GTSRBNet.forward (reference GTSRB/GTSRBNet.py:35-52) as differentiable plain-PyTorch
fp32 code on a state dict, for the input-gradient kernels (gr_model_input_grad): torch autograd of this function is the
reference of the backward pass.  Only tests/ and bench.py's checker legs import it."""
import torch
import torch.nn.functional as F


def forward(sd, x):
    """sd: dict name -> float32 tensor with the reference's state_dict keys; x [N,3,32,32] in [-0.5, 0.5]"""
    h = F.relu(F.conv2d(x, sd["conv1_1.weight"], sd["conv1_1.bias"]))
    for a, b in (("conv2_1", "conv2_2"), ("conv3_1", "conv3_2"), ("conv4_1", "conv4_2")):
        h = F.relu(F.conv2d(h, sd[a + ".weight"], sd[a + ".bias"], padding=2))
        h = F.relu(F.conv2d(h, sd[b + ".weight"], sd[b + ".bias"], padding=2))
        h = F.max_pool2d(h, 2)
    h = h.reshape(-1, 4 * 4 * 128)
    h = F.relu(F.linear(h, sd["fc5_1.weight"], sd["fc5_1.bias"]))
    h = F.relu(F.linear(h, sd["fc6_1.weight"], sd["fc6_1.bias"]))
    return F.linear(h, sd["fc7_1.weight"], sd["fc7_1.bias"])


def forward_bf16(sd, x, layers=("conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "fc5_1", "fc6_1")):
    """The GR_PREC_BF16 mode of the HIP library restated: conv2_1 .. conv4_2 take activations and weights rounded to
    bfloat16 (nearest even), products are exact and summed wide (float64 here, so the sum does not depend on the order;
    the matrix cores accumulate in float32), bias / ReLU / max-pool in float32; fc5 / fc6 likewise; conv1_1 and fc7 stay
    fp32.  A tolerance reference, not a bit-exact one."""
    def cbf(h, name):
        if name not in layers:
            return F.relu(F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], padding=2))
        y = F.conv2d(h.bfloat16().double(), sd[name + ".weight"].bfloat16().double(), None, padding=2).float()
        return F.relu(y + sd[name + ".bias"][None, :, None, None])
    h = F.relu(F.conv2d(x, sd["conv1_1.weight"], sd["conv1_1.bias"]))
    h = F.max_pool2d(cbf(cbf(h, "conv2_1"), "conv2_2"), 2)
    h = F.max_pool2d(cbf(cbf(h, "conv3_1"), "conv3_2"), 2)
    h = F.max_pool2d(cbf(cbf(h, "conv4_1"), "conv4_2"), 2)
    def lbf(h, name):
        if name not in layers:
            return F.relu(F.linear(h, sd[name + ".weight"], sd[name + ".bias"]))
        y = F.linear(h.bfloat16().double(), sd[name + ".weight"].bfloat16().double()).float()
        return F.relu(y + sd[name + ".bias"])
    h = h.reshape(-1, 4 * 4 * 128)
    h = lbf(lbf(h, "fc5_1"), "fc6_1")
    return F.linear(h, sd["fc7_1.weight"], sd["fc7_1.bias"])


def input_grad(sd, x, target):
    """(logits, d CE(logits_n, target)/d x_n) with reduction='sum', i.e. per-image gradients"""
    x = x.clone().requires_grad_(True)
    logits = forward(sd, x)
    loss = F.cross_entropy(logits, torch.full((x.shape[0],), int(target), dtype=torch.long), reduction="sum")
    g, = torch.autograd.grad(loss, x)
    return logits.detach(), g

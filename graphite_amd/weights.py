"""state_dict -> flat tensor lists in the order the C ABI expects
(include/graphite_hip.h: gr_model_load_gtsrb / gr_model_load_wrn)."""
import numpy as np

GTSRB_LAYERS = ("conv1_1", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2",
                "fc5_1", "fc6_1", "fc7_1")


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(t), dtype=np.float32)


def gtsrb_tensor_list(state_dict):
    """20 arrays: weight, bias of each layer in GTSRBNet.__init__ order (GTSRB/GTSRBNet.py:10-33)."""
    out = []
    for name in GTSRB_LAYERS:
        out.append(_np(state_dict[name + ".weight"]))
        out.append(_np(state_dict[name + ".bias"]))
    return out


def wrn_tensor_list(state_dict, depth=34, widen=10):
    """conv1.weight; per used BasicBlock: bn1 (weight,bias,running_mean,running_var), conv1.weight,
    bn2 (same four), conv2.weight, [convShortcut.weight]; then bn1 (four), fc.weight, fc.bias.
    `sub_block1.*` (built but never used by forward, cifar/wideresnet.py:65) and
    `num_batches_tracked` are skipped."""
    n = (depth - 4) // 6
    ch = [16, 16 * widen, 32 * widen, 64 * widen]
    out = [_np(state_dict["conv1.weight"])]

    def bn(prefix):
        return [_np(state_dict[prefix + s]) for s in (".weight", ".bias", ".running_mean", ".running_var")]

    for g in range(3):
        for i in range(n):
            p = f"block{g + 1}.layer.{i}"
            cin = ch[g] if i == 0 else ch[g + 1]
            out += bn(p + ".bn1")
            out.append(_np(state_dict[p + ".conv1.weight"]))
            out += bn(p + ".bn2")
            out.append(_np(state_dict[p + ".conv2.weight"]))
            if cin != ch[g + 1]:
                out.append(_np(state_dict[p + ".convShortcut.weight"]))
    out += bn("bn1")
    out.append(_np(state_dict["fc.weight"]))
    out.append(_np(state_dict["fc.bias"]))
    return out


def bagnet_tensor_list(state_dict, layers=(3, 4, 6, 3)):
    """conv1.weight, conv2.weight, bn1 (weight, bias, running_mean, running_var); per stage and block: conv1.weight, bn1
    (four), conv2.weight, bn2 (four), conv3.weight, bn3 (four), [downsample.0.weight, downsample.1 (four) in block 0];
    fc.weight, fc.bias.  Keys of a DataParallel checkpoint ('module.' prefix, patchguard_predict.py:27-29) are accepted;
    `num_batches_tracked` is skipped."""
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}

    def bn(prefix):
        return [_np(sd[prefix + s]) for s in (".weight", ".bias", ".running_mean", ".running_var")]

    out = [_np(sd["conv1.weight"]), _np(sd["conv2.weight"])] + bn("bn1")
    for g, nblk in enumerate(layers):
        for i in range(nblk):
            p = f"layer{g + 1}.{i}"
            for j in (1, 2, 3):
                out.append(_np(sd[f"{p}.conv{j}.weight"]))
                out += bn(f"{p}.bn{j}")
            if i == 0:
                out.append(_np(sd[p + ".downsample.0.weight"]))
                out += bn(p + ".downsample.1")
    out.append(_np(sd["fc.weight"]))
    out.append(_np(sd["fc.bias"]))
    return out

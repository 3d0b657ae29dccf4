"""Which layer of the bf16 mode deviates from the bf16-operand restatement? (GRAPHITE_BF16_LAYERS = layer mask, see gr_model_set_bf16_layers.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import gtsrb_torch_reference as G
from graphite_amd import models, synth
names = {0: "conv2_1", 1: "conv2_2", 2: "conv3_1", 3: "conv3_2", 4: "conv4_1", 5: "conv4_2", 6: "fc"}
mask = int(os.environ.get("GRAPHITE_BF16_LAYERS", "0x7f"), 0)
layers = tuple(n for l, n in names.items() if (mask >> l) & 1)
if "fc" in layers:
    layers = tuple(n for n in layers if n != "fc") + ("fc5_1", "fc6_1")
sd = synth.gtsrb_state_dict(readout=os.environ.get("READOUT") or None)
model = models.GTSRBNet().load_state_dict(sd)
torch.manual_seed(0)
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 16, 3, 32, 32) - 0.5
f32 = model.forward(x).cpu()
model.set_precision("bf16")
model.hip_model.set_bf16_layers(mask)
got = model.forward(x).cpu()
tsd = {k: torch.as_tensor(v) for k, v in sd.items()}
ref = G.forward_bf16(tsd, x, layers)
ref32 = G.forward(tsd, x)
sc = float(ref.abs().max())
e = ((got - ref).abs().max(1).values / sc).sort().values
print("per-image kernel-vs-ref: median %.2e p90 %.2e max %.2e" % (float(e[len(e) // 2]), float(e[int(len(e) * 0.9)]), float(e[-1])))
print(layers, "kernel-vs-ref %.2e  ref-vs-f32ref %.2e  f32kernel-vs-f32ref %.2e" % (
    float((got - ref).abs().max()) / sc, float((ref - ref32).abs().max()) / sc, float((f32 - ref32).abs().max()) / sc), flush=True)

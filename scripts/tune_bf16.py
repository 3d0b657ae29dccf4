"""Per-layer timing of GTSRBNet in bf16 mode (gr_profile_*): python3 scripts/tune_bf16.py [N] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphite_amd import _hip, models, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_hip.set_chunk(max(64, N))
model = models.GTSRBNet().load_state_dict(synth.gtsrb_state_dict())
x = torch.rand(N, 3, 32, 32, device="cuda") - 0.5
for prec in ("f32", "bf16"):
    model.set_precision(prec)
    model.forward(x); torch.cuda.synchronize()
    _hip.profile_reset(); _hip.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        model.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{prec}: {N / dt:,.0f} images/s ({dt * 1e3:.2f} ms per {N})")
    for name, r in _hip.profile_read().items():
        print(f"   {name:34s} {r['ms'] / reps:8.3f} ms  {r['flops'] / r['ms'] / 1e9:7.1f} TF")
    _hip.profile_enable(False)

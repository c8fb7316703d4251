"""Why bench.py's secondary ContextAEReal leg reads slower than tools/bench_real.py: fused step vs forward_backward + adam, with / without another live handle."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
other = None
if mode in ("other", "other_used"):
    other = Translator(64, 64, 64, 1024, max_batch=256)
    other.init_params(0)
    if mode == "other_used":
        g = torch.Generator(device="cuda").manual_seed(1)
        dd = [(torch.randint(0, 256, (256, 64, 64, 3), device="cuda", generator=g, dtype=torch.uint8).float() / 127.5 - 1.0).contiguous() for _ in range(3)]
        for _ in range(10):
            other.dev_train_step(*(t.data_ptr() for t in dd), 256, 1e-4)
        other.sync()
H, W, B = 36, 64, 256
tr = Translator(H, W, featsize=100, max_batch=B, variant="real")
tr.init_params(0)
g = torch.Generator(device="cuda").manual_seed(7)
d = [(torch.randint(0, 256, (B, H, W, 3), device="cuda", generator=g, dtype=torch.uint8).float() / 127.5 - 1.0).contiguous() for _ in range(3)]
torch.cuda.synchronize()
def fused(): tr.dev_train_step(*(t.data_ptr() for t in d), B, 1e-4)
def split(): tr.dev_forward_backward(*(t.data_ptr() for t in d), B); tr.dev_adam(1e-4)
for name, fn in (("fused dev_train_step", fused), ("forward_backward + adam", split), ("fused dev_train_step", fused)):
    for _ in range(30): fn()
    tr.sync()
    t0 = time.perf_counter()
    for _ in range(100): fn()
    tr.sync()
    print(f"{mode:11s} {name:26s} {(time.perf_counter() - t0) * 10:.3f} ms")

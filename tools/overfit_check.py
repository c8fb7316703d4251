"""Sanity run, not a test: Adam on ONE fixed batch of synthetic frames must drive the loss down (both precisions).
    python tools/overfit_check.py [steps]"""
import sys

import os

import numpy as np
import torch  # noqa: F401  (loads the HIP runtime before libctxtrans)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
B = 32
# smooth synthetic frames (random low-frequency fields), the same scene family for src / ctx / tgt
def frames():
    low = rng.standard_normal((B, 8, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(low, 8, axis=1), 8, axis=2)
    return np.tanh(up).astype(np.float32)
src, ctx, tgt = frames(), frames(), frames()
for prec in ("f32", "bf16x3"):
    with Translator(64, 64, 64, 1024, max_batch=B, precision=prec) as tr:
        tr.init_params(1)
        hist = [tr.train_step(src, ctx, tgt, lr=1e-4)["loss"] for _ in range(steps)]
        print(f"[{prec}] loss: step 0 {hist[0]:.4e}  step {steps // 10} {hist[steps // 10]:.4e}  step {steps // 2} {hist[steps // 2]:.4e}  "
              f"last {hist[-1]:.4e}  (x{hist[0] / hist[-1]:.1f} lower, all finite: {bool(np.isfinite(hist).all())})")

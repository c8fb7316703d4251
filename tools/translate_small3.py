import os, sys, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402
rng = np.random.default_rng(0)
mode = sys.argv[1]
for H, W in ((36, 64), (64, 64)):
    tr = Translator(H, W, featsize=100, max_batch=1000, variant="real")
    tr.init_params(0)
    if mode == "enc":
        for B in (25, 250, 1000):
            x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
            for _ in range(5):
                tr.encode(x)
    for B in (25, 250):
        x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        for i in range(8):
            t0 = time.perf_counter(); tr.translate(x, x[0]); dt = time.perf_counter() - t0
            print(f"{H}x{W} B={B} call {i}: {1e3 * dt:.3f} ms")
    tr.close()

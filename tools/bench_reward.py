"""Times the reward hook's two call sites (rllab/sampler/base.py:216-218, 234-235) through the C ABI with host
uint8 frames in and host f32 out (PCIe included), at the reference's batch of 25 and with several paths per
launch.  Development tool; prints a small table.   python tools/bench_reward.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402

H = W = 64
tr = Translator(H, W, 64, 1024, max_batch=1000)
tr.init_params(0)
rng = np.random.default_rng(0)
print(f"{'call':28s} {'B':>5s} {'ms/call':>9s} {'frames/s':>10s} {'paths(25)/s':>12s}")
for name, fn in (("encode (per-path cost)", lambda x: tr.encode(x)), ("translate (demo cache)", lambda x: tr.translate(x, x[0]))):
    for B in (25, 100, 250, 1000):
        x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        for _ in range(3):
            fn(x)
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            fn(x)
        dt = (time.perf_counter() - t0) / n
        print(f"{name:28s} {B:5d} {dt * 1e3:9.3f} {B / dt:10.0f} {B / 25 / dt:12.0f}")
tr.close()

"""ContextAEReal (BASELINE configs[4]: the 'sweep' translator): train-step and reward-call timings through the C ABI.
Development tool; prints a small table.   python tools/bench_real.py"""
import gc
import os
import sys
import time

import numpy as np
import torch  # before the library: both must share one HIP runtime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402

rng = np.random.default_rng(0)
print(f"{'call':34s} {'HxW':>7s} {'B':>5s} {'ms/call':>9s} {'frames/s':>10s}")
for H, W in ((36, 64), (64, 64)):
    tr = Translator(H, W, featsize=100, max_batch=1000, variant="real")
    tr.init_params(0)
    gc.freeze()        # a latency-critical caller's idiom: CPython's full collection (35-56 ms with torch imported) otherwise lands in one of the
                       # timed calls -- profiles/round5_c_reward_latency.txt
    for B in (256, 1000):
        fr = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(3)]
        f32 = [(x.astype(np.float32) / 127.5 - 1) for x in fr]
        for _ in range(3):
            tr.train_step(*f32, lr=1e-4)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            tr.train_step(*f32, lr=1e-4)
        dt = (time.perf_counter() - t0) / n
        print(f"{'train_step (host f32 in, PCIe incl.)':34s} {H:3d}x{W:<3d} {B:5d} {dt * 1e3:9.3f} {B / dt:10.0f}")
        d = [torch.from_numpy(x).cuda() for x in f32]
        torch.cuda.synchronize()
        for _ in range(3):
            tr.dev_forward_backward(*(t.data_ptr() for t in d), B)
            tr.dev_adam(1e-4)
        tr.sync()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            tr.dev_forward_backward(*(t.data_ptr() for t in d), B)
            tr.dev_adam(1e-4)
        tr.sync()
        dt = (time.perf_counter() - t0) / n
        print(f"{'train_step (frames resident in HBM)':34s} {H:3d}x{W:<3d} {B:5d} {dt * 1e3:9.3f} {B / dt:10.0f}")
    for name, fn in (("encode", lambda x: tr.encode(x)), ("translate", lambda x: tr.translate(x, x[0]))):
        for B in (25, 250, 1000):
            x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
            for _ in range(3):
                fn(x)
            n = 20
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                fn(x)
                ts.append(time.perf_counter() - t0)
            dt = sum(ts) / n
            note = ""
            if max(ts) > 3 * sorted(ts)[n // 2]:          # a few slow calls inside the mean: say so
                note = "   (median %.3f ms, max %.3f ms)" % (1e3 * sorted(ts)[n // 2], 1e3 * max(ts))
            print(f"{name:34s} {H:3d}x{W:<3d} {B:5d} {dt * 1e3:9.3f} {B / dt:10.0f}{note}")
    tr.close()

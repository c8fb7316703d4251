"""tools/bench_real.py's sequence (two handles in turn) with per-call times of every inference section: where is the slow call?"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
rng = np.random.default_rng(0)
variant = sys.argv[1] if len(sys.argv) > 1 else "asis"
import gc
if variant == "nogc":
    gc.disable()
if variant == "gcstats":
    gc.callbacks.append(lambda phase, info: phase == "stop" and info["generation"] == 2 and print("   [gc] full collection, collected", info["collected"], flush=True))
for H, W in ((36, 64), (64, 64)):
    tr = Translator(H, W, featsize=100, max_batch=1000, variant="real")
    tr.init_params(0)
    for B in (256, 1000):
        fr = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(3)]
        f32 = [(x.astype(np.float32) / 127.5 - 1) for x in fr]
        for _ in range(13):
            tr.train_step(*f32, lr=1e-4)
        d = [torch.from_numpy(x).cuda() for x in f32]
        torch.cuda.synchronize()
        for _ in range(23):
            tr.dev_forward_backward(*(t.data_ptr() for t in d), B)
            tr.dev_adam(1e-4)
        tr.sync()
    for name, fn in (("encode", lambda x: tr.encode(x)), ("translate", lambda x: tr.translate(x, x[0]))):
        for B in (25, 250, 1000):
            x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
            ts = []
            for _ in range(23):
                t0 = time.perf_counter(); fn(x); ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{H}x{W} {name} B={B}: " + " ".join(f"{t:.2f}" for t in ts))
    tr.close()
    if variant == "sleep":
        time.sleep(1.0)

"""Where the one slow `encode 64x64 B = 25` call of tools/bench_real.py comes from: the same sequence with per-call times, in variants.
    python tools/outlier_hunt.py [full|notorch|nogc|pool0]"""
import gc, os, sys, time
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
if mode != "notorch":
    import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
if mode == "nogc":
    gc.disable()
rng = np.random.default_rng(0)
H = W = 64
tr = Translator(H, W, featsize=100, max_batch=1000, variant="real")
tr.init_params(0)
if mode == "pool0":
    tr._pool.min_bytes = 1 << 40            # no pooling: every result array fresh
for B in (256, 1000):
    fr = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(3)]
    f32 = [(x.astype(np.float32) / 127.5 - 1) for x in fr]
    for _ in range(13):
        tr.train_step(*f32, lr=1e-4)
    if mode != "notorch":
        d = [torch.from_numpy(x).cuda() for x in f32]
        torch.cuda.synchronize()
        for _ in range(23):
            tr.dev_forward_backward(*(t.data_ptr() for t in d), B)
            tr.dev_adam(1e-4)
        tr.sync()
x = rng.integers(0, 256, (25, H, W, 3), dtype=np.uint8)
ts = []
for i in range(23):
    t0 = time.perf_counter(); tr.encode(x); ts.append((time.perf_counter() - t0) * 1e3)
print(mode, "encode B=25 calls [ms]:", " ".join(f"{t:.2f}" for t in ts))
tr.close()

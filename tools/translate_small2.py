import os, sys, time
import numpy as np
import torch  # noqa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402
H, W = 64, 64
tr = Translator(H, W, featsize=100, max_batch=1000, variant="real")
tr.init_params(0)
rng = np.random.default_rng(0)
def t(fn, n=6, tag=""):
    for i in range(n):
        t0 = time.perf_counter(); fn(); print(f"{tag} call {i}: {1e3 * (time.perf_counter() - t0):.3f} ms")
x25 = rng.integers(0, 256, (25, H, W, 3), dtype=np.uint8)
stage = sys.argv[1] if len(sys.argv) > 1 else "all"
if stage in ("all", "train"):
    for B in (256, 1000):
        f32 = [(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1) for _ in range(3)]
        for _ in range(3):
            tr.train_step(*f32, lr=1e-4)
        if stage == "all":
            d = [torch.from_numpy(x).cuda() for x in f32]
            torch.cuda.synchronize()
            for _ in range(5):
                tr.dev_forward_backward(*(t_.data_ptr() for t_ in d), B)
                tr.dev_adam(1e-4)
            tr.sync()
if stage in ("all", "encode"):
    for B in (25, 250, 1000):
        x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        for _ in range(3):
            tr.encode(x)
t(lambda: tr.translate(x25, x25[0]), tag="translate25")
tr.close()

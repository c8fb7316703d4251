"""ContextAEReal train step (frames resident) at several batch sizes: the FC middle on rchain.hip (CTX_RCHAIN=1) or on the implicit GEMM (0)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
rng = np.random.default_rng(0)
for H, W in ((36, 64), (64, 64)):
    tr = Translator(H, W, featsize=100, max_batch=1024, variant="real")
    tr.init_params(0)
    out = []
    for B in (32, 100, 256, 512, 1024):
        d = [torch.from_numpy((rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1)).cuda() for _ in range(3)]
        torch.cuda.synchronize()
        for _ in range(5):
            tr.dev_forward_backward(*(t.data_ptr() for t in d), B); tr.dev_adam(1e-4)
        tr.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(30):
                tr.dev_forward_backward(*(t.data_ptr() for t in d), B); tr.dev_adam(1e-4)
            tr.sync()
            best = min(best, (time.perf_counter() - t0) / 30)
        out.append(f"B={B}: {best * 1e3:.3f}")
    print(f"rchain={os.environ.get('CTX_RCHAIN', 'default')} {H}x{W}  " + "  ".join(out))
    tr.close()

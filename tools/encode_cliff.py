"""ctx_encode per-call time against the batch (the B = 1000 cliff of VERDICT r1 #12): with / without the float frames
coming back, and the device-only part (HIP events around the forward).   python tools/encode_cliff.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
Bs = [int(a) for a in sys.argv[1:]] or [250, 500, 750, 1000, 1024]
tr = Translator(64, 64, 64, 1024, max_batch=max(Bs)); tr.init_params(0)
rng = np.random.default_rng(0)
for B in Bs:
    x = rng.integers(0, 256, (B, 64, 64, 3), dtype=np.uint8)
    for rf in (False, True):
        for _ in range(3): tr.encode(x, return_frames=rf)
        t0 = time.perf_counter()
        for _ in range(10): tr.encode(x, return_frames=rf)
        dt = (time.perf_counter() - t0) / 10
        print(f"B {B:5d} frames_back {int(rf)}  {dt*1e3:8.3f} ms  {B/dt:10.0f} frames/s", flush=True)
    out = (np.empty((B, 1024), np.float32), np.empty(x.shape, np.float32))        # the same result buffers every call
    for _ in range(3): tr.encode(x, out=out)
    t0 = time.perf_counter()
    for _ in range(10): tr.encode(x, out=out)
    dt = (time.perf_counter() - t0) / 10
    print(f"B {B:5d} frames_back 1 (caller-owned result buffers)  {dt*1e3:8.3f} ms  {B/dt:10.0f} frames/s", flush=True)
tr.close()

"""Wall time of each of the FIRST calls of a reward-hook fetch on a fresh handle (where the 37-56 ms call of tools/bench_real.py sits).
    python tools/first_calls.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
rng = np.random.default_rng(0)
for label, mk, H, W in (("ContextAEReal 64x64 max_batch 1000", lambda: Translator(64, 64, featsize=100, max_batch=1000, variant="real"), 64, 64),
                        ("ContextAEReal 36x64 max_batch 1000", lambda: Translator(36, 64, featsize=100, max_batch=1000, variant="real"), 36, 64),
                        ("ContextSkipNew 64x64 max_batch 25", lambda: Translator(64, 64, 64, 1024, max_batch=25), 64, 64)):
    tr = mk(); tr.init_params(0)
    x = rng.integers(0, 256, (25, H, W, 3), dtype=np.uint8)
    for name, fn in (("encode", lambda: tr.encode(x)), ("translate", lambda: tr.translate(x, x[0]))):
        ts = []
        for i in range(12):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{label:36s} {name:10s} " + " ".join(f"{t:7.2f}" for t in ts))
    tr.close()

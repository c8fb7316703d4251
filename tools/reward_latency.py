"""Latency distribution of the reward hook's two fetches at the reference's call shape (rllab/sampler/base.py:216-218, 234-235: 25 frames
per call, >= 250 calls per TRPO iteration): median, p99 and max over N calls, and WHERE the slow calls sit.   python tools/reward_latency.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
print(f"{'model':26s} {'call':10s} {'median':>8s} {'p99':>8s} {'max':>8s} ms   slowest calls (index: ms)")
for label, mk, H, W in (("ContextSkipNew 64x64", lambda: Translator(64, 64, 64, 1024, max_batch=25), 64, 64),
                        ("ContextAEReal 36x64", lambda: Translator(36, 64, featsize=100, max_batch=25, variant="real"), 36, 64),
                        ("ContextAEReal 64x64", lambda: Translator(64, 64, featsize=100, max_batch=25, variant="real"), 64, 64)):
    tr = mk()
    tr.init_params(0)
    x = rng.integers(0, 256, (25, H, W, 3), dtype=np.uint8)
    # "path cost": the whole of base.py:232-249 for one path on the device (ctx_reward_costs: encoder + both distances; 25 floats come back)
    tr.reward_set_cache(0, rng.standard_normal((25, tr.featsize)).astype(np.float32), rng.uniform(-1, 1, (25, H, W, 3)).astype(np.float32))
    for name, fn in (("encode", lambda: tr.encode(x)), ("translate", lambda: tr.translate(x, x[0])), ("path cost", lambda: tr.reward_costs(0, x, 0.1))):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(N):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        a = np.array(ts)
        worst = np.argsort(-a)[:4]
        print(f"{label:26s} {name:10s} {np.median(a):8.3f} {np.percentile(a, 99):8.3f} {a.max():8.3f}      " + "  ".join(f"{i}: {a[i]:.2f}" for i in sorted(worst)))
    tr.close()

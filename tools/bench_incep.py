"""ContextAEInception2 (BASELINE configs[3]: 64 triples per GPU of Mixed_7c feature maps, filters 1024/1024/512/512):
train-step time on synthetic features resident in HBM, per precision, with the per-kernel table.
Development tool.   python tools/bench_incep.py [grid] [batch]"""
import os
import sys
import time

import numpy as np
import torch  # before the library: both must share one HIP runtime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = 2048
gen = torch.Generator(device="cuda").manual_seed(0)
feats = [torch.relu(torch.randn((B, G, G, C), device="cuda", generator=gen)) for _ in range(3)]
for prec in ("f32", "bf16x3"):
    with Translator(G, G, 64, 1024, max_batch=B, variant="inception2", C=C, precision=prec) as tr:
        t0 = time.perf_counter()
        tr.init_params(1234)
        print(f"[{prec}] grid {G}x{G}x{C}, batch {B}, {tr.n_params / 1e6:.1f} M parameters (init {time.perf_counter() - t0:.1f} s)")
        ptrs = [t.data_ptr() for t in feats]
        for _ in range(3):
            tr.dev_forward_backward(*ptrs, B)
            tr.dev_adam(1e-4)
        tr.sync()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            tr.dev_forward_backward(*ptrs, B)
            tr.dev_adam(1e-4)
        tr.sync()
        dt = (time.perf_counter() - t0) / n
        ents = tr.profile_step(*ptrs, B, iters=3)
        fl = sum(e["flops"] for e in ents)
        print(f"   {dt * 1e3:.2f} ms/step  {B / dt:.0f} triples/s  {fl / dt / 1e12:.1f} TF/s algorithmic  loss {tr.dev_scalars()['loss']:.4g}")
        for k, v in Translator.kernel_table(ents).items():
            print(f"      {k:36s} {v['ms']:8.3f} ms  {v['launches']:3d} launches  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:7.1f} TF/s")

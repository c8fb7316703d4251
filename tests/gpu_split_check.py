"""Error and speed of CTX_PREC_BF16X3 against exact f32 and the float64 oracle.  Development tool."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402
from oracle import ctx_oracle as o  # noqa: E402
from tests._align import align_skipnew_cache  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def case(H, W, d, F, B, stddev):
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=d, gf_dim=d, featsize=F)
    p = o.init_params(cfg, 7, np.float64, stddev=stddev)
    rng = np.random.default_rng(0)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = rng.standard_normal(p[n].shape) * stddev
    fr = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(3)]
    src, ctx, tgt = (o.preprocess_u8(x).astype(np.float64) for x in fr)
    print(f"--- {H}x{W} d={d} F={F} B={B} stddev={stddev}")
    for prec in ("f32", "bf16x3"):
        res, c = o.forward(p, src, ctx, tgt, cfg)
        with Translator(H, W, d, F, max_batch=B, precision=prec) as tr:
            tr.set_params(p)
            ev = tr.evaluate(*(x.astype(np.float32) for x in (src, ctx, tgt)))
            nflip, worst = align_skipnew_cache(tr, c, B)
            print(f"        {nflip} lrelu' branches aligned (largest |x|/max {worst:.1e})")
            g = o.backward(p, c, cfg)
            tr.train_step(*(x.astype(np.float32) for x in (src, ctx, tgt)), lr=0.0)
            gg = tr.get_grads()
            ge = {n: rel(gg[n], g[n]) for n in g}
            worst = max(ge, key=ge.get)
            gl2 = {n: float(np.linalg.norm(np.asarray(gg[n], np.float64) - g[n]) / (np.linalg.norm(g[n]) + 1e-300)) for n in g}
            print(f"{prec:7s} out {rel(ev['out'], res['out']):.2e} out2 {rel(ev['out2'], res['out2']):.2e} "
                  f"loss {abs(ev['loss'] - res['loss']) / res['loss']:.2e} sim {abs(ev['simloss'] - res['simloss']) / res['simloss']:.2e} "
                  f"grad max-rel worst {ge[worst]:.2e} ({worst}) median {np.median(list(ge.values())):.2e} | L2-rel worst {max(gl2.values()):.2e}")


case(32, 32, 32, 128, 4, 0.05)
case(64, 64, 64, 1024, 4, 0.02)

B = 256
rng = np.random.default_rng(0)
fr = [torch.from_numpy(rng.integers(0, 256, (B, 64, 64, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1).cuda() for _ in range(3)]
for prec in ("f32", "bf16x3"):
    with Translator(max_batch=B, precision=prec) as tr:
        tr.init_params(1234)
        for _ in range(3):
            tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
            tr.dev_adam(1e-4)
        tr.sync()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
            tr.dev_adam(1e-4)
        tr.sync()
        dt = (time.perf_counter() - t0) / n
        print(f"{prec}: {dt * 1e3:.2f} ms/step  {B / dt:.0f} frames/s   scalars {tr.dev_scalars()}")
        ents = tr.profile_step(*(t.data_ptr() for t in fr), B, iters=3)
        for k, v in Translator.kernel_table(ents).items():
            print(f"   {k:36s} {v['ms']:8.3f} ms  {v['launches']:3d}  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:7.1f} TF/s")

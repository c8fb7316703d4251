"""cold-run hunt: dconv2 FIRST in the process (evaluate at B), then dconv; where do the buffers differ?"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
from tests.golden import make_golden as mg
tag = sys.argv[1] if len(sys.argv) > 1 else "real_f100_36x64_b256"
mod, cfg, p32, (src, ctx, tgt) = mg.big_case(tag)
B, H, W = src.shape[0], cfg.H, cfg.W
sizes = {"a0": (3*B, H, W, 32), "a1": (3*B, H//2, W//2, 16), "a2": (3*B, H//2, W//2, 16), "a3": (3*B, H//4, W//4, 8),
         "e1": (2*B, H//2, W//2, 16), "e2": (2*B, H//2, W//2, 16), "e3": (2*B, H, W, 32), "out": (2*B, H, W, 3)}
res = {}
for mode in ("3", "1", "3b"):
    os.environ["CTX_DCONV"] = mode[0]
    with Translator(H, W, featsize=100, max_batch=B, variant="real") as tr:
        tr.set_params(p32)
        ev = tr.evaluate(src, ctx, tgt)
        res[mode] = {k: tr.debug_read(k, int(np.prod(sh))).reshape(sh) for k, sh in sizes.items()}
        print(mode, [ev[k] for k in ("loss", "simloss", "recon1", "recon2")])
for m in ("3", "3b"):
    for k in sizes:
        a, b = res["1"][k], res[m][k]
        bad = np.abs(a - b) > 1e-4 * np.abs(a).max()
        if bad.any():
            idx = np.argwhere(bad)
            print(m, k, "bad", int(bad.sum()), "images", sorted(set(idx[:, 0].tolist()))[:10], "rows", sorted(set(idx[:, 1].tolist()))[:12], "cols", (idx[:, 2].min(), idx[:, 2].max()), "ch", sorted(set(idx[:, 3].tolist()))[:16], "nan", int(np.isnan(b).sum()))
        else:
            print(m, k, "ok")

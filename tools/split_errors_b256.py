"""Error table of the split-precision mode (ctx_config.precision = bf16x3: whatever operand format / term count libctxtrans.so was BUILT
with -- igemm_split.h: CTX_SPLIT_F16, CTX_SPLIT_TERMS) against the B = 256 float64 fixture of BASELINE configs[1]
(tests/golden/b256_skipnew_*.npz): outputs, scalars, per-tensor gradient deviations (4096 samples + 16 projections), and ms/step.
   python tools/split_errors_b256.py [label]
Measurement tool of VERDICT r5 item 8 (profiles/round6_e_fp16_split_errors.txt); the exact-f32 row is printed first as the yardstick."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imitation_from_observation_amd import Translator  # noqa: E402
from oracle import ctx_oracle as o  # noqa: E402
from tests.golden import make_golden as mg  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "split build"
z = np.load(os.path.join(ROOT, "tests", "golden", mg.B256_TAG + ".npz"))
cfg, p, frames = mg.b256_case()
B = int(z["B"])
src, ctx, tgt = (o.preprocess_u8(f) for f in frames)
names = [n for n, _ in o.param_specs(cfg)]
keep = list(z["keep"])


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


for prec in ("f32", "bf16x3"):
    with Translator(cfg.H, cfg.W, cfg.df_dim, cfg.featsize, max_batch=B, precision=prec) as tr:
        tr.set_params(p)
        ev = tr.evaluate(src, ctx, tgt)
        iz, tz = tr.last_codes()
        outs = {k: relmax(got[keep], z[k + "_keep"]) for got, k in ((ev["out"], "out"), (ev["out2"], "out2"), (tz, "translated_z"), (iz, "input_z"))}
        scal = {k: abs(ev[k] - r) / abs(r) for k, r in zip(("loss", "simloss", "recon1", "recon2"), z["scalars"])}
        tr.train_step(src, ctx, tgt, lr=0.0)
        gg = tr.get_grads()
        probes = mg.b256_probes([(n, int(np.prod(gg[n].shape))) for n in names])
        samp, proj = {}, {}
        for i, n in enumerate(names):
            a = np.asarray(gg[n], np.float64).reshape(-1)
            seed, idx = probes[n]
            ref_s = z["grad_samples"][i][: len(idx)]
            samp[n] = float(np.linalg.norm(a[idx] - ref_s) / (np.linalg.norm(ref_s) + 1e-30))
            proj[n] = float(np.sqrt(np.mean((mg.b256_project(a, seed) - z["grad_proj"][i]) ** 2)) / z["grad_digest"][i][2])
        fr = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (src, ctx, tgt)]
        for _ in range(3):
            tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
            tr.dev_adam(0.0)
        tr.sync()
        t0 = time.perf_counter()
        for _ in range(15):
            tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
            tr.dev_adam(0.0)
        tr.sync()
        ms = (time.perf_counter() - t0) / 15 * 1e3
    tag = "exact f32" if prec == "f32" else label
    wn = max(samp, key=samp.get)
    d4 = max(samp[n] for n in names if n.startswith("deconv/d_h4"))       # upstream of every lrelu' mask: pure product error, no branch flips
    print(f"{tag:34s} {ms:6.2f} ms/step | out {outs['out']:.1e} out2 {outs['out2']:.1e} translated_z {outs['translated_z']:.1e} input_z {outs['input_z']:.1e} | "
          f"loss {scal['loss']:.1e} simloss {scal['simloss']:.1e} | gradient rel-L2: d_h4 (no flips) {d4:.1e}, worst {samp[wn]:.1e} ({wn}), median {np.median(list(samp.values())):.1e}, "
          f"worst projection {max(proj.values()):.1e}")

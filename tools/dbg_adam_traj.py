"""Per-tensor error of Adam's first moment after 3 steps against the float64 oracle (debug of tests/test_gpu_parity.py::test_adam_trajectory)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
from oracle import ctx_oracle as o
from tests.test_gpu_parity import make_case
H, W, d, F, B = 32, 32, 32, 128, 4
cfg, p, fr = make_case(H, W, d, F, B, seed=3)
src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
q = {k: v.copy() for k, v in p.items()}
m = {k: np.zeros_like(v) for k, v in q.items()}
v = {k: np.zeros_like(v_) for k, v_ in q.items()}
tr = Translator(H, W, d, F, max_batch=B)
tr.set_params(p)
for t in range(1, 4):
    r, _ = o.train_step(q, m, v, t, *(x.astype(np.float64) for x in (src, ctx, tgt)), 1e-3, cfg)
    sc = tr.train_step(src, ctx, tgt, lr=1e-3)
    gg = tr.get_grads()
    mm, vv, step = tr.get_adam_state()
    got = tr.get_params()
    off = 0
    print(f"-- step {t} loss rel {abs(sc['loss'] - r['loss']) / abs(r['loss']):.2e}")
    for n, shp in o.param_specs(cfg):
        sz = int(np.prod(shp))
        a = mm[off:off + sz]; b = np.asarray(m[n]).reshape(-1)
        pa = np.asarray(got[n], np.float64).reshape(-1); pb = np.asarray(q[n]).reshape(-1)
        e = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
        ep = np.abs(pa - pb).max()
        if e > 2e-5 or ep > 2e-4: print(f"   {n:34s} m rel_l2 {e:.2e}   |m| {np.linalg.norm(b):.3e}   max |param diff| {ep:.2e}")
        off += sz
tr.close()

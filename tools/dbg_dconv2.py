"""dconv (CTX_DCONV=1) against dconv2 (CTX_DCONV=3) on the parameters / frames of tests/test_gpu_real.py::make: activations, sign changes, gradients."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
from oracle import ctx_oracle as o
from tests.test_gpu_real import make
H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg, p, fr = make(H, W, B)
src, ctx, tgt = (o.preprocess_u8(x) for x in fr)
res = {}
for mode in ("1", "3"):
    os.environ["CTX_DCONV"] = mode
    with Translator(H, W, featsize=100, max_batch=B, variant="real") as tr:
        tr.set_params(p)
        tr.train_step(src, ctx, tgt, lr=0.0)
        g = tr.get_grads()
        sizes = {"a0": 3*B*H*W*32, "a1": 3*B*(H//2)*(W//2)*16, "a2": 3*B*(H//2)*(W//2)*16, "a3": 3*B*(H//4)*(W//4)*8,
                 "e1": 2*B*(H//2)*(W//2)*16, "e2": 2*B*(H//2)*(W//2)*16, "e3": 2*B*H*W*32, "out": 2*B*H*W*3}
        res[mode] = ({k: tr.debug_read(k, n) for k, n in sizes.items()}, g)
for k in res["1"][0]:
    a, b = res["1"][0][k], res["3"][0][k]
    flip = (a >= 0) != (b >= 0)
    print(k, "max rel diff %.2e" % (np.abs(a - b).max() / np.abs(a).max()), "sign changes", int(flip.sum()), "largest |x|/max at a change %.1e" % (np.abs(a[flip]).max() / np.abs(a).max() if flip.any() else 0))
for k in res["1"][1]:
    a, b = res["1"][1][k], res["3"][1][k]
    print("grad %-28s %.2e" % (k, np.abs(a - b).max() / (np.abs(a).max() + 1e-30)))

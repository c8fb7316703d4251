"""Runs the REFERENCE'S OWN Inception-v3 graph code -- nets/inception_v3.py `inception_v3(images, num_classes=1001, is_training=False)` under
`inception_v3_arg_scope()` (nets/inception_utils.py), exactly the call of rllab/sampler/base.py:121-127 and scripts/train_script.py:104-114 --
loaded from REFERENCE_ROOT at run time on the eager tf / slim stand-in of tests/golden/tf_standin.py, and compares all 18 end points up to
Mixed_7c with oracle/inception_oracle.py in float64 (bar 1e-9), together with the variable inventory: every variable the reference creates
under InceptionV3/ up to Mixed_7c must be one the oracle holds, with its shape (the AuxLogits / Logits heads, which the path never fetches,
get zeros and are listed).

Build container only; nothing of the reference is stored.  Pins the front end's WIRING (branches, kernel sizes, strides, paddings, concat
orders, scope names) to the reference's code -- not slim's / TensorFlow's op semantics.

    python tests/golden/check_reference_inception.py [-v]
"""
import contextlib
import importlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import tf_standin  # noqa: E402
from oracle import inception_oracle as io_  # noqa: E402

BAR = 1e-9


def reference_root():
    return os.environ.get("REFERENCE_ROOT", "/root/reference")


def run(S, B, seed):
    """frames [B, S, S, 3] in [-1, 1]; returns (rows, worst, created, extra)."""
    p = io_.init_params(seed, np.float64)
    values = {"InceptionV3/" + k if not k.startswith("InceptionV3/") else k: v for k, v in p.items()}
    rng = np.random.default_rng(seed + 1)
    x = rng.uniform(-1, 1, (B, S, S, 3))
    root = reference_root()
    with tf_standin.install(values) as st:
        st.lenient = ("InceptionV3/AuxLogits", "InceptionV3/Logits")
        saved = list(sys.path)
        for m in [k for k in sys.modules if k == "nets" or k.startswith("nets.")]:
            del sys.modules[m]
        sys.path.insert(0, root)
        try:
            net = importlib.import_module("nets.inception_v3")
            tf = sys.modules["tensorflow"]
            slim = tf.contrib.slim
            with contextlib.redirect_stdout(io.StringIO()):
                with slim.arg_scope(net.inception_v3_arg_scope()):
                    _, end_points = net.inception_v3(tf_standin.placeholder(x), num_classes=1001, is_training=False)
        finally:
            sys.path[:] = saved
            for m in [k for k in sys.modules if k == "nets" or k.startswith("nets.")]:
                del sys.modules[m]
        created = [n for n, new in st.get_variable_calls if new]
        extra = list(st.extra_vars)
    want = io_.forward({k: v for k, v in p.items()}, x)
    rows, worst = [], 0.0
    for name, ref in want.items():
        got = end_points[name].numpy()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        dev = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300))
        rows.append((name, got.shape, dev))
        worst = max(worst, dev)
    inv = ["InceptionV3/" + k if not k.startswith("InceptionV3/") else k for k in p]
    assert sorted(created) == sorted(inv), (sorted(set(inv) - set(created))[:5], sorted(set(created) - set(inv))[:5])
    return rows, worst, created, extra


CASES = {"inception_v3_125x125_b2": lambda: run(125, 2, 41), "inception_v3_299x299_b1": lambda: run(299, 1, 42)}


def main():
    if not os.path.isdir(reference_root()):
        print("no reference tree at", reference_root(), "- nothing checked")
        return 2
    bad = 0
    for name, fn in CASES.items():
        rows, worst, created, extra = fn()
        print(f"{name:28s} worst deviation {worst:.2e} over {len(rows)} end points; {len(created)} variables of the oracle's inventory created, "
              f"{len(extra)} head variables outside the path (zeros): {'OK' if worst <= BAR else 'DIFFERS'}")
        for n, shp, dev in rows:
            if dev > BAR or "-v" in sys.argv:
                print(f"    {n:18s} {str(shp):22s} {dev:.2e}")
        if "-v" in sys.argv:
            print("    outside the path:", ", ".join(f"{n} {s}" for n, s in extra))
        bad += worst > BAR
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

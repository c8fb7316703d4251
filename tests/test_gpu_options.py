"""Every value of the per-handle tuning switches (include/ctxtrans.h: "tuning switches") that selects ANOTHER kernel or schedule, run on
launch shapes where the switch takes effect, against the same step on the default switches (which the oracle suites pin: the B = 256
fixture of tests/test_gpu_baseline_configs.py for the production net, tests/test_gpu_bench_shapes.py for ContextAEReal).

The header's promise -- "results never depend on a switch beyond f32 summation order" -- as a test: outputs and codes <= 1e-5 of the
default's, d_h4's gradients (upstream of every lrelu' mask) <= 1e-5, every other gradient tensor <= 1e-2 rel-L2 (a different summation
order puts a handful of activations on the other side of lrelu's kink -- tests/_align.py; measured up to 3.1e-3 -- while a kernel that drops ONE
of 25 taps moves its tensor by 4e-2).
The matrix below is the list the header documents ("each combination is a code path the parity suite does not enumerate")."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (option, values other than the default).  Bit masks are walked bit by bit down from the default AND as single bits, so that every
# documented bit is both removed from the full set and run alone with its prerequisites.
SKIPNEW_MATRIX = [
    ("overlap", [0]),                       # one stream (default -1 -> 1: three lanes)
    ("posmajor", [0]),                      # image-major implicit GEMM for the conv / transposed conv / filter gradient at >= 64 images
    ("xcd_swizzle", [0, 1, 2, 4, 3, 5, 6]),
    ("balance", [0, 1, 8, 2, 3, 4, 5, 13]),
    ("wconvt", [0, 1, 3, 5, 7, 15, 23, 29]),     # 23 / 29: the two combinations round 4 shipped at different times
    ("direct3", [0, 1, 3, 5, 7, 9, 15, 23]),
    ("early_adam", [0]),
    ("adam_prio", [0, 1, -1]),
]
REAL_MATRIX = [
    ("overlap", [0, 1]),
    ("dconv", [0, 1, 5, 7]),                # 0: implicit GEMM on padded channels; 1: dconv_fwd_kernel only; 5 / 7: + the four-class LDS-DMA launches
    ("rchain", [0]),
    ("direct3", [0, 15]),                   # d_h4 forward: product + gather, convt3 (vector ALUs), convt3m (default 31)
    ("posmajor", [0]),
]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def run_step(make, frames, env, monkeypatch):
    """One evaluate + one Adam step on a fresh handle created under `env` (create-only switches are read from the environment at ctx_create)."""
    for k in [k for k in os.environ if k.startswith("CTX_") and k not in ("CTX_RCCL_LIB",)]:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    with make() as tr:
        for k, v in env.items():
            assert tr.get_option(k[4:].lower()) == v or k == "CTX_ADAM_PRIO", (k, v, tr.get_option(k[4:].lower()))
        tr.init_params(11)
        src, ctx, tgt = frames
        ev = tr.evaluate(src, ctx, tgt)
        iz, tz = tr.last_codes()
        sc = tr.train_step(src, ctx, tgt, lr=1e-4)
        g = tr.get_grads()
        return dict(out=ev["out"].copy(), out2=ev["out2"].copy(), iz=iz.copy(), tz=tz.copy(), loss=sc["loss"], g=g)


def check(ref, got, tag):
    for k in ("out", "out2", "iz", "tz"):
        assert relmax(got[k], ref[k]) <= 1e-5, (tag, k, relmax(got[k], ref[k]))
    assert abs(got["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"]), tag
    worst = 0.0
    for n in ref["g"]:
        e = rel_l2(got["g"][n], ref["g"][n])
        worst = max(worst, e)
        assert e <= (1e-5 if n.startswith("deconv/d_h4") else 1e-2), (tag, n, e)
    return worst


@pytest.fixture(scope="module")
def T():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    return Translator


def frames_for(B, H, W, seed):
    rng = np.random.default_rng(seed)
    return [(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1.0) for _ in range(3)]


def test_skipnew_switches_change_nothing_but_the_summation_order(T, monkeypatch):
    """The production net (64x64, df_dim 64) at BASELINE's B = 256: the bench's own launch shapes (512- / 256-image launches) -- position-major
    convs, the rectangle-ordered filter gradient, the LDS-resident transposed convs on 4x4 / 8x8 / 16x16 grids incl. the 512-image
    column-uniform kernel, the 3-channel direct kernels, convt3m -- all active on the defaults."""
    B = 256
    fr = frames_for(B, 64, 64, 3)
    make = lambda: T(64, 64, 64, 1024, max_batch=B)
    ref = run_step(make, fr, {}, monkeypatch)
    again = run_step(make, fr, {}, monkeypatch)
    np.testing.assert_array_equal(ref["out"], again["out"])                     # the default path is bit-reproducible
    report = {}
    for name, values in SKIPNEW_MATRIX:
        for v in values:
            report[f"{name}={v}"] = check(ref, run_step(make, fr, {"CTX_" + name.upper(): v}, monkeypatch), f"{name}={v}")
    print("worst gradient rel-L2 vs the default switches:", {k: float(f"{v:.1e}") for k, v in report.items()})


def test_real_switches_change_nothing_but_the_summation_order(T, monkeypatch):
    """ContextAEReal 36x64 at B = 64 (192 encoder / 128 decoder images per launch)."""
    B = 64
    fr = frames_for(B, 36, 64, 4)
    make = lambda: T(36, 64, featsize=100, max_batch=B, variant="real")
    ref = run_step(make, fr, {}, monkeypatch)
    report = {}
    for name, values in REAL_MATRIX:
        for v in values:
            report[f"{name}={v}"] = check(ref, run_step(make, fr, {"CTX_" + name.upper(): v}, monkeypatch), f"{name}={v}")
    print("worst gradient rel-L2 vs the default switches:", {k: float(f"{v:.1e}") for k, v in report.items()})

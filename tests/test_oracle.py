"""Pins the CPU oracle (oracle/ctx_oracle.py).  The reference has no golden vectors for this path
(parity unpinned), so the oracle is pinned by (1) hand-computable known-answer
tests of TF's published SAME / conv2d_transpose index rules, (2) an independent torch-autograd
statement, (3) finite differences, (4) the structural facts the reference records, (5) a closed
form TF-Adam step."""
import numpy as np
import pytest
import torch

from oracle import ctx_oracle as o
from tests import _torch_ref as tr


# ----------------------------------------------------------------------------- structural facts
def test_param_count_and_flops_match_survey():
    cfg = o.SkipNewConfig()
    assert o.param_count(cfg) == 47_647_811
    assert o.flops_forward(cfg) == 2_367_291_392


def test_h3_shape_at_48_matches_notebook():
    # notebooks/reach.ipynb (JSON line 386) prints (100, 3, 3, 512) for h3 at 48x48
    cfg = o.SkipNewConfig(H=48, W=48)
    p = o.init_params(cfg, 0, np.float32)
    x = np.zeros((2, 48, 48, 3), np.float32)
    _, c = o.forward(p, x, x, x, cfg)
    assert c["e_src"][3].shape == (2, 3, 3, 512)
    assert p["conv/h4_lin/Matrix"].shape == (4608, 1024)


def test_param_names_are_tf_scopes():
    names = [n for n, _ in o.param_specs(o.SkipNewConfig())]
    assert names[0] == "conv_context/h0_conv/w" and names[1] == "conv_context/h0_conv/biases"
    assert "conv/h4_lin/Matrix" in names and "translate/trans_z/bias" in names
    assert dict(o.param_specs(o.SkipNewConfig()))["deconv/d_h1/w"] == (5, 5, 256, 1024)
    assert dict(o.param_specs(o.SkipNewConfig()))["deconv/d_h4/w"] == (5, 5, 3, 128)


# ----------------------------------------------------------------------------- KATs: index rules
def test_same_pad_is_asymmetric_for_even_input():
    assert o.same_pad(64) == (32, 1, 2)
    assert o.same_pad(6) == (3, 1, 2)
    assert o.same_pad(5) == (3, 2, 2)
    assert o.same_pad(36, 5, 1) == (36, 2, 2)


def test_conv_delta_image_kat():
    """Input delta at (y0,x0): out[i,j] = w[y0+1-2i, x0+1-2j] wherever that tap index is in 0..4
    (from y = 2i + ky - 1)."""
    H = 8
    w = np.arange(25, dtype=np.float64).reshape(5, 5, 1, 1) + 1
    for (y0, x0) in [(0, 0), (3, 4), (7, 7), (6, 1)]:
        x = np.zeros((1, H, H, 1))
        x[0, y0, x0, 0] = 1.0
        y = o.conv2d(x, w, np.zeros(1))
        exp = np.zeros((4, 4))
        for i in range(4):
            for j in range(4):
                ky, kx = y0 + 1 - 2 * i, x0 + 1 - 2 * j
                if 0 <= ky < 5 and 0 <= kx < 5:
                    exp[i, j] = w[ky, kx, 0, 0]
        np.testing.assert_array_equal(y[0, :, :, 0], exp)


def test_conv_corner_sees_one_row_of_padding_top_two_bottom():
    x = np.ones((1, 4, 4, 1))
    w = np.ones((5, 5, 1, 1))
    y = o.conv2d(x, w, np.zeros(1))[0, :, :, 0]
    # out(0,0): rows -1..3 -> 4 valid rows, cols likewise -> 16 ; out(1,1): rows 1..5 -> 3 valid -> 9
    np.testing.assert_array_equal(y, [[16, 12], [12, 9]])


def test_deconv_delta_kat():
    """Input delta at (i0,j0): out[2*i0+ky-1, 2*j0+kx-1] = w[ky,kx] (cropped to the output)."""
    h = 3
    w = (np.arange(25, dtype=np.float64).reshape(5, 5, 1, 1) + 1)
    for (i0, j0) in [(0, 0), (1, 2), (2, 2)]:
        x = np.zeros((1, h, h, 1))
        x[0, i0, j0, 0] = 1.0
        y = o.deconv2d(x, w, np.zeros(1), (2 * h, 2 * h))[0, :, :, 0]
        exp = np.zeros((2 * h, 2 * h))
        for ky in range(5):
            for kx in range(5):
                yy, xx = 2 * i0 + ky - 1, 2 * j0 + kx - 1
                if 0 <= yy < 2 * h and 0 <= xx < 2 * h:
                    exp[yy, xx] = w[ky, kx, 0, 0]
        np.testing.assert_array_equal(y, exp)


def test_deconv_is_adjoint_of_conv():
    """conv2d_transpose is DEFINED as the input-gradient of the SAME conv: <conv(x), y> == <x, deconv(y)>
    with the same filter read as [k,k,in_of_conv,out_of_conv] == [k,k,out_of_deconv,in_of_deconv]."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 8, 12, 3))
    y = rng.standard_normal((2, 4, 6, 5))
    w = rng.standard_normal((5, 5, 3, 5))
    lhs = np.sum(o.conv2d(x, w, np.zeros(5)) * y)
    rhs = np.sum(x * o.deconv2d(y, w, np.zeros(3), (8, 12)))
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)


def test_concat_order_decoder_then_skip():
    """w[..., :C] acts on the decoder stream, w[..., C:] on the skip (arm_shaping.py:1323)."""
    cfg = o.SkipNewConfig(H=16, W=16, df_dim=4, gf_dim=4, featsize=8)
    p = o.init_params(cfg, 1)
    rng = np.random.default_rng(2)
    x = [rng.uniform(-1, 1, (2, 16, 16, 3)) for _ in range(3)]
    base, _ = o.forward(p, *x, cfg)
    q = dict(p)
    wz = p["deconv/d_h4/w"].copy()
    wz[..., cfg.gf_dim:] = 0           # kill the skip half: output must change, and equal deconv of decoder half only
    q["deconv/d_h4/w"] = wz
    r, c = o.forward(q, *x, cfg)
    only_dec = o.deconv2d(c["d1"][3], wz[..., :cfg.gf_dim], p["deconv/d_h4/biases"], (16, 16))
    np.testing.assert_allclose(r["out"], only_dec, rtol=1e-12, atol=1e-14)
    assert np.abs(r["out"] - base["out"]).max() > 1e-6


def test_preprocess_u8():
    x = np.array([0, 1, 127, 128, 255], np.uint8)
    y = o.preprocess_u8(x)
    assert y.dtype == np.float32
    np.testing.assert_allclose(y, (x / 255.0 - 0.5) * 2, atol=2e-7)
    assert y[0] == -1.0 and y[-1] == 1.0


# ----------------------------------------------------------------------------- torch cross-check
def _torch_grads(cfg, p, src, ctx, tgt):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    res = tr.forward(tp, *(torch.tensor(a, dtype=torch.float64) for a in (src, ctx, tgt)), cfg.H, cfg.W, cfg.gf_dim)
    res["loss"].backward()
    return res, {k: v.grad.numpy() for k, v in tp.items()}


@pytest.mark.parametrize("H,W,d,F,B", [(16, 16, 4, 8, 3), (32, 16, 8, 16, 2), (48, 48, 4, 8, 2)])
def test_oracle_matches_independent_torch_autograd(H, W, d, F, B):
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=d, gf_dim=d, featsize=F)
    p = o.init_params(cfg, 3, np.float64, stddev=0.2)   # larger weights -> both lrelu branches live
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = np.random.default_rng(5).standard_normal(p[n].shape) * 0.1
    rng = np.random.default_rng(4)
    src, ctx, tgt = (rng.uniform(-1, 1, (B, H, W, 3)) for _ in range(3))
    res, c = o.forward(p, src, ctx, tgt, cfg)
    g = o.backward(p, c, cfg)
    tres, tg = _torch_grads(cfg, p, src, ctx, tgt)
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], tres[k].detach().numpy(), rtol=1e-9, atol=1e-11)
    for k in ["simloss", "recon1", "recon2", "loss"]:
        assert abs(res[k] - float(tres[k])) <= 1e-10 * abs(float(tres[k]))
    for k in g:
        scale = np.abs(tg[k]).max() + 1e-30
        assert np.abs(g[k] - tg[k]).max() <= 1e-9 * scale, k


def test_oracle_matches_torch_autograd_on_smooth_frames():
    """SURVEY 8(d)'s second input distribution (tests/_frames.py: low-frequency blobs, uint8-quantised, through the inference
    preprocessing): the two independent statements of the arithmetic agree on it as on noise."""
    from tests._frames import blob_frames
    H, W, d, F, B = 32, 32, 8, 16, 3
    cfg = o.SkipNewConfig(H=H, W=W, df_dim=d, gf_dim=d, featsize=F)
    p = o.init_params(cfg, 9, np.float64, stddev=0.2)
    rng = np.random.default_rng(21)
    src, ctx, tgt = (o.preprocess_u8(blob_frames(rng, B, H, W)).astype(np.float64) for _ in range(3))
    assert np.abs(np.diff(src, axis=2)).mean() < 0.1            # smooth: neighbours differ by a few grey levels (noise: 0.67)
    res, c = o.forward(p, src, ctx, tgt, cfg)
    g = o.backward(p, c, cfg)
    tres, tg = _torch_grads(cfg, p, src, ctx, tgt)
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], tres[k].detach().numpy(), rtol=1e-9, atol=1e-11)
    for k in ["simloss", "recon1", "recon2", "loss"]:
        assert abs(res[k] - float(tres[k])) <= 1e-10 * abs(float(tres[k]))
    for k in g:
        assert np.abs(g[k] - tg[k]).max() <= 1e-9 * (np.abs(tg[k]).max() + 1e-30), k


def test_finite_difference_gradient():
    cfg = o.SkipNewConfig(H=16, W=16, df_dim=4, gf_dim=4, featsize=8)
    p = o.init_params(cfg, 7, np.float64, stddev=0.2)
    rng = np.random.default_rng(8)
    src, ctx, tgt = (rng.uniform(-1, 1, (2, 16, 16, 3)) for _ in range(3))
    res, c = o.forward(p, src, ctx, tgt, cfg)
    g = o.backward(p, c, cfg)
    eps = 1e-6
    for name in ["conv_context/h1_conv/w", "conv/h0_conv/w", "conv/hz_lin/Matrix", "translate/trans_h0/bias",
                 "deconv/d_h2/w", "deconv/d_h4/biases", "conv_context/h3_conv/biases", "deconv/d_h0_lin/Matrix"]:
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        q = {k: v.copy() for k, v in p.items()}
        q[name][idx] += eps
        lp = o.forward(q, src, ctx, tgt, cfg)[0]["loss"]
        q[name][idx] -= 2 * eps
        lm = o.forward(q, src, ctx, tgt, cfg)[0]["loss"]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[name][idx]) <= 1e-5 * max(1.0, abs(fd)), (name, fd, g[name][idx])


def test_data_parallel_shard_gradients_sum_to_full_batch():
    """recon terms are sums, simloss is a mean over the GLOBAL batch."""
    cfg = o.SkipNewConfig(H=16, W=16, df_dim=4, gf_dim=4, featsize=8)
    p = o.init_params(cfg, 9, np.float64, stddev=0.2)
    rng = np.random.default_rng(10)
    src, ctx, tgt = (rng.uniform(-1, 1, (4, 16, 16, 3)) for _ in range(3))
    _, c = o.forward(p, src, ctx, tgt, cfg)
    full = o.backward(p, c, cfg)
    parts = []
    for sl in (slice(0, 2), slice(2, 4)):
        _, cs = o.forward(p, src[sl], ctx[sl], tgt[sl], cfg)
        parts.append(o.backward(p, cs, cfg, sim_batch=4))
    for k in full:
        np.testing.assert_allclose(parts[0][k] + parts[1][k], full[k], rtol=1e-9, atol=1e-12)


# ----------------------------------------------------------------------------- Adam
def test_tf_adam_closed_form_first_steps():
    """Step 1 of TF Adam from m=v=0: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2
    => theta -= lr * g / (|g| + eps*sqrt(1-b2)) ... checked against the literal formula."""
    g = {"a": np.array([0.5, -2.0, 1e-9, 0.0])}
    p = {"a": np.zeros(4)}
    m = {"a": np.zeros(4)}
    v = {"a": np.zeros(4)}
    lr, b1, b2, eps = 1e-4, 0.9, 0.999, 1e-8
    o.adam_step(p, g, m, v, 1, lr)
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    exp = -lr_t * (0.1 * g["a"]) / (np.sqrt(0.001 * g["a"] ** 2) + eps)
    np.testing.assert_allclose(p["a"], exp, rtol=1e-12, atol=0)
    # eps is OUTSIDE the bias correction: for |g| >> eps the first step is ~ -lr*sign(g)
    assert abs(p["a"][0] + lr) < 1e-9 and abs(p["a"][1] - lr) < 1e-9
    # and differs from torch.optim.Adam's (eps inside) for tiny g
    torch_style = -lr * g["a"][2] / (abs(g["a"][2]) + eps)
    assert abs(p["a"][2] - torch_style) > 1e-6 * lr
    o.adam_step(p, g, m, v, 2, lr)
    np.testing.assert_allclose(m["a"], (1 - 0.9 ** 2) * g["a"], rtol=1e-12)


# ----------------------------------------------------------------------------- call-site semantics
def test_translate_and_encode_call_sites():
    cfg = o.SkipNewConfig(H=16, W=16, df_dim=4, gf_dim=4, featsize=8)
    p = o.init_params(cfg, 11, np.float32, stddev=0.2)
    rng = np.random.default_rng(12)
    frames = rng.integers(0, 256, (5, 16, 16, 3), dtype=np.uint8)
    ctx0 = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    pred, feat = o.translate(p, frames, ctx0, cfg)
    assert pred.shape == (5, 16, 16, 3) and feat.shape == (5, 8)
    pred_b, feat_b = o.translate(p, frames, np.broadcast_to(ctx0, frames.shape), cfg)
    np.testing.assert_array_equal(pred, pred_b)
    f, x = o.encode(p, frames, cfg)
    res, _ = o.forward(p, x, np.broadcast_to(x[0], x.shape), x, cfg)
    np.testing.assert_array_equal(f, res["input_z"])


def test_reward_costs_and_apply():
    rng = np.random.default_rng(0)
    feats, means = rng.standard_normal((25, 8)), rng.standard_normal((25, 8))
    fr, im = rng.standard_normal((25, 4, 4, 3)), rng.standard_normal((25, 4, 4, 3))
    c = o.reward_costs(feats, fr, means, im, 0.5)
    j = 7
    assert abs(c[j] - (((means[j] - feats[j]) ** 2).sum() + 0.5 * ((im[j] - fr[j]) ** 2).sum())) < 1e-12
    r = o.apply_costs(np.zeros(50), c)
    assert r[2 * j + 1] == -c[j] * 49 and r[2 * j] == 0 and r[1] == 0

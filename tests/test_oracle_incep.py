"""oracle/ctx_oracle_incep.py (ContextAEInception2, arm_shaping.py:1786-1894): shapes the reference records, hand KATs
for the k = 3 SAME rules, and an independent torch-autograd statement (outputs and every gradient, float64)."""
import numpy as np
import pytest
import torch

from oracle import ctx_oracle_incep as oi
from tests import _torch_ref as tr


def test_param_inventory_and_sizes():
    cfg = oi.Incep2Config()                     # 125x125 frames -> Mixed_7c 2x2x2048
    assert cfg.sizes == [(2, 2), (1, 1), (1, 1), (1, 1)]
    assert oi.Incep2Config(H=8, W=8).sizes == [(8, 8), (4, 4), (4, 4), (2, 2)]      # 299x299 (inception_v3_test.py:45-54)
    specs = dict(oi.param_specs(cfg))
    assert specs["conv/h0_conv/w"] == (3, 3, 2048, 1024) and specs["deconv/d_h4/w"] == (3, 3, 2048, 2048)
    assert specs["deconv/d_h1/w"] == (3, 3, 512, 1024) and specs["conv_context/h4_lin/Matrix"] == (512, 1024)
    names = [n for n, _ in oi.param_specs(cfg)]
    assert names[0].startswith("conv_context/") and names[12].startswith("conv/") and names[-1] == "deconv/d_h4/biases"


def test_k3_same_rules_by_hand():
    # stride 2 on an even grid: pad (0, 1): output (i) reads rows 2i .. 2i+2
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    w = np.zeros((3, 3, 1, 1)); w[0, 0] = 1                                  # picks x[2i, 2j]
    np.testing.assert_array_equal(oi.conv2d(x, w, np.zeros(1), 2)[0, :, :, 0], x[0, ::2, ::2, 0])
    w = np.zeros((3, 3, 1, 1)); w[2, 2] = 1                                  # x[2i+2, 2j+2], zero past the edge
    np.testing.assert_array_equal(oi.conv2d(x, w, np.zeros(1), 2)[0, :, :, 0], [[10, 0], [0, 0]])
    # stride 1: pad (1, 1): centre tap is the identity
    w = np.zeros((3, 3, 1, 1)); w[1, 1] = 1
    np.testing.assert_array_equal(oi.conv2d(x, w, np.zeros(1), 1), x)
    # 1x1 grid under stride 2: stays 1x1, only the centre tap sees data
    x1 = np.array(7.0).reshape(1, 1, 1, 1)
    w = np.arange(9, dtype=np.float64).reshape(3, 3, 1, 1)
    assert oi.conv2d(x1, w, np.zeros(1), 2).shape == (1, 1, 1, 1) and oi.conv2d(x1, w, np.zeros(1), 2)[0, 0, 0, 0] == 7 * 4
    assert oi.deconv2d(x1, w, np.zeros(1), (1, 1), 2)[0, 0, 0, 0] == 7 * 4
    # transpose of the stride-2 case: input pixel (i) lands on rows 2i + ky
    y = oi.deconv2d(np.ones((1, 1, 1, 1)), w, np.zeros(1), (2, 2), 2)
    np.testing.assert_array_equal(y[0, :, :, 0], [[0, 1], [3, 4]])


@pytest.mark.parametrize("H,W,C,filters", [(2, 2, 24, (16, 16, 8, 8)), (8, 8, 12, (8, 8, 4, 4)), (4, 6, 6, (8, 4, 4, 4))])
def test_oracle_matches_torch_autograd(H, W, C, filters):
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=16, filters=filters)
    p = oi.init_params(cfg, 3, np.float64, stddev=0.2)
    rng = np.random.default_rng(0)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = rng.standard_normal(p[n].shape) * 0.1
    B = 3
    src, ctx, tgt = (rng.standard_normal((B, H, W, C)) for _ in range(3))
    res, c = oi.forward(p, src, ctx, tgt, cfg)
    g = oi.backward(p, c, cfg)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tres = tr.forward_incep2(tp, *(torch.tensor(a) for a in (src, ctx, tgt)), H, W, cfg.strides, cfg.filters)
    for k in ("input_z", "translated_z", "out", "out2", "simloss", "recon1", "recon2", "loss"):
        np.testing.assert_allclose(res[k], tres[k].detach().numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
    tres["loss"].backward()
    for n in g:
        np.testing.assert_allclose(g[n], tp[n].grad.numpy(), rtol=1e-8, atol=1e-10, err_msg=n)


def test_finite_difference_spot_check():
    cfg = oi.Incep2Config(H=2, W=2, C=8, featsize=8, filters=(8, 8, 4, 4))
    p = oi.init_params(cfg, 1, np.float64, stddev=0.3)
    rng = np.random.default_rng(1)
    src, ctx, tgt = (rng.standard_normal((2, 2, 2, 8)) for _ in range(3))
    res, c = oi.forward(p, src, ctx, tgt, cfg)
    g = oi.backward(p, c, cfg)
    for name in ("conv/h1_conv/w", "deconv/d_h3/w", "conv_context/h0_conv/w", "deconv/d_h4/biases", "translate/trans_h0/Matrix"):
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        eps = 1e-6
        q = {k: v.copy() for k, v in p.items()}
        q[name][idx] += eps
        lp = oi.forward(q, src, ctx, tgt, cfg)[0]["loss"]
        q[name][idx] -= 2 * eps
        lm = oi.forward(q, src, ctx, tgt, cfg)[0]["loss"]
        assert abs((lp - lm) / (2 * eps) - g[name][idx]) <= 1e-5 * max(1.0, abs(g[name][idx])), name


@pytest.mark.parametrize("ablation", ["L2", "L2L3", "L1"])
def test_oracle_loss_ablations_match_torch_autograd(ablation):
    """The loss switch of ablations_code/ablations.py:175-182 on ContextAEInception2's graph: `loss` and every gradient for the
    selected terms against torch autograd of that sum."""
    from oracle.ctx_oracle import LOSS_ABLATIONS
    H, W, C, filters, B = 4, 4, 8, (8, 8, 4, 4), 2
    cfg = oi.Incep2Config(H=H, W=W, C=C, featsize=16, filters=filters)
    p = oi.init_params(cfg, 5, np.float64, stddev=0.2)
    rng = np.random.default_rng(6)
    src, ctx, tgt = (rng.standard_normal((B, H, W, C)) for _ in range(3))
    res, c = oi.forward(p, src, ctx, tgt, cfg, ablation_type=ablation)
    g = oi.backward(p, c, cfg)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tres = tr.forward_incep2(tp, *(torch.tensor(a) for a in (src, ctx, tgt)), H, W, cfg.strides, cfg.filters)
    tloss = sum(tres[t] for t in LOSS_ABLATIONS[ablation])
    np.testing.assert_allclose(res["loss"], tloss.item(), rtol=1e-10)
    tloss.backward()
    for n in g:
        tg = tp[n].grad.numpy() if tp[n].grad is not None else np.zeros_like(p[n])
        np.testing.assert_allclose(g[n], tg, rtol=1e-8, atol=1e-10, err_msg=n)

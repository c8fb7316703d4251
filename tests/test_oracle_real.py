"""Pins oracle/ctx_oracle_real.py (ContextAEReal, arm_shaping.py:1599-1684) the same way as the ContextSkipNew
oracle: structural facts, stride-1 SAME / conv2d_transpose KATs, an independent torch-autograd statement."""
import numpy as np
import pytest
import torch

from oracle import ctx_oracle as o
from oracle import ctx_oracle_real as r
from tests import _torch_ref as tr


def test_shapes_at_the_reference_size():
    cfg = r.RealConfig()                                   # 36x64 (run_trpo_sweep_ours.py:64)
    assert cfg.sizes == [(36, 64), (18, 32), (18, 32), (9, 16)]
    specs = dict(r.param_specs(cfg))
    assert specs["conv/h4_lin/Matrix"] == (9 * 16 * 8, 100)
    assert specs["deconv/d_h1/w"] == (5, 5, 16, 16) and specs["deconv/d_h4/w"] == (5, 5, 3, 64)
    assert specs["translate/trans_h0/Matrix"] == (200, 100)
    p = r.init_params(cfg, 0, np.float32)
    x = np.zeros((2, 36, 64, 3), np.float32)
    res, c = r.forward(p, x, x, x, cfg)
    assert [a.shape[1:] for a in c["e_src"][:4]] == [(36, 64, 32), (18, 32, 16), (18, 32, 16), (9, 16, 8)]
    assert res["out"].shape == (2, 36, 64, 3) and res["input_z"].shape == (2, 100)


def test_stride1_same_pad_and_transpose_kat():
    assert o.same_pad(36, 5, 1) == (36, 2, 2)
    w = np.arange(25, dtype=np.float64).reshape(5, 5, 1, 1) + 1
    x = np.zeros((1, 6, 6, 1))
    x[0, 2, 3, 0] = 1.0
    y = o.conv2d(x, w, np.zeros(1), s=1)[0, :, :, 0]      # out[i,j] = w[y0+2-i, x0+2-j]
    for i in range(6):
        for j in range(6):
            ky, kx = 2 + 2 - i, 3 + 2 - j
            exp = w[ky, kx, 0, 0] if 0 <= ky < 5 and 0 <= kx < 5 else 0.0
            assert y[i, j] == exp
    d = o.deconv2d(x, w, np.zeros(1), (6, 6), s=1)[0, :, :, 0]   # out[i0+ky-2, j0+kx-2] = w[ky,kx]
    for ky in range(5):
        for kx in range(5):
            yy, xx = 2 + ky - 2, 3 + kx - 2
            if 0 <= yy < 6 and 0 <= xx < 6:
                assert d[yy, xx] == w[ky, kx, 0, 0]
    rng = np.random.default_rng(0)
    a, b2 = rng.standard_normal((2, 6, 8, 3)), rng.standard_normal((2, 6, 8, 4))
    ww = rng.standard_normal((5, 5, 3, 4))
    lhs = np.sum(o.conv2d(a, ww, np.zeros(4), s=1) * b2)
    rhs = np.sum(a * o.deconv2d(b2, ww, np.zeros(3), (6, 8), s=1))
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)


@pytest.mark.parametrize("H,W,B", [(36, 64, 2), (12, 8, 3)])
def test_real_oracle_matches_torch_autograd(H, W, B):
    cfg = r.RealConfig(H=H, W=W)
    p = r.init_params(cfg, 3, np.float64, stddev=0.2)
    brng = np.random.default_rng(5)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * 0.1
    rng = np.random.default_rng(4)
    src, ctx, tgt = (rng.uniform(-1, 1, (B, H, W, 3)) for _ in range(3))
    res, c = r.forward(p, src, ctx, tgt, cfg)
    g = r.backward(p, c, cfg)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tres = tr.forward_real(tp, *(torch.tensor(a, dtype=torch.float64) for a in (src, ctx, tgt)), H, W)
    tres["loss"].backward()
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], tres[k].detach().numpy(), rtol=1e-9, atol=1e-11)
    for k in ["simloss", "recon1", "recon2", "loss"]:
        assert abs(res[k] - tres[k].item()) <= 1e-10 * abs(tres[k].item())
    for k in g:
        tg = tp[k].grad.numpy()
        assert np.abs(g[k] - tg).max() <= 1e-9 * (np.abs(tg).max() + 1e-30), k


def _load_real_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "real_f100_36x64_b3.npz"))
    H, W, C, F = (int(v) for v in z["cfg"])
    cfg = r.RealConfig(H=H, W=W, C=C, featsize=F)
    p = r.init_params(cfg, int(z["pseed"]), np.float64, stddev=float(z["stddev"]))
    brng = np.random.default_rng(int(z["pseed"]) + 1)
    for n in p:
        if n.endswith("bias") or n.endswith("biases"):
            p[n] = brng.standard_normal(p[n].shape) * float(z["stddev"])
    return z, cfg, p


def test_oracle_reproduces_real_golden():
    from oracle import ctx_oracle as o
    z, cfg, p = _load_real_golden()
    dg = lambda a: np.array([np.sum(a), np.abs(a).sum(), np.sqrt((np.asarray(a, np.float64) ** 2).sum())])
    np.testing.assert_allclose(dg(r.flatten(p, cfg)), z["param_digest"], rtol=1e-12)
    src, ctx, tgt = (o.preprocess_u8(z[k]).astype(np.float64) for k in ("src_u8", "ctx_u8", "tgt_u8"))
    res, c = r.forward(p, src, ctx, tgt, cfg)
    for k in ["input_z", "translated_z", "out", "out2"]:
        np.testing.assert_allclose(res[k], z[k], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([res["loss"], res["simloss"], res["recon1"], res["recon2"]], z["scalars"], rtol=1e-12)
    g = r.backward(p, c, cfg)
    for i, (n, _) in enumerate(r.param_specs(cfg)):
        np.testing.assert_allclose(dg(g[n]), z["grad_digest"][i], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("ablation", ["L2", "L2L3", "L1"])
def test_real_oracle_loss_ablations_match_torch_autograd(ablation):
    """ablations_code/ablations.py:477-484 (ContextAESweep / ContextAEPushReal carry the same switch, :175-182): the oracle's
    `loss` and every gradient for the selected terms against torch autograd of that sum."""
    H, W, B = 12, 8, 3
    cfg = r.RealConfig(H=H, W=W)
    p = r.init_params(cfg, 13, np.float64, stddev=0.2)
    rng = np.random.default_rng(14)
    src, ctx, tgt = (rng.uniform(-1, 1, (B, H, W, 3)) for _ in range(3))
    res, c = r.forward(p, src, ctx, tgt, cfg, ablation_type=ablation)
    g = r.backward(p, c, cfg)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tres = tr.forward_real(tp, *(torch.tensor(a, dtype=torch.float64) for a in (src, ctx, tgt)), H, W)
    tloss = sum(tres[t] for t in o.LOSS_ABLATIONS[ablation])
    assert abs(res["loss"] - tloss.item()) <= 1e-10 * abs(tloss.item())
    for k in ("simloss", "recon1", "recon2"):                                   # reported whatever the switch
        assert abs(res[k] - tres[k].item()) <= 1e-10 * abs(tres[k].item())
    tloss.backward()
    for k in g:
        tg = tp[k].grad.numpy() if tp[k].grad is not None else np.zeros_like(p[k])
        assert np.abs(g[k] - tg).max() <= 1e-9 * (np.abs(tg).max() + 1e-30) + 1e-30, k

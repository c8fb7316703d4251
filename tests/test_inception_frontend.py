"""The Inception-v3 front end (imitation_from_observation_amd/inception_frontend.py + csrc/cnn.cpp).
CPU: the op-list layout against what the reference's test holds (end-point shapes, variable total).
GPU: every end point against oracle/inception_oracle.py on the same synthetic variables."""
import numpy as np
import pytest

from imitation_from_observation_amd.inception_frontend import _Layout
from oracle import inception_oracle as io
from tests.test_oracle_inception import REF_SHAPES


def test_layout_matches_reference_endpoints_and_variable_total():
    for merge in (False, True):
        _check_layout(_Layout(299, 299, merge))
    assert len(_Layout(299, 299, True).ops) == 88 and len(_Layout(299, 299, False).ops) == 107   # 19 launches fewer with the 1x1 heads merged
    assert _Layout(125, 125).out[1:] == (2, 2, 2048)


def _check_layout(lay):
    assert list(lay.endpoints) == list(REF_SHAPES)
    for k, v in REF_SHAPES.items():
        assert lay.endpoints[k][1:] == v, k                                   # nets/inception_v3_test.py:87-104
    assert sum(int(np.prod(c["k"])) * c["cin"] * c["cout"] + 3 * c["cout"] for c in lay.convs) == 21802784     # :120
    # same variables, same order of creation, as the oracle's walk of the reference graph
    names = [c["scope"] for c in lay.convs]
    assert names == [n[:-len("/weights")] for n, _ in io.param_specs() if n.endswith("/weights")]
    # every concat is a set of adjacent, non-overlapping channel slices that tile the block output
    by_dst = {}
    for op in lay.ops:
        width = (op.get("nsplit") or op["cout"]) if op["kind"] == 0 else lay.bufs[op["src"]][2]
        by_dst.setdefault(op["dst"], []).append((op["dst_ch0"], op["dst_ch0"] + width))
        if op.get("nsplit"):
            by_dst.setdefault(op["dst2"], []).append((op["dst2_ch0"], op["dst2_ch0"] + op["cout"] - op["nsplit"]))
    for name in ("Mixed_5b", "Mixed_6a", "Mixed_6e", "Mixed_7a", "Mixed_7c"):
        bid, _, _, c = lay.endpoints[name]
        sl = sorted(by_dst[bid])
        assert sl[0][0] == 0 and sl[-1][1] == c and all(a[1] == b[0] for a, b in zip(sl, sl[1:])), name


@pytest.fixture(scope="module")
def front():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd.inception_frontend import InceptionFrontend
    f = InceptionFrontend(125, 125, max_images=3)
    tree = f.init_synthetic(0)
    yield f, {k: v.astype(np.float64) for k, v in tree.items()}
    f.close()


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.gpu
def test_every_endpoint_matches_the_oracle(front):
    from oracle.ctx_oracle import preprocess_u8
    f, p = front
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, (2, 125, 125, 3), dtype=np.uint8)
    feat = f.features(u8)
    ref = io.forward(p, preprocess_u8(u8).astype(np.float64))
    assert feat.shape == (2, 2, 2, 2048)
    for name, r in ref.items():
        got = f.endpoint(name, 2)
        assert got.shape == r.shape, name
        assert relmax(got, r) < 1e-4, name                                   # fp32 through <= 47 conv layers: ~1e-6
    np.testing.assert_array_equal(feat, f.endpoint("Mixed_7c", 2))
    assert relmax(feat, ref["Mixed_7c"]) < 1e-4


@pytest.mark.gpu
def test_chunking_device_entry_and_padding_stay_clean(front):
    import torch
    from oracle.ctx_oracle import preprocess_u8
    f, p = front
    rng = np.random.default_rng(1)
    u8 = rng.integers(0, 256, (7, 125, 125, 3), dtype=np.uint8)            # 7 > max_images = 3: three passes
    feat = f.features(u8)
    one = np.concatenate([f.features(u8[i:i + 1]) for i in range(7)])
    np.testing.assert_array_equal(feat, one)                                 # per-image independent, bit for bit
    x = torch.from_numpy(preprocess_u8(u8[:3])).cuda()
    torch.cuda.synchronize()
    d = f.features_dev(x.data_ptr(), 3)
    f.sync()
    assert d
    np.testing.assert_array_equal(f.endpoint("Mixed_7c", 3), feat[:3])
    # padded channels (80 -> 96) were never written
    bid = f.endpoints["Conv2d_3b_1x1"][0]
    raw = np.empty((3,) + f._bufs[bid], np.float32)
    import ctypes
    f._ck(f._lib.ctx_cnn_read_buffer(f._h, bid, 3, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    assert not raw[..., 80:].any() and raw[..., :80].any()


@pytest.mark.gpu
def test_position_major_batch_matches_per_image_pass():
    """>= 64 images per pass take the position-major conv for the SAME-padded layers (taps outside the grid are skipped:
    on the 2x2 grid of Mixed_7 that is 5 of the 9 taps of a 3x3 kernel): same features as the small-batch path, and the oracle's."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd.inception_frontend import InceptionFrontend
    from oracle.ctx_oracle import preprocess_u8
    u8 = np.random.default_rng(5).integers(0, 256, (66, 125, 125, 3), dtype=np.uint8)
    with InceptionFrontend(125, 125, max_images=66) as big, InceptionFrontend(125, 125, max_images=2) as small:
        tree = big.init_synthetic(7)
        small.set_variables(tree)
        fb, fs = big.features(u8), small.features(u8)
        assert relmax(fb, fs) < 1e-5
        ref = io.forward({k: v.astype(np.float64) for k, v in tree.items()}, preprocess_u8(u8[:2]).astype(np.float64))["Mixed_7c"]
        assert relmax(fb[:2], ref) < 1e-4


@pytest.mark.gpu
def test_split_precision_front_end():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd.inception_frontend import InceptionFrontend
    from oracle.ctx_oracle import preprocess_u8
    with InceptionFrontend(125, 125, max_images=2, precision="bf16x3") as f:
        p = {k: v.astype(np.float64) for k, v in f.init_synthetic(3).items()}
        u8 = np.random.default_rng(2).integers(0, 256, (2, 125, 125, 3), dtype=np.uint8)
        assert relmax(f.features(u8), io.forward(p, preprocess_u8(u8).astype(np.float64))["Mixed_7c"]) < 1e-3


@pytest.mark.gpu
def test_three_channel_input_convs_any_kernel():
    """The executor stores its 3-channel input 4 wide and runs every conv that reads it on the cin = 3 gather (`stem4_ok`,
    cnn.cpp).  Inception only exercises 3x3 VALID stride 2 there; this op list reads the frames with a 5x5 SAME stride-2 and a
    3x3 SAME stride-1 conv whose filters carry junk in the 29 padding rows per tap (never multiplied), and reads buffer 0 back."""
    import ctypes
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import _lib
    from imitation_from_observation_amd._lib import CnnBuf, CnnOp
    from oracle.ctx_oracle import preprocess_u8
    from tests._torch_ref import tf_conv_ks
    lib = _lib.load()
    H, W, n = 18, 20, 5
    bufs = [(H, W, 32), (9, 10, 32), (H, W, 64)]
    rng = np.random.default_rng(11)
    blob, ops, ws = [], [], []
    off = 0
    for dst, k, s, cout in ((1, 5, 2, 32), (2, 3, 1, 40)):
        w = rng.standard_normal((k, k, 32, cout)).astype(np.float32) * 0.2
        b = rng.standard_normal(cout).astype(np.float32)
        ops.append(CnnOp(_lib.CTX_CNN_CONV, 0, dst, 0, k, k, s, 1, cout, 0, off, off + w.size))
        blob += [w.ravel(), b, np.zeros((-(w.size + cout)) % 4, np.float32)]
        off += (w.size + cout + 3) // 4 * 4
        ws.append((w, b, s))
    blob = np.concatenate(blob)
    h = ctypes.c_void_p()
    cb = (CnnBuf * 3)(*[CnnBuf(*x) for x in bufs])
    co = (CnnOp * 2)(*ops)
    assert lib.ctx_cnn_create(cb, 3, co, 2, blob.size, 8, 0, 0, ctypes.c_void_p(0), ctypes.byref(h)) == _lib.CTX_OK
    try:
        FP = ctypes.POINTER(ctypes.c_float)
        assert lib.ctx_cnn_set_weights(h, blob.ctypes.data_as(FP), blob.size) == _lib.CTX_OK
        u8 = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        assert lib.ctx_cnn_forward_u8(h, u8.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n, None) == _lib.CTX_OK
        x = preprocess_u8(u8)
        raw0 = np.empty((n, H, W, 32), np.float32)
        assert lib.ctx_cnn_read_buffer(h, 0, n, raw0.ctypes.data_as(FP)) == _lib.CTX_OK
        np.testing.assert_array_equal(raw0[..., :3], x)
        assert not raw0[..., 3:].any()
        for bid, (w, b, s) in zip((1, 2), ws):
            got = np.empty((n,) + bufs[bid], np.float32)
            assert lib.ctx_cnn_read_buffer(h, bid, n, got.ctypes.data_as(FP)) == _lib.CTX_OK
            cout = w.shape[3]
            want = torch.relu(tf_conv_ks(torch.from_numpy(x).double(), torch.from_numpy(w[:, :, :3]).double(), torch.from_numpy(b).double(), s)).numpy()
            np.testing.assert_allclose(got[..., :cout], want, rtol=1e-5, atol=1e-5)
            assert not got[..., cout:].any()
    finally:
        lib.ctx_cnn_destroy(h)


@pytest.mark.gpu
def test_inception_translator_trains_without_host_feature_maps():
    """mode 'oursinception' on the product path (scripts/train_script.py:98-114, 163, 176): uint8 frame triples -> frozen Inception-v3
    -> ContextAEInception2 train step / validation fetch, the feature maps handed over as device pointers; scalars against the two
    oracles composed on the CPU, for the sampler's lists and for another (strides, kernels, filters)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd.oursinception import InceptionTranslator
    from oracle import ctx_oracle as o
    from oracle import ctx_oracle_incep as oi
    rng = np.random.default_rng(21)
    B, S = 3, 125
    frames = [rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8) for _ in range(3)]
    for lists in (None, dict(strides=[2, 1, 1, 1], kernels=[3, 1, 3, 3], filters=[64, 32, 32, 64])):
        kw = lists or {}
        with InceptionTranslator((S, S), max_batch=B, df_dim=4, **kw) as it:
            ip = {k: v.astype(np.float64) for k, v in it.front.init_synthetic(4).items()}
            cfg = oi.Incep2Config(H=2, W=2, C=2048, **({k: tuple(v) for k, v in lists.items()} if lists else dict(filters=(64, 64, 32, 32))))
            tp = oi.init_params(cfg, 9, np.float32, stddev=0.01)
            it.tr.set_params(tp)
            f = [io.forward(ip, o.preprocess_u8(x).astype(np.float64))["Mixed_7c"] for x in frames]
            res, _ = oi.forward({k: v.astype(np.float64) for k, v in tp.items()}, *f, cfg)
            ev = it.evaluate_u8(*frames)
            for k in ("loss", "simloss", "recon1", "recon2"):
                assert abs(ev[k] - res[k]) <= 1e-3 * abs(res[k]) + 1e-6, k
            assert np.abs(ev["out"] - res["out"]).max() <= 1e-3 * np.abs(res["out"]).max()
            assert np.abs(ev["tgt"] - f[2]).max() <= 1e-3 * np.abs(f[2]).max()
            l0 = it.train_step_u8(*frames, lr=1e-4)["loss"]
            assert abs(l0 - res["loss"]) <= 1e-3 * abs(res["loss"])
            for _ in range(3):
                l1 = it.train_step_u8(*frames, lr=1e-4)["loss"]
            assert l1 < l0
            pred, feat = it.translate(frames[0], frames[1][0])
            c0 = np.broadcast_to(f[1][0], f[0].shape)
            assert pred.shape == (B, 2, 2, 2048) and feat.shape == (B, 1024)

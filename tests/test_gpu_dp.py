"""The data-parallel trainer on the HIP engine with a one-rank RCCL group: the bucketed, overlapped all-reduce path
(CTX_DP_OVERLAP=1: a callback from inside the backward pass starts the all-reduce of the translate/deconv gradients while
the encoders' backward is still being enqueued) must leave exactly the parameters of the plain path.  Multi-rank
arithmetic is covered on CPU by tests/test_dp_gloo.py; this covers the stream / callback plumbing on the device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_overlapped_allreduce_path_equals_plain_path(monkeypatch):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd.dp import DataParallelTrainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B = 64
        g = torch.Generator(device="cuda").manual_seed(3)
        fr = [torch.rand((B, 32, 32, 3), device="cuda", generator=g) * 2 - 1 for _ in range(3)]
        out = {}
        for mode in ("plain", "overlap"):
            monkeypatch.setenv("CTX_DP_FORCE", "1")                       # run the collectives although world == 1
            monkeypatch.setenv("CTX_DP_OVERLAP", "1" if mode == "overlap" else "0")
            tr = DataParallelTrainer(32, 32, 32, 128, max_batch=B, device=0, seed=11)
            calls = []
            if mode == "overlap":
                real = tr.engine.forward_backward

                def spy(*a, bucket_cb=None, **k):
                    def cb(first, count):
                        calls.append((first, count))
                        bucket_cb(first, count)
                    return real(*a, bucket_cb=cb, **k)
                tr.engine.forward_backward = spy
            for _ in range(3):
                tr.step(*fr, lr=1e-3)
            sc = tr.scalars()
            out[mode] = (tr.translator.get_params_flat(), sc, calls)
            tr.translator.close()
        np.testing.assert_array_equal(out["plain"][0], out["overlap"][0])
        assert out["plain"][1] == out["overlap"][1]
        calls = out["overlap"][2]
        assert len(calls) == 3 and all(c == calls[0] for c in calls)
        first, count = calls[0]
        info = {n: o for n, _, o in DataParallelTrainer(32, 32, 32, 128, max_batch=1, device=0).translator.param_info()}
        assert first == info["translate/trans_h0/Matrix"] and first + count >= max(info.values())
    finally:
        dist.destroy_process_group()


def test_rccl_behind_the_c_abi_on_one_rank():
    """ctx_dp_* (SURVEY.md 8b): the library's own RCCL communicator.  On a one-rank communicator the SUM is the identity, so
    the two-bucket overlapped step (ctx_dp_train_step) and the phase form (ctx_dev_forward_backward -> ctx_dp_allreduce_grads
    -> ctx_dev_adam) must leave bit-for-bit the parameters and scalars of ctx_train_step; ctx_dp_init's broadcast must keep
    rank 0's parameters."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import CtxError, Translator
    from imitation_from_observation_amd.dp import RcclTrainer
    H, W, d, F, B = 32, 32, 32, 128, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    fr = [(torch.rand((B, H, W, 3), device="cuda", generator=g) * 2 - 1).contiguous() for _ in range(3)]
    torch.cuda.synchronize()
    host = [x.cpu().numpy() for x in fr]
    uid = Translator.dp_unique_id()
    assert len(uid) == 128 and any(uid)
    tr = RcclTrainer(H, W, d, F, max_batch=B, device=0, seed=21, rank=0, world=1, unique_id=uid)
    with Translator(H, W, d, F, max_batch=B) as ref, Translator(H, W, d, F, max_batch=B) as ph:
        p0 = tr.translator.get_params_flat()
        assert np.abs(p0).max() > 0                                   # the broadcast kept rank 0's parameters
        ref.set_params_flat(p0)
        ph.set_params_flat(p0)
        with pytest.raises(CtxError):
            ph.dp_allreduce_grads()                                   # no communicator on this handle yet
        ph.dp_init(Translator.dp_unique_id(), 0, 1)
        with pytest.raises(CtxError):
            ph.dp_init(uid, 0, 1)                                     # twice
        for it in range(3):
            sc = tr.step(*fr, lr=1e-3, scalars=True)
            sref = ref.train_step(*host, lr=1e-3)
            assert sc == sref, it
            ph.dev_forward_backward(fr[0].data_ptr(), fr[1].data_ptr(), fr[2].data_ptr(), B, sim_batch=B)
            ph.dp_allreduce_grads()
            ph.dev_adam(1e-3)
            assert ph.dp_scalars() == sref
        np.testing.assert_array_equal(tr.translator.get_params_flat(), ref.get_params_flat())
        np.testing.assert_array_equal(ph.get_params_flat(), ref.get_params_flat())
    tr.translator.close()


@pytest.mark.gpu
def test_frames_written_in_place_skip_the_staging_copy():
    """ctx_dev_frames (ABI 4): a caller that writes its shard straight into the handle's [tgt | src | ctx] slots and
    passes those pointers to ctx_dev_forward_backward / ctx_dev_train_step gets bit for bit what the copying path gives -- and a
    pointer that is NOT the slot is still copied (mixed: only ctx in place)."""
    import ctypes
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    H = W = 32
    B = 6
    g = torch.Generator(device="cuda").manual_seed(5)
    fr = [torch.rand((B, H, W, 3), device="cuda", generator=g) * 2 - 1 for _ in range(3)]      # src, ctx, tgt
    torch.cuda.synchronize()
    with Translator(H, W, 32, 128, max_batch=8) as a, Translator(H, W, 32, 128, max_batch=8) as b:
        a.init_params(3)
        b.set_params_flat(a.get_params_flat())
        a.dev_forward_backward(*(t.data_ptr() for t in fr), B)
        a.sync()
        slots = b.dev_frames(B)
        assert len(set(slots)) == 3 and b.dev_frames(B - 1) != slots                        # packed per batch size
        hip = ctypes.CDLL("libamdhip64.so")
        for dst, t in zip(slots, fr):
            assert hip.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(t.numel() * 4), 3) == 0   # hipMemcpyDeviceToDevice
        b.dev_forward_backward(*slots, B)
        b.sync()
        np.testing.assert_array_equal(a.get_grads_flat(), b.get_grads_flat())
        assert a.dev_scalars() == b.dev_scalars()
        # mixed: src and tgt from the caller's tensors (copied), ctx in place
        b.dev_train_step(fr[0].data_ptr(), slots[1], fr[2].data_ptr(), B, lr=1e-3)
        a.dev_train_step(*(t.data_ptr() for t in fr), B, lr=1e-3)
        a.sync(); b.sync()
        np.testing.assert_array_equal(a.get_params_flat(), b.get_params_flat())

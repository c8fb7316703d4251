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

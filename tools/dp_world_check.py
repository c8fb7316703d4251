"""One rank of tools/scale_check.sh: initialises the C-ABI data-parallel group (ctx_dp_init over RCCL) the way bench.py does and
checks that ctx_dp_world reports this rank and the launcher's world size; then one ctx_dp_train_step at a tiny size."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd.dp import RcclTrainer  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("gloo", rank=rank, world_size=world)          # (ships the 128-byte rendezvous blob only)
tr = RcclTrainer(32, 32, 32, 128, max_batch=8, device=local, seed=1, rank=rank, world=world)
r, w = tr.translator.dp_world()
assert (r, w) == (rank, world), ((r, w), (rank, world))
x = [torch.rand(8, 32, 32, 3, device="cuda") * 2 - 1 for _ in range(3)]
torch.cuda.synchronize()
sc = tr.step(*x, lr=1e-4, scalars=True)
p = torch.from_numpy(tr.translator.get_params_flat())
ref = p.clone()
dist.broadcast(ref, src=0)
assert torch.equal(p, ref), "replicas differ after one step"
print(f"rank {rank}: world ok ({r}, {w}), loss {sc['loss']:.4f}, replicas identical", flush=True)
dist.destroy_process_group()

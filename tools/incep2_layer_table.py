"""Development tool: per-launch table of one ContextAEInception2 training step (config 4's translator: 2x2x2048 feature maps, B = 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
B, h, w, c = 64, 2, 2, 2048
g = torch.Generator(device="cuda").manual_seed(0)
fr = [torch.rand((B, h, w, c), device="cuda", generator=g) for _ in range(3)]
with Translator(h, w, 64, 1024, max_batch=B, variant="inception2", C=c) as tr:
    tr.init_params(0)
    print("params", tr.n_params)
    for _ in range(2):
        tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
        tr.dev_adam(1e-4)
    tr.sync()
    ents = tr.profile_step(*(t.data_ptr() for t in fr), B, iters=3)
    print("total", sum(e["ms"] for e in ents))
    for e in sorted(ents, key=lambda e: -e["ms"])[:40]:
        print(f"{e['name']:32s} {e['kernel']:34s} {e['ms']:8.3f} ms {e['flops']/e['ms']/1e9 if e['ms'] else 0:7.1f} TF/s")

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator
B = 256
g = torch.Generator(device="cuda").manual_seed(0)
fr = [torch.rand((B, 36, 64, 3), device="cuda", generator=g) * 2 - 1 for _ in range(3)]
with Translator(36, 64, featsize=100, max_batch=B, variant="real") as tr:
    tr.init_params(0)
    for _ in range(40):
        tr.dev_forward_backward(*(t.data_ptr() for t in fr), B)
        tr.dev_adam(1e-4)
    tr.sync()

import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imitation_from_observation_amd import Translator
tr=Translator(max_batch=250); tr.init_params(0)
rng=np.random.default_rng(0)
for B in (25,250):
    x=rng.integers(0,256,(B,64,64,3),dtype=np.uint8)
    for rf in (True, False):
        for _ in range(5): tr.encode(x, return_frames=rf)
        t0=time.perf_counter()
        for _ in range(50): tr.encode(x, return_frames=rf)
        print(B, 'frames' if rf else 'nofrm', (time.perf_counter()-t0)/50*1e3, 'ms')

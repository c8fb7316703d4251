import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd.inception_frontend import InceptionFrontend
S=125; N=192
x=(torch.rand((N,S,S,3),device='cuda')*2-1)
torch.cuda.synchronize()
for prec in ('f32','bf16x3'):
    with InceptionFrontend(S,S,max_images=N,precision=prec) as f:
        f.init_synthetic(0)
        for _ in range(3): f.features_dev(x.data_ptr(), N)
        f.sync()
        t0=time.perf_counter()
        for _ in range(20): f.features_dev(x.data_ptr(), N)
        t1=time.perf_counter(); f.sync(); t2=time.perf_counter()
        print(prec, 'enqueue ms/pass', (t1-t0)/20*1e3, 'total ms/pass', (t2-t0)/20*1e3)

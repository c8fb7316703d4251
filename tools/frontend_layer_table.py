"""Per-op table of the Inception-v3 front end (HIP events per op).  python tools/frontend_layer_table.py [frame] [images] [precision]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the library: one HIP runtime)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd.inception_frontend import InceptionFrontend  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 125
N = int(sys.argv[2]) if len(sys.argv) > 2 else 192
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
with InceptionFrontend(S, S, max_images=N, precision=prec) as f:
    f.init_synthetic(0)
    f.features(np.random.default_rng(0).integers(0, 256, (N, S, S, 3), dtype=np.uint8))
    tab = f.profile(N, iters=5)
    tot = sum(t[3] for t in tab)
    print(f"{len(tab)} ops, {tot:.3f} ms for {N} images of {S}x{S} ({prec}); {sum(t[4] for t in tab) / tot / 1e9:.1f} TF/s")
    for name, shape, grid, ms, fl in sorted(tab, key=lambda t: -t[3])[:40]:
        print(f"{name[11:]:42s} {shape:28s} {grid[0]:3d}x{grid[1]:<3d} {ms:7.3f} ms {fl / ms / 1e9 if ms else 0:7.1f} TF/s")

import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from imitation_from_observation_amd import Translator
rng = np.random.default_rng(0)
fr = rng.integers(0, 256, (25, 64, 64, 3), dtype=np.uint8)
with Translator(max_batch=25) as tr:
    tr.init_params(1)
    tr.set_option("trace_launch", 1)
    tr.set_option("graphs", 0)
    sys.stderr.write("=== translate\n")
    tr.translate(fr, fr[0])
    sys.stderr.write("=== encode\n")
    tr.encode(fr)

"""BASELINE configs[3] on one GPU: 64 triples of 125x125 frames -> frozen Inception-v3 (192 images) -> ContextAEInception2
fwd + bwd + Adam on the 2x2x2048 feature maps, everything resident in HBM and on one stream.  Development tool.
  python tools/bench_config4.py [frame_size] [batch] [layers]     ("layers": also the translator's per-launch table, f32)"""
import os
import sys
import time

import numpy as np
import torch  # before the library: both must share one HIP runtime

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imitation_from_observation_amd import Translator  # noqa: E402
from imitation_from_observation_amd.inception_frontend import InceptionFrontend  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 125
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
LAYERS = len(sys.argv) > 3 and sys.argv[3] == "layers"
AB = len(sys.argv) > 3 and sys.argv[3] == "ab"                  # the translator's stream lanes on / off, alternating, 50 steps each
gen = torch.Generator(device="cuda").manual_seed(0)
frames = torch.randint(0, 256, (3 * B, S, S, 3), device="cuda", generator=gen, dtype=torch.uint8).float() / 127.5 - 1.0
stream = torch.cuda.Stream()
torch.cuda.synchronize()
for prec in (("f32",) if sys.argv[3:] == ["f32only"] else ("f32", "bf16x3")):
    front = InceptionFrontend(S, S, max_images=3 * B, precision=prec, stream=stream.cuda_stream)
    front.init_synthetic(0)
    h, w, c = front.out_shape
    tr = Translator(h, w, 64, 1024, max_batch=B, variant="inception2", C=c, precision=prec, stream=stream.cuda_stream)
    tr.init_params(1)
    lay_flops = front.flops_per_image()
    per = h * w * c * 4

    def step():
        d = front.features_dev(frames.data_ptr(), 3 * B)           # slots: [src | ctx | tgt] frames
        tr.dev_forward_backward(d, d + B * per, d + 2 * B * per, B)
        tr.dev_adam(1e-4)

    for _ in range(3):
        step()
    tr.sync()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        front.features_dev(frames.data_ptr(), 3 * B)
    front.sync()
    tf_ = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    tr.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"[{prec}] frames {S}x{S}, {B} triples/step: Inception-v3 on {3 * B} images {tf_ * 1e3:.2f} ms "
          f"({3 * B / tf_:.0f} images/s, {lay_flops * 3 * B / tf_ / 1e12:.1f} TF/s, {lay_flops / 1e9:.2f} GFLOP/image); "
          f"whole step {dt * 1e3:.2f} ms = {B / dt:.0f} triples/s; loss {tr.dev_scalars()['loss']:.4g}")
    if AB:
        for v in (1, 0, 1, 0, 1, 0):
            tr.set_option("overlap", v)
            for _ in range(3):
                step()
            tr.sync()
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            tr.sync()
            print(f"   translator overlap={v}: whole step {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
    if LAYERS and prec == "f32":
        d = front.features_dev(frames.data_ptr(), 3 * B)
        front.sync()
        ents = tr.profile_step(d, d + B * per, d + 2 * B * per, B, iters=5)
        tot = sum(e["ms"] for e in ents)
        print(f"translator launches one by one: {tot:.3f} ms in {len(ents)} groups")
        for e in sorted(ents, key=lambda e: -e["ms"])[:40]:
            tf = e["flops"] / e["ms"] / 1e9 if e["ms"] > 0 else 0.0
            print(f"  {e['name']:38s} {e['kernel']:36s} {e['ms']:7.3f} ms {tf:7.1f} TF/s all-taps  useful {e['useful_flops'] / max(e['flops'], 1):5.2f}")
    tr.close()
    front.close()

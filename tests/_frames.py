"""Synthetic frame distributions of SURVEY.md 8(d).  The uniform-noise frames (`rng.integers(0, 256, ...)`) live at the call sites;
this is the SECOND, smoother one: low-frequency blobs -- per channel the sum of 4 random 2-D Gaussians, quantised to uint8 -- which
mimic rendered MuJoCo frames (large flat regions, soft gradients) better than noise does.  Smooth inputs change what the parity
tests exercise: neighbouring pixels agree, so conv outputs are large and coherent, filter-gradient sums cancel far less than on
noise, and flat regions put whole rows of activations on the same lrelu branch."""
import numpy as np


def blob_frames(rng, B, H, W, nblobs=4):
    """uint8 [B, H, W, 3]: each channel = sum of `nblobs` Gaussians (random centre, width 0.1 .. 0.5 of the frame, random amplitude),
    scaled to the channel's own [0, 255] range and rounded."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((B, H, W, 3), np.uint8)
    for b in range(B):
        for c in range(3):
            img = np.zeros((H, W))
            for _ in range(nblobs):
                cy, cx = rng.uniform(0, H), rng.uniform(0, W)
                sy, sx = rng.uniform(0.1, 0.5) * H, rng.uniform(0.1, 0.5) * W
                img += rng.uniform(0.3, 1.0) * np.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
            lo, hi = img.min(), img.max()
            out[b, :, :, c] = np.rint(255.0 * (img - lo) / (hi - lo + 1e-12)).astype(np.uint8)
    return out

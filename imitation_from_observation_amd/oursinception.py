"""mode == 'oursinception' of the reward hook (rllab/sampler/base.py:121-132): frames -> frozen Inception-v3 ->
Mixed_7c feature maps -> ContextAEInception2.  `InceptionTranslator` chains the two device handles behind the same
translate / encode surface as `Translator`, so `TranslatorReward` runs unchanged; in this mode the 'image' the cost
compares is the feature tensor (base.py:132: self.image_trans = featreshape)."""
from __future__ import annotations

import numpy as np

from .inception_frontend import InceptionFrontend
from .translator import Translator


class InceptionTranslator:
    def __init__(self, imsize=(125, 125), max_batch=25, device=0, precision=None, df_dim=64, featsize=1024, stream=None):
        self.front = InceptionFrontend(imsize[0], imsize[1], max_images=min(2 * max_batch + 1, 256), device=device,
                                       precision=precision or "f32", stream=stream)
        h, w, c = self.front.out_shape
        self.tr = Translator(h, w, df_dim, featsize, max_batch=max_batch, device=device, variant="inception2", C=c,
                             precision=precision, stream=stream)
        self.H, self.W, self.featsize, self.max_batch = imsize[0], imsize[1], featsize, max_batch
        self.pred_shape = (h, w, c)

    def close(self):
        self.front.close()
        self.tr.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def translate(self, obs_src, obs_tgt0):
        """uint8 frames [B,H,W,3] and the context frame [H,W,3] (or [B,H,W,3]) -> (translated feature maps, translated_z):
        sess.run([model.translated_z, model.out], {image: [src, [ctx]*B, [ctx]*B]}), base.py:216-218."""
        src = np.asarray(obs_src)
        ctx = np.asarray(obs_tgt0)
        batched = ctx.ndim == 4
        f = self.front.features(np.concatenate([src, ctx if batched else ctx[None]]))
        B = src.shape[0]
        return self.tr.translate_f32(f[:B], f[B:] if batched else f[B])

    def encode(self, frames, return_frames=True):
        """(input_z, image_trans[0]) of base.py:234-235 -- image_trans is the feature tensor in this mode."""
        f = self.front.features(frames)
        return self.tr.encode_f32(f), f

    def train_step_u8(self, src, ctx, tgt, lr=1e-4):
        """One Adam step of the translator on uint8 frame triples (scripts/train_script.py:98-114, 163); the front end is frozen."""
        B = len(src)
        f = self.front.features(np.concatenate([src, ctx, tgt]))
        return self.tr.train_step(f[:B], f[B:2 * B], f[2 * B:], lr=lr)

"""mode == 'oursinception' of the reward hook (rllab/sampler/base.py:121-132): frames -> frozen Inception-v3 ->
Mixed_7c feature maps -> ContextAEInception2.  `InceptionTranslator` chains the two device handles behind the same
translate / encode surface as `Translator`, so `TranslatorReward` runs unchanged; in this mode the 'image' the cost
compares is the feature tensor (base.py:132: self.image_trans = featreshape).

Both handles share ONE stream and the feature maps stay in HBM: frames go up as uint8, the front end writes Mixed_7c into its
output buffer, the translator reads that buffer through device pointers (ctx_cnn_forward_u8_dev -> ctx_translate_dev /
ctx_encode_dev / ctx_dev_forward_backward), and only codes, predicted feature maps and scalars come back."""
from __future__ import annotations

import numpy as np

from .inception_frontend import InceptionFrontend
from .translator import Translator


class InceptionTranslator:
    def __init__(self, imsize=(125, 125), max_batch=25, device=0, precision=None, df_dim=64, featsize=1024, stream=None,
                 strides=None, kernels=None, filters=None, train=True):
        # the front end sees src + ctx (+ tgt when training) frames of one batch in a single pass: 3 * max_batch images for the
        # trainer, 2 * max_batch for the reward hook (train=False: translate = B + B, encode = B), which sizes every activation buffer
        self.train = bool(train)
        self.front = InceptionFrontend(imsize[0], imsize[1], max_images=(3 if train else 2) * max_batch, device=device,
                                       precision=precision or "f32", stream=stream)
        h, w, c = self.front.out_shape
        self.tr = Translator(h, w, df_dim, featsize, max_batch=max_batch, device=device, variant="inception2", C=c,
                             precision=precision, stream=self.front.stream, strides=strides, kernels=kernels, filters=filters)
        self.H, self.W, self.featsize, self.max_batch = imsize[0], imsize[1], featsize, max_batch
        self.pred_shape = (h, w, c)
        self._per = h * w * c * 4                          # bytes of one image's feature maps

    def close(self):
        self.tr.close()
        self.front.close()

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
        B = src.shape[0]
        d = self.front.features_u8_dev(np.concatenate([src, ctx if batched else ctx[None]]))
        return self.tr.translate_dev(d, d + B * self._per, B, ctx_batched=batched)

    def encode(self, frames, return_frames=True):
        """(input_z, image_trans[0]) of base.py:234-235 -- image_trans is the feature tensor in this mode (the reward's image term
        compares feature maps, so they do come back to the host here: 32 KB per 125x125 frame)."""
        fr = np.asarray(frames)
        d = self.front.features_u8_dev(fr)
        feat = self.tr.encode_dev(d, fr.shape[0])
        return feat, (self.front.output(fr.shape[0]) if return_frames else None)

    def _triple_dev(self, src, ctx, tgt):
        B = len(src)
        if 3 * B > self.front.max_images:
            raise ValueError(f"{B} triples need {3 * B} front-end images, the handle holds {self.front.max_images}"
                             + ("" if self.train else " (built with train=False: the reward hook's two fetches only)"))
        d = self.front.features_u8_dev(np.concatenate([src, ctx, tgt]))
        return d, d + B * self._per, d + 2 * B * self._per, B

    def train_step_u8(self, src, ctx, tgt, lr=1e-4):
        """One Adam step of the translator on uint8 frame triples (scripts/train_script.py:98-114, 163); the front end is frozen."""
        ds, dc, dt, B = self._triple_dev(src, ctx, tgt)
        self.tr.dev_forward_backward(ds, dc, dt, B)
        self.tr.dev_adam(lr)
        return self.tr.dev_scalars()

    def evaluate_u8(self, src, ctx, tgt, outputs=True):
        """The trainer's validation fetch (train_script.py:176) on uint8 frames: loss, simloss, recon1, recon2 (+ out, out2 and the
        tgt feature maps the nn_err fetch compares with, :148)."""
        ds, dc, dt, B = self._triple_dev(src, ctx, tgt)
        self.tr.dev_forward(ds, dc, dt, B)
        res = self.tr.dev_scalars()
        if outputs:
            res["out"], res["out2"], res["tgt"] = self.tr.last_outputs(out=True, out2=True, tgt=True)
        return res

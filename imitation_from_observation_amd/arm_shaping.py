"""Drop-in mirror of the reference's model interface for this path.

The reference builds `arm_shaping.ContextSkipNew()` , calls `.build(image_placeholder)` and then fetches
the attributes `.input_z .translated_z .out .out2 .recon1 .recon2 .simloss .loss` through
`sess.run(fetches, {image: [src, ctx, tgt]})` (gym/envs/mujoco/arm_shaping.py:1260-1354;
rllab/sampler/base.py:134-145, 216-218, 234-235; scripts/train_script.py:118-129, 163-193).  This class
keeps those names and argument meanings with the TensorFlow session replaced by the HIP translator:

    model = ContextSkipNew()                       # same constructor signature
    model.build((3, 25, 64, 64, 3))                # the placeholder's shape
    model.restore(path)                            # saver.restore(sess, modelname)
    tfeat, timg = model.run([model.translated_z, model.out], [input_img, [context] * 25, [context] * 25])
    feats, image_trans = model.run([model.input_z, model.image_trans], [curimgs, [curimgs[0]] * 25, curimgs])
    _, loss, sim, r1, r2 = model.run([model.optimizer, model.loss, model.simloss, model.recon1, model.recon2],
                                     batch, learning_rate=1e-4)
"""
from __future__ import annotations

import numpy as np

from .translator import Translator

_FETCHES = ("input_z", "translated_z", "out", "out2", "recon1", "recon2", "simloss", "loss", "image_trans", "optimizer")


def preprocess_u8(frames):
    """tf.image.convert_image_dtype(uint8->f32), -0.5, *2.0 (rllab/sampler/base.py:116-119); the same three
    rounded float32 operations the device kernel performs (bit-identical, tests/test_gpu_parity.py)."""
    x = np.asarray(frames).astype(np.float32) * np.float32(1.0 / 255.0)
    return (x - np.float32(0.5)) * np.float32(2.0)


class ContextSkipNew:
    def __init__(self, gf_dim=64, df_dim=64, gfc_dim=1024, dfc_dim=1024, c_dim=3):
        if gf_dim != df_dim:
            raise ValueError("the skip concat pairs encoder/decoder widths: gf_dim must equal df_dim")
        if c_dim != 3:
            raise ValueError("c_dim must be 3")
        self.gf_dim, self.df_dim, self.gfc_dim, self.dfc_dim, self.c_dim = gf_dim, df_dim, gfc_dim, dfc_dim, c_dim
        self.featsize = 1024                               # hard-coded in build(), arm_shaping.py:1277
        self.translator = None
        for f in _FETCHES:                                 # fetch handles: model.out etc.
            setattr(self, f, f)

    variant = "skipnew"

    def build(self, image, device=0, seed=None, ablation_type="None", keep_prob=None):
        """`image`: the placeholder's shape (3, batch, H, W, 3) or an array of that shape.  ablation_type: the loss switch the
        model classes of ablations_code/ablations.py take in build() (:175-182).  keep_prob: ContextAEReal's dropout keep
        probability in training (ablations.py:544 feeds 0.5; the sampler's graph has 1.0)."""
        shape = tuple(getattr(image, "shape", image))
        if len(shape) != 5 or shape[0] != 3 or shape[-1] != self.c_dim:
            raise ValueError(f"expected (3, batch, H, W, {self.c_dim}), got {shape}")
        self.batch_size, self.output_height, self.output_width = shape[1], shape[2], shape[3]
        self.translator = Translator(self.output_height, self.output_width, self.df_dim, self.featsize,
                                     max_batch=self.batch_size, device=device, variant=self.variant, ablation_type=ablation_type,
                                     keep_prob=keep_prob)
        if seed is not None:
            self.translator.init_params(seed)              # tf.global_variables_initializer
        return self

    # saver.restore / saver.save
    def restore(self, path):
        self.translator.load(path)

    def save(self, path, prefix="contextmodel/"):
        self.translator.save(path, prefix=prefix)

    def run(self, fetches, image, learning_rate=None):
        """sess.run(fetches, {image: [src, ctx, tgt]}).  uint8 frames take the sampler's preprocessing
        (base.py:116-119); float frames are used as they are (train_script.py feeds [-1,1] floats)."""
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        for n in names:
            if n not in _FETCHES:
                raise KeyError(f"unknown fetch {n!r}")
        src, ctx, tgt = (np.asarray(x) for x in image)
        u8 = src.dtype == np.uint8
        res = {}
        want = set(names)
        if "optimizer" in want:
            if learning_rate is None:
                raise ValueError("fetching the optimizer needs learning_rate")
            step = self.translator.train_step_u8 if u8 else self.translator.train_step
            res.update(step(src, ctx, tgt, lr=learning_rate))
            res["optimizer"] = None
            want -= {"optimizer", "loss", "simloss", "recon1", "recon2"}
            if want - {"image_trans"}:
                raise ValueError("tensor fetches together with the optimizer are not supported; run them separately")
        if u8 and want and want <= {"input_z", "image_trans"}:
            res["input_z"], x = self.translator.encode(src, return_frames="image_trans" in want)
            if "image_trans" in want:                      # [3, B, H, W, 3]: all three preprocessed slots
                res["image_trans"] = np.stack([x, preprocess_u8(ctx), preprocess_u8(tgt)])
        elif u8 and want and want <= {"translated_z", "out"}:
            res["out"], res["translated_z"] = self.translator.translate(src, ctx)
        elif want:
            f = [preprocess_u8(x) if u8 else np.asarray(x, np.float32) for x in (src, ctx, tgt)]
            res.update(self.translator.evaluate(*f))
            if "image_trans" in want:
                res["image_trans"] = np.stack(f)
            if "translated_z" in want or "input_z" in want:
                res["input_z"], res["translated_z"] = self.translator.last_codes()   # stride-aware (ContextAEReal pads its rows)
        out = [res[n] for n in names]
        return out[0] if single else out


class ContextAEReal(ContextSkipNew):
    """gym/envs/mujoco/arm_shaping.py:1599-1684 -- the model the sampler builds for name in ('real', 'sweep')
    (rllab/sampler/base.py:134-135): one shared encoder, filters 32/16/16/8 with strides 1/2/1/2, featsize 100
    (hard-coded in build(), :1616), keep_prob = 1 (:1476).  Same constructor and fetch names."""
    variant = "real"

    def __init__(self, gf_dim=64, df_dim=64, gfc_dim=1024, dfc_dim=1024, c_dim=3):
        super().__init__(gf_dim, df_dim, gfc_dim, dfc_dim, c_dim)
        self.featsize = 100


class ContextAEInception2(ContextSkipNew):
    """gym/envs/mujoco/arm_shaping.py:1786-1894 -- the translator of mode 'oursinception', built by the sampler as
    `ContextAEInception2(strides=[1,2,1,2], kernels=[3,3,3,3], filters=[1024,1024,512,512])` (rllab/sampler/base.py:126)
    on the Mixed_7c feature maps of the frozen Inception-v3 (`inception_frontend.InceptionFrontend` produces them from
    frames; `oursinception.InceptionTranslator` chains the two).  Same constructor arguments and fetch names; `image` is
    [src, ctx, tgt] FEATURE MAPS, float [3, B, h, w, 2048], and `out = decode + tgtctx` (:1890-1891)."""
    variant = "inception2"

    def __init__(self, strides, kernels, filters):
        """The reference's three lists (arm_shaping.py:1787-1803): s1..s4, k1..k4 (k x k), f1..f4; the decoder mirrors them.
        libctxtrans builds strides 1 | 2, kernel sizes 1..5 and filter counts that are multiples of 32."""
        strides, kernels, filters = [int(v) for v in strides], [int(v) for v in kernels], [int(v) for v in filters]
        if not (len(strides) == len(kernels) == len(filters) == 4):
            raise ValueError("strides, kernels and filters have four entries each (arm_shaping.py:1801-1803)")
        if any(s not in (1, 2) for s in strides) or any(not 1 <= k <= 5 for k in kernels) or any(f <= 0 or f % 32 for f in filters):
            raise ValueError(f"libctxtrans builds strides 1|2, kernels 1..5, filters multiples of 32; got {strides}, {kernels}, {filters}")
        self.strides, self.kernels, self.filters = strides, kernels, filters
        self.df_dim = self.gf_dim = max(filters[3] // 8, 4)
        self.featsize = 1024                               # hard-coded in build(), arm_shaping.py:1797
        self.translator = None
        for f in _FETCHES:
            setattr(self, f, f)

    def build(self, image, device=0, seed=None, precision=None):
        shape = tuple(getattr(image, "shape", image))
        if len(shape) != 5 or shape[0] != 3:
            raise ValueError(f"expected (3, batch, h, w, C) feature maps, got {shape}")
        self.batch_size, self.output_height, self.output_width, self.c_dim = shape[1:]
        self.translator = Translator(self.output_height, self.output_width, self.df_dim, self.featsize, max_batch=self.batch_size,
                                     device=device, variant=self.variant, C=self.c_dim, precision=precision,
                                     strides=self.strides, kernels=self.kernels, filters=self.filters)
        if seed is not None:
            self.translator.init_params(seed)
        return self

    def run(self, fetches, image, learning_rate=None):
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        for n in names:
            if n not in _FETCHES:
                raise KeyError(f"unknown fetch {n!r}")
        src, ctx, tgt = (np.asarray(x, np.float32) for x in image)
        res, want = {}, set(names)
        if "optimizer" in want:
            if learning_rate is None:
                raise ValueError("fetching the optimizer needs learning_rate")
            res.update(self.translator.train_step(src, ctx, tgt, lr=learning_rate))
            res["optimizer"] = None
            want -= {"optimizer", "loss", "simloss", "recon1", "recon2"}
            if want - {"image_trans"}:
                raise ValueError("tensor fetches together with the optimizer are not supported; run them separately")
        if "image_trans" in want:                          # base.py:132: image_trans = featreshape, the fed tensor itself
            res["image_trans"] = np.stack([src, ctx, tgt])
            want.discard("image_trans")
        if "input_z" in want:
            res["input_z"] = self.translator.encode_f32(src)
        if want & {"translated_z", "out"} and not want & {"out2", "loss", "simloss", "recon1", "recon2"}:   # neither depends on tgt
            res["out"], res["translated_z"] = self.translator.translate_f32(src, ctx)   # the demo-cache fetch, base.py:216-218
        elif want - {"input_z"}:
            res.update(self.translator.evaluate(src, ctx, tgt))
            if "translated_z" in want:
                res["translated_z"] = self.translator.translate_f32(src, ctx)[1]
        out = [res[n] for n in names]
        return out[0] if single else out

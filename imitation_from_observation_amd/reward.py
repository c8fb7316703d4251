"""The image-feature reward of the reference's TRPO loop, on top of `Translator`.

Restates the `mode == 'ours'` branch of BaseSampler.process_samples (rllab/sampler/base.py:192-257):
  * once per experiment: translate every demo video into the current context (first rollout frame of
    each viewpoint) and cache the mean translated feature track and mean translated frames
    (base.py:195-223);
  * per path: encode the 25 rollout frames, cost_j = ||means_j - feat_j||^2 + scale * ||imgs_j - x_j||^2
    summed over viewpoints (base.py:232-245), and  rewards[2j+1] -= cost_j * j^2  (base.py:256-257).
The arithmetic on the frames runs in the HIP translator; several paths are encoded per launch (the
encoder is per-frame independent, so results do not depend on the grouping).  The reference's internal
inconsistencies on this path (SURVEY.md 3.4 e-g) are resolved to the intended behaviour and noted inline.
"""
from __future__ import annotations

import numpy as np


class TranslatorReward:
    def __init__(self, translator, nvp, scale, name="strike", ablation_type="None", batch_size=25):
        if ablation_type not in ("None", "nofeat", "noimage"):
            # 'recon' reads an undefined `image_recon` in the reference (base.py:250-252)
            raise NotImplementedError(f"ablation_type {ablation_type!r} is not runnable in the reference either")
        self.tr, self.nvp, self.scale, self.name = translator, int(nvp), float(scale), name
        self.ablation_type, self.batch_size = ablation_type, int(batch_size)
        self.skip = 2 if name in ("real", "sweep") else 1        # base.py:209-211
        self.means, self.imgs = None, None
        self.validdata = None                                    # set_demos(): the cache is then built lazily on the first path
        # mode 'oursinception' caps the demo videos at 50 (base.py:203-204); every other mode uses them all
        self.nvideos_cap = 50 if hasattr(translator, "front") else None

    @classmethod
    def for_sampler(cls, name, imsize, nvp, scale, modelname=None, ablation_type="None", batch_size=25,
                    paths_per_launch=10, device=0, mode="ours", inception_ckpt=None):
        """What BaseSampler.initialize() sets up for mode 'ours' (base.py:113-145): the model class follows the
        experiment name -- ContextAEReal for 'real'/'sweep', ContextSkipNew otherwise (:134-137) -- on the
        sampler's imsize, restored from `modelname` when given (:138).  mode 'oursinception' (:121-132): frames go
        through the frozen Inception-v3 (variables from `inception_ckpt`, an .npz keyed by the TF names) and
        ContextAEInception2 runs on the Mixed_7c feature maps."""
        from .translator import Translator
        if mode == "oursinception":
            from .oursinception import InceptionTranslator
            it = InceptionTranslator(imsize, max_batch=batch_size * paths_per_launch, device=device, train=False)
            if inception_ckpt is not None:
                it.front.load(inception_ckpt)
            if modelname is not None:
                it.tr.load(modelname)
            return cls(it, nvp, scale, name=name, ablation_type=ablation_type, batch_size=batch_size)
        real = name in ("real", "sweep")
        tr = Translator(imsize[0], imsize[1], featsize=100 if real else 1024, max_batch=batch_size * paths_per_launch,
                        device=device, variant="real" if real else "skipnew")
        if modelname is not None:
            tr.load(modelname)
        return cls(tr, nvp, scale, name=name, ablation_type=ablation_type, batch_size=batch_size)

    # ------------------------------------------------------------------ base.py:195-223
    @staticmethod
    def _frames_of(path):
        """env_infos['imgs'] holds, every other step, a list over viewpoints of uint8 frames (base.py:193)."""
        return [img for img in path["env_infos"]["imgs"] if img is not None]

    def set_demos(self, validdata):
        """np.load(self.algo._kwargs['modeldata']) (base.py:198), kept for the lazy cache build of process_paths."""
        self.validdata = np.asarray(validdata)
        return self

    def build_demo_cache(self, validdata, first_frames, distributed=False):
        """validdata: demo tensor [T, Nvid, H, W, 3] in [-1,1] (np.load(modeldata), base.py:198);
        first_frames[vp]: uint8 context frame = first frame of the current rollout (base.py:200).
        In mode 'oursinception' only the first 50 videos are used (`nvideos = 50`, base.py:203-204) and the reference feeds
        `validdata[::skip, i]` to its uint8 placeholder WITHOUT the (x+1)*127.5 conversion (:212-213) -- so a uint8 demo
        tensor is taken as it is there; a float one is converted like in the other modes (deviations).
        distributed=True (one rank per GPU): the demo videos are sharded rank::world, every rank translates its shard and
        the partial feature / frame sums are combined with ONE all-reduce per viewpoint -- the demo means are a plain sum
        over videos.  The group is the translator's own RCCL group when it has one (Translator.dp_init:
        ctx_dp_allreduce_host_f64, no torch in the sampler process), else an initialised torch.distributed group."""
        validdata = np.asarray(validdata)
        nvid = validdata.shape[1]
        if self.nvideos_cap is not None:
            nvid = min(nvid, self.nvideos_cap)
        raw_u8 = validdata.dtype == np.uint8
        bs = self.batch_size
        self.means, self.imgs = [], []
        per_call = max(1, self.tr.max_batch // bs)
        rank, world = 0, 1
        cabi = False
        if distributed:
            own = getattr(self.tr, "dp_world", None)
            if callable(own) and own()[1] > 1:
                rank, world = own()
                cabi = True
            else:
                import torch
                import torch.distributed as dist
                rank, world = dist.get_rank(), dist.get_world_size()
        mine = list(range(rank, nvid, world))
        for vp in range(self.nvp):
            ctx = np.ascontiguousarray(first_frames[vp], dtype=np.uint8)
            fsum = np.zeros((bs, self.tr.featsize), np.float64)
            pshape = tuple(getattr(self.tr, "pred_shape", (self.tr.H, self.tr.W, 3)))   # feature maps in mode 'oursinception'
            isum = np.zeros((bs,) + pshape, np.float64)
            for i0 in range(0, len(mine), per_call):
                vids = mine[i0:i0 + per_call]
                # ((validdata[::skip, i] + 1) * 127.5).astype(np.uint8), base.py:215
                u8 = np.concatenate([validdata[::self.skip, i][:bs] if raw_u8 else
                                     ((validdata[::self.skip, i][:bs] + 1) * 127.5).astype(np.uint8) for i in vids])
                timg, tfeat = self.tr.translate(u8, ctx)                   # [translated_z, out], base.py:216-218
                fsum += tfeat.reshape(len(vids), bs, -1).sum(0)
                isum += timg.reshape((len(vids), bs) + pshape).sum(0)
            if cabi:
                flat = self.tr.dp_allreduce_host(np.concatenate([fsum.ravel(), isum.ravel()]))
                fsum, isum = flat[:fsum.size].reshape(fsum.shape), flat[fsum.size:].reshape(isum.shape)
            elif distributed and world > 1:
                flat = torch.from_numpy(np.concatenate([fsum.ravel(), isum.ravel()]))
                if dist.get_backend() == "nccl":
                    flat = flat.cuda()
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat = flat.cpu().numpy()
                fsum, isum = flat[:fsum.size].reshape(fsum.shape), flat[fsum.size:].reshape(isum.shape)
            self.means.append((fsum / nvid).astype(np.float32))            # np.mean(tfeats, axis=0), base.py:221
            self.imgs.append((isum / nvid).astype(np.float32))             # np.mean(timgs, axis=0), base.py:222
            if hasattr(self.tr, "reward_set_cache"):                       # the cost is then computed on the device, next to the encoder
                self.tr.reward_set_cache(vp, self.means[vp], self.imgs[vp])
        return self

    # ------------------------------------------------------------------ base.py:232-252
    def _costs_from(self, feats, frames_f32, vp):
        cf = np.sum((self.means[vp] - feats) ** 2, axis=1)
        ci = self.scale * np.sum((self.imgs[vp] - frames_f32) ** 2, axis=(1, 2, 3))
        if self.ablation_type == "nofeat":      # reference indexes self.imgs without [vp]
            return ci
        if self.ablation_type == "noimage":
            return cf
        return cf + ci

    def _group(self, distributed):
        """(rank, world, allsum) of the group a distributed call runs on: the translator's own RCCL group when it has one
        (Translator.dp_init -- ctx_dp_allreduce_host_f64, no torch in the sampler process), else an initialised torch.distributed group."""
        if not distributed:
            return 0, 1, (lambda x: x)
        own = getattr(self.tr, "dp_world", None)
        if callable(own) and own()[1] > 1:
            rank, world = own()
            return rank, world, self.tr.dp_allreduce_host
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()

        def allsum(x):
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.cpu().numpy()
        return rank, world, (allsum if world > 1 else (lambda x: x))

    def paths_costs(self, paths, distributed=False):
        """costs[p][j] for every path, many paths per encoder launch.
        distributed=True (one rank per GPU, every rank holding the same `paths` and the same demo cache): the >= 250 rollout paths of a
        TRPO iteration are sharded rank::world -- a path's cost depends on nothing but its own frames (base.py:232-249) -- and the
        [npaths, bs] cost table is completed with ONE all-reduce (last sentence); every rank returns the full table."""
        rank, world, allsum = self._group(distributed)
        if world > 1:
            if self.means is None:
                raise RuntimeError("distributed paths_costs needs the demo cache first (build_demo_cache(..., distributed=True))")
            mine = list(range(rank, len(paths), world))
            part = np.zeros((len(paths), self.batch_size), np.float64)
            if mine:
                part[mine] = self.paths_costs([paths[i] for i in mine])
            return allsum(part.ravel()).reshape(part.shape).astype(np.float32)
        bs = self.batch_size
        frames = [self._frames_of(p) for p in paths]
        for f in frames:
            if len(f) != bs:
                raise ValueError(f"a path has {len(f)} rendered frames, the sampler's placeholder holds {bs} (base.py:115)")
        if self.means is None:
            if self.validdata is None:
                raise RuntimeError("no demo cache: call build_demo_cache(validdata, first_frames), or set_demos(validdata) to have it "
                                   "built on the first path like the reference (base.py:195-223)")
            # `context = imgs[0][vp]`: the first rendered frame of the FIRST path, per viewpoint (base.py:200)
            self.build_demo_cache(self.validdata, [frames[0][0][vp] for vp in range(self.nvp)])
        costs = np.zeros((len(paths), bs), np.float32)
        per_call = max(1, self.tr.max_batch // bs)
        for vp in range(self.nvp):
            for p0 in range(0, len(paths), per_call):
                grp = range(p0, min(len(paths), p0 + per_call))
                u8 = np.concatenate([np.stack([fr[vp] for fr in frames[p]]).astype(np.uint8) for p in grp])
                if hasattr(self.tr, "reward_costs"):
                    # encoder + cost on the device: only the [paths, bs] costs cross PCIe (not the 4-bytes-per-pixel frames)
                    dev = self.tr.reward_costs(vp, u8, self.scale, self.ablation_type)
                else:
                    feats, x = self.tr.encode(u8)                          # [input_z, image_trans[0]], base.py:234-235
                for k, p in enumerate(grp):
                    sl = slice(k * bs, (k + 1) * bs)
                    c = dev[k] if hasattr(self.tr, "reward_costs") else self._costs_from(feats[sl], x[sl], vp)
                    # 'None' accumulates over viewpoints (costs += ...); the ablations overwrite (costs = ...)
                    costs[p] = costs[p] + c if self.ablation_type == "None" else c
        return costs

    # ------------------------------------------------------------------ base.py:256-257
    def process_paths(self, paths, distributed=False):
        """In place: path['rewards'][2j+1] -= costs[j] * j**2.  After set_demos(validdata) the demo cache is built lazily from
        the first path's first frame per viewpoint, as the reference does (base.py:195-200); otherwise build_demo_cache() must
        have been called.  distributed=True: the paths' costs are computed rank::world and gathered (paths_costs); every rank then
        applies them to its copy of `paths`."""
        if distributed and self.means is None and self.validdata is not None:
            frames0 = self._frames_of(paths[0])
            self.build_demo_cache(self.validdata, [frames0[0][vp] for vp in range(self.nvp)], distributed=True)
        costs = self.paths_costs(paths, distributed=distributed)
        for p, c in zip(paths, costs):
            for j in range(self.batch_size):
                p["rewards"][j * 2 + 1] -= c[j] * (j ** 2)
        return costs

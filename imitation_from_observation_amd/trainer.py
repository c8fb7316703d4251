"""Translator training loop: `ModelTrainer` of scripts/train_script.py:28-204 over the HIP translator.

Same constructor arguments, same loop (`train_script.py:144-203`):
  * demo tensor vdata[T, N, H, W, 3] split into the first `ntrain` videos (training) and the rest (validation); the first
    200 training videos are written to `<basedir>vdata_train.npy` (:144-152);
  * for itr in 1 .. nitr-1: choicesrc, choicetgt = np.random.choice(ntrain, batch_size) twice;
    src[b] = traindata[b % nlen, choicesrc[b]], tgt[b] = traindata[b % nlen, choicetgt[b]], ctx[b] = traindata[0, choicetgt[b]];
    one Adam(1e-4) step; every 4th iteration logs "itr loss sim r1 r2 err" (:153-167);
  * every 40 iterations (and at every `save_every`) a validation batch sampled the same way from the held-out videos is
    evaluated and logged with a trailing "E" (:169-178);
  * every `save_every` iterations a checkpoint `<basedir><itr>/model_<itr>_<loss>_<r1>_<r2>_<err>` and `validloss.npy` are
    written, plus -- not for the Inception variant -- ten (translation, reconstruction) clips of single video pairs (:179-195);
  * from `save_every` on, every validation appends a row Iteration, Loss, Sim, R1, R2, NNErr to the tabular log (:196-203).

What differs, and why:
  * mp4 DECODING (:69, imageio + ffmpeg) is not available here: the demo tensor comes in as an array or a `.npy` path (the file
    the reference itself saves), or as `videos=` -- decoded videos [51, H, W, 3] uint8 -- which then go through the reference's own
    loop (:59-96: frame selection by nskip, `transform` = scipy.misc.imresize bilinear + /127.5 - 1, black-frame drop; restated in
    demo_pipeline.py, the resize bit-exact against Pillow; the list is shuffled with np.random first, :66) and are saved as
    `<basedir>vdata_strike<itr>.npy` like :95 (itr = the videos looked at);
  * the sess.run calls are `Translator.train_step_sampled` / `eval_sampled` on the demo tensor resident in HBM (uint8;
    `gather_triples_kernel` builds the batch with the trainer's x / 127.5 - 1 scaling) when the float demo tensor lies
    exactly on that uint8 lattice -- bit-identical to feeding the host-gathered float batch -- and
    `train_step` / `evaluate` on the host-gathered batch otherwise;
  * `nn_err` (:148) reads `featreshape`, which only exists in the Inception branch (NameError for the other
    models).  Its intended meaning -- for every output j the index of the nearest tgt frame, compared with j % nlen -- is
    computed on the host from `out` and the tgt slot for every model;
  * clips are written as `__<k>trans.gif` / `__<k>recon.gif` like :23-26, :193-194 -- by Pillow (imageio is absent), the frames
    `(clip(inverse_transform(f), 0, 1) * 255).astype(uint8)`; a box without Pillow gets the same frames as `.npy`;
  * the tabular log is a CSV `<basedir>progress.csv` (rllab's logger is outside the hot path).  As in the reference, on a
    save iteration its R1 / R2 columns hold the LAST CLIP's recon1 / recon2 (the clip loop re-uses the names, :192-193),
    while Loss / Sim / NNErr stay the validation batch's.
Random draws come from the global `np.random` in the reference's order, so a seeded run samples the same batches.

N GPUs (`rank`, `world`; new -- the reference is single-device): one ModelTrainer per process.  Every rank
keeps the whole uint8 demo tensor in HBM and runs the SAME loop: rank 0's `np.random` state is handed to the others once, so all
draw the same index arrays; `ctx_dp_train_step_sampled` gathers rank r's rows [r B/world, (r+1) B/world) of the GLOBAL batch on the
device (t = b % nlen on the global row b) and runs the bucketed RCCL step; the validation batch is sharded the same way
(`ctx_dp_eval_sampled`).  Logged scalars are GLOBAL (recon sums add, simloss is the mean over the global batch), `nn_err` is the sum
of the ranks' shares (each rank compares its outputs with ALL tgt frames of the batch), and only rank 0 writes checkpoints, clips,
`validloss.npy`, `vdata_train.npy` and the CSV.  The parameters after k steps equal the single-GPU run's on the same draws up to f32
summation order (tests/test_gpu_dp_two_ranks.py).
"""
from __future__ import annotations

import csv
import os

import numpy as np

LEARNING_RATE = 1e-4          # fed at every step, train_script.py:163,167


def save_clip(stem, frames_u8):
    """savegif (train_script.py:22-26): one GIF of the uint8 frames [n, H, W, 3], 100 ms each, written by Pillow (which folds a frame equal
    to its predecessor into that one's display time: the animation is the same); `.npy` where Pillow is missing."""
    try:
        from PIL import Image
    except ImportError:
        np.save(stem, frames_u8)
        return stem + ".npy"
    ims = [Image.fromarray(np.ascontiguousarray(f)) for f in frames_u8]
    ims[0].save(stem + ".gif", save_all=True, append_images=ims[1:], loop=0, duration=100)
    return stem + ".gif"


def nn_err(tgt, out, nlen, j0=0):
    """train_script.py:148 with featreshape[2] = the tgt slot:
    sum_j | argmin_i mean((tgt_i - out_j)^2) - (j % nlen) |.
    j0: `out` holds rows j0, j0 + 1, ... of the batch (a data-parallel shard's share of the sum; `tgt` is always the whole batch)."""
    a = np.asarray(tgt, np.float64).reshape(len(tgt), -1)
    b = np.asarray(out, np.float64).reshape(len(out), -1)
    # mean((a_i - b_j)^2) = (|a_i|^2 + |b_j|^2 - 2 a_i.b_j) / n
    d = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * (a @ b.T)
    return int(np.abs(np.argmin(d, axis=0) - (j0 + np.arange(len(out))) % nlen).sum())


def on_u8_lattice(vdata):
    """(uint8 tensor, True) when the float demo tensor is exactly what transform() makes of uint8 frames
    (k / 127.5 - 1, train_script.py:16-19), else (None, False)."""
    v = np.asarray(vdata)
    if v.dtype == np.uint8:
        return v, True
    k = np.rint((v.astype(np.float64) + 1.0) * 127.5)
    if k.min() < 0 or k.max() > 255:
        return None, False
    # exact: the f32 value the session is fed must BE the device table's entry  float32(k / 127.5 - 1)  (ctx_demos_upload)
    if not np.array_equal((k / 127.5 - 1.0).astype(np.float32), v.astype(np.float32)):
        return None, False
    return k.astype(np.uint8), True


class ModelTrainer:
    MODELS = {"ContextSkipNew": "skipnew", "ContextAEReal": "real", "ContextAEInception": "inception2"}

    def __init__(self, idims, nvideos, ntrain, batch_size, model, nitr, save_every, nlen, nskip, rescale=True, inception=False,
                 strides=None, kernels=None, filters=None, *, vdata=None, basedir="model/", device=0, seed=0, translator=None,
                 precision=None, log=print, rank=0, world=1, dp_unique_id=None, videos=None):
        """The reference's 14 positional arguments (train_script.py:29-30; the launchers omit the last five, SURVEY.md 3.4-b,
        hence the defaults), then: vdata (array or .npy path of the demo tensor), basedir (logger._snapshot_dir), device,
        seed of the parameter initialiser, an optional ready-made translator (tests), the arithmetic, the log sink.
        rank / world: one trainer per GPU process (module docstring); `batch_size` stays the GLOBAL batch (a multiple of world).
        dp_unique_id: the 128-byte blob of Translator.dp_unique_id() made on rank 0 and shipped to every rank by the launcher (omitted:
        broadcast through an initialised torch.distributed group; not needed when `translator` already went through dp_init)."""
        if model not in self.MODELS:
            raise ValueError(f"model must be one of {sorted(self.MODELS)}")
        self.idims, self.nvideos, self.ntrain, self.batch_size = tuple(idims), nvideos, ntrain, batch_size
        self.model, self.nitr, self.save_every, self.nlen, self.nskip = model, nitr, save_every, nlen, nskip
        self.rescale, self.inception = rescale, inception
        self.strides, self.kernels, self.filters = strides, kernels, filters
        self.vdata, self.basedir, self.device, self.seed = vdata, basedir, device, seed
        self.videos = videos       # decoded demo videos (arrays [51, H, W, 3] uint8 or callables returning them): train_script.py:59-96
        self.translator, self.precision, self.log = translator, precision, log
        self.rank, self.world, self.dp_unique_id = int(rank), int(world), dp_unique_id
        if self.world < 1 or not 0 <= self.rank < self.world:
            raise ValueError(f"rank {rank} / world {world}")
        if self.world > 1 and batch_size % self.world:
            raise ValueError(f"batch_size {batch_size} (the GLOBAL batch) must be a multiple of world = {world}")
        if self.world > 1 and inception:
            raise ValueError("the Inception variant trains on one GPU here (its demo frames are not kept resident)")
        self.allloss, self.validloss = [], []

    # ------------------------------------------------------------------ the model behind the four sess.run sites
    def _build(self):
        if self.translator is not None:
            return self.translator
        from .translator import Translator
        H, W = self.idims
        if self.inception:
            from .oursinception import InceptionTranslator
            tr = InceptionTranslator((H, W), max_batch=self.batch_size, device=self.device, precision=self.precision,
                                     strides=self.strides, kernels=self.kernels, filters=self.filters)   # train_script.py:110
            tr.tr.init_params(self.seed)
            return tr
        variant = self.MODELS[self.model]
        tr = Translator(H, W, featsize=100 if variant == "real" else 1024, max_batch=self.batch_size // self.world, device=self.device,
                        variant=variant, precision=self.precision)
        tr.init_params(self.seed)                                  # tf.global_variables_initializer, train_script.py:129
        return tr

    def _join_group(self, tr):
        """world > 1: this rank's handle joins the RCCL group (ctx_dp_init: rank 0's parameters / Adam state reach every replica)."""
        if tr.dp_world()[1] == self.world:
            return
        uid = self.dp_unique_id
        if uid is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise ValueError("world > 1 needs dp_unique_id (Translator.dp_unique_id() of rank 0) or an initialised torch.distributed group")
            box = [type(tr).dp_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        tr.dp_init(uid, self.rank, self.world)

    def _batch(self, data, choicesrc, choicetgt):
        ar = np.arange(0, self.batch_size) % self.nlen
        return data[ar, choicesrc], data[0, choicetgt], data[ar, choicetgt]          # srcdata, tgtctx, tgtdata (:156-159)

    def train(self):
        basedir = self.basedir if self.basedir.endswith("/") else self.basedir + "/"
        os.makedirs(basedir, exist_ok=True)
        rank, world, dp = self.rank, self.world, self.world > 1
        tr = self._build()
        if dp:
            self._join_group(tr)

        def allsum(x):
            """sum over the ranks of a small host array (ctx_dp_allreduce_host_f64); the identity on one GPU"""
            x = np.ascontiguousarray(x, dtype=np.float64)
            return tr.dp_allreduce_host(x) if dp else x
        if dp:
            # every rank draws the same numbers: rank 0's np.random state (MT19937: 624 words + position + the cached gaussian) --
            # BEFORE build_vdata, whose np.random.shuffle(videos) decides the video order, the subset and the train / valid split
            st = np.random.get_state()
            flat = np.zeros(627, np.float64)
            if rank == 0:
                flat[:624], flat[624], flat[625], flat[626] = st[1], st[2], st[3], st[4]
            flat = allsum(flat)
            np.random.set_state((st[0], flat[:624].astype(np.uint32), int(flat[624]), int(flat[625]), float(flat[626])))
        if self.vdata is None and self.videos is not None:
            from .demo_pipeline import build_vdata
            vdata, looked_at = build_vdata(self.videos, self.idims, self.nvideos, self.nlen, self.nskip, self.rescale, self.inception,
                                           log=self.log if self.rank == 0 else None, return_count=True)
            if self.rank == 0:
                np.save(basedir + "vdata_strike" + str(looked_at), vdata)                      # train_script.py:95 (named after `itr`, the videos looked at)
        else:
            vdata = np.load(self.vdata) if isinstance(self.vdata, (str, os.PathLike)) else np.asarray(self.vdata)
        if vdata.ndim != 5 or vdata.shape[2:4] != self.idims or vdata.shape[0] < self.nlen:
            raise ValueError(f"vdata must be [T >= {self.nlen}, N, {self.idims[0]}, {self.idims[1]}, 3], got {vdata.shape}")
        B, nlen = self.batch_size, self.nlen
        log = self.log if rank == 0 else (lambda s: None)          # one log, one set of files: rank 0's
        if not (self.vdata is None and self.videos is not None):
            log(str(vdata.shape))                                  # (:96 -- build_vdata logs it on the `videos=` path)
        Bl, j0 = B // world, rank * (B // world)                   # this rank's rows of the global batch
        n = vdata.shape[1]
        ntrain = self.ntrain
        nvalid = n - ntrain
        if ntrain <= 0 or nvalid <= 0:
            raise ValueError(f"ntrain = {ntrain} of {n} videos leaves no training / validation split")
        log("%s %s" % (ntrain, nvalid))
        validdata = vdata[:, ntrain:]
        traindata = vdata[:, :ntrain]
        log(str(validdata.shape) + str(traindata.shape))
        if rank == 0:
            np.save(basedir + "vdata_train", traindata[:, :200])
        # device-resident demo tensor + device sampler where that is bit-identical to the host gather
        u8, lattice = on_u8_lattice(vdata) if not self.inception else (None, False)
        resident = lattice and hasattr(tr, "load_demos")
        if dp and not resident:
            raise ValueError("data-parallel training runs on the device-resident sampler: the demo tensor must lie on the uint8 lattice "
                             "(k / 127.5 - 1, what train_script.py:16-19 makes of video frames)")
        if resident:
            # only frames t < nlen are ever sampled (t = b % nlen, and frame 0 for the context); the device sampler takes
            # t = b % T with T = the uploaded tensor's length, so upload exactly nlen frames (vdata may hold more)
            tr.load_demos(np.ascontiguousarray(u8[:nlen]))

        def train_step(cs, ct):
            if dp:
                return tr.dp_train_step_sampled(cs, ct, lr=LEARNING_RATE)          # GLOBAL scalars
            if resident:
                return tr.train_step_sampled(cs, ct, lr=LEARNING_RATE)
            src, ctx, tgt = self._batch(traindata, cs, ct)
            if self.inception:
                return tr.train_step_u8(src, ctx, tgt, lr=LEARNING_RATE)
            return tr.train_step(src, ctx, tgt, lr=LEARNING_RATE)

        def evaluate(cs, ct):
            """loss, sim, r1, r2, out, out2, tgt of a validation batch (indices into validdata)."""
            src, ctx, tgt = self._batch(validdata, cs, ct)
            if dp:                                             # GLOBAL scalars; out / out2 = this rank's rows
                ev = tr.dp_eval_sampled(np.asarray(cs) + ntrain, np.asarray(ct) + ntrain)
            elif resident:
                ev = tr.eval_sampled(np.asarray(cs) + ntrain, np.asarray(ct) + ntrain)
            elif self.inception:                               # feature maps stay on the device; nn_err compares with the tgt MAPS
                ev = tr.evaluate_u8(src, ctx, tgt)
                tgt = ev["tgt"]
            else:
                ev = tr.evaluate(src, ctx, tgt)
            return ev, tgt

        core = getattr(tr, "tr", tr)                               # the Translator inside an InceptionTranslator
        rows = []
        for itr in range(1, self.nitr):
            choicesrc = np.random.choice(ntrain, B)
            choicetgt = np.random.choice(ntrain, B)
            sc = train_step(choicesrc, choicetgt)
            if itr % 4 == 0:
                out, _, tgt = core.last_outputs(out=True, tgt=True)
                if dp:                                         # this rank's outputs against ALL tgt frames of the batch (:148)
                    tgt = self._batch(traindata, choicesrc, choicetgt)[2]
                err = int(allsum([nn_err(tgt, out, nlen, j0)])[0])
                log("%s %s %s %s %s %s" % (itr, sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"], err))
                self.allloss.append(sc["loss"])
            if itr % 40 == 0 or itr % self.save_every == 0:
                choicesrc = np.random.choice(nvalid, B)
                choicetgt = np.random.choice(nvalid, B)
                ev, tgt = evaluate(choicesrc, choicetgt)
                loss, sim, r1, r2 = ev["loss"], ev["simloss"], ev["recon1"], ev["recon2"]
                err = int(allsum([nn_err(tgt, ev["out"], nlen, j0)])[0])
                log("%s %s %s %s %s %s E" % (itr, loss, sim, r1, r2, err))
                self.validloss.append(loss)
                if itr % self.save_every == 0:
                    if rank == 0:
                        os.mkdir(basedir + str(itr))
                        core.save("%s%d/model_%d_%.2f_%.2f_%.2f_%d" % (basedir, itr, itr, loss, r1, r2, err), prefix="contextmodel/")
                        np.save("%s%d/validloss" % (basedir, itr), self.validloss)
                    if not self.inception:
                        for kk in range(10):
                            choicesrc = [np.random.randint(nvalid)] * B
                            choicetgt = [np.random.randint(nvalid)] * B
                            clip, _ = evaluate(choicesrc, choicetgt)
                            r1, r2 = clip["recon1"], clip["recon2"]           # the reference's clip fetch overwrites r1 / r2 (:192-193): the
                                                                              # tabular R1 / R2 of a save iteration are the last clip's
                            for tag, frames in (("trans", clip["out"]), ("recon", clip["out2"])):
                                if dp:                         # the clip is rows 0 .. nlen-1 of the GLOBAL batch: every rank adds the rows it holds
                                    head = np.zeros((nlen,) + frames.shape[1:], np.float64)
                                    lo, hi = max(j0, 0), min(j0 + Bl, nlen)
                                    if hi > lo:
                                        head[lo:hi] = frames[lo - j0:hi - j0]
                                    # (the group sums in float64; the quantisation below is the single-GPU path's and the reference's float32 arithmetic)
                                    frames = allsum(head.ravel()).reshape(head.shape).astype(np.float32)
                                u = (np.clip((frames[:nlen] + 1.0) / 2.0, 0, 1) * 255).astype(np.uint8)     # savegif's frames (:23-26)
                                if rank == 0:
                                    save_clip("%s%d/__%d%s" % (basedir, itr, kk, tag), u)
                if itr >= self.save_every and rank == 0:
                    rows.append(dict(Iteration=itr, Loss=loss, Sim=sim, R1=r1, R2=r2, NNErr=err))
                    with open(basedir + "progress.csv", "w", newline="") as f:
                        w = csv.DictWriter(f, fieldnames=["Iteration", "Loss", "Sim", "R1", "R2", "NNErr"])
                        w.writeheader()
                        w.writerows(rows)
        self.translator = tr
        return tr

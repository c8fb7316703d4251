"""`imitation_from_observation_amd.trainer.ModelTrainer` against a literal restatement of the reference's training loop
(scripts/train_script.py:144-203) on the oracle: same batches from the same np.random stream, same log lines, same
validation cadence, checkpoint names and tabular rows.  The GPU test runs the real thing on a tiny net and checks the
device-resident sampler path against the host-gather path."""
import csv
import os

import numpy as np
import pytest

from imitation_from_observation_amd.trainer import ModelTrainer, nn_err, on_u8_lattice
from oracle import ctx_oracle as o

H = W = 16
CFG = o.SkipNewConfig(H=H, W=W, df_dim=4, gf_dim=4, featsize=8)
B, NLEN, NVID, NTRAIN, NITR, SAVE = 6, 3, 8, 5, 45, 20


def read_clip(path):
    """frames [n, H, W, 3] uint8 of a clip written by trainer.save_clip (a GIF: palette colours, so compare clips with clips).  Pillow's
    writer folds a frame that equals its predecessor into the predecessor's display time: a frame shown for k x 100 ms counts k times."""
    from PIL import Image, ImageSequence
    with Image.open(path) as im:
        out = []
        for f in ImageSequence.Iterator(im):
            out += [np.asarray(f.convert("RGB"))] * max(1, int(round(f.info.get("duration", 100) / 100)))
        return np.stack(out)


class OracleModel:
    """The four sess.run sites on the oracle: the surface ModelTrainer drives."""

    def __init__(self, seed):
        self.p = o.init_params(CFG, seed, np.float64, stddev=0.05)
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.t = 0
        self.saved = []

    def train_step(self, src, ctx, tgt, lr):
        self.t += 1
        res, _ = o.train_step(self.p, self.m, self.v, self.t, *(np.asarray(x, np.float64) for x in (src, ctx, tgt)), lr, CFG)
        self._last = (res["out"], np.asarray(tgt, np.float64))
        return {k: float(res[k]) for k in ("loss", "simloss", "recon1", "recon2")}

    def evaluate(self, src, ctx, tgt):
        res, _ = o.forward(self.p, *(np.asarray(x, np.float64) for x in (src, ctx, tgt)), CFG)
        return {k: (float(res[k]) if np.ndim(res[k]) == 0 else res[k]) for k in ("loss", "simloss", "recon1", "recon2", "out", "out2")}

    def last_outputs(self, out=True, out2=False, tgt=False):
        return self._last[0], None, self._last[1]

    def save(self, path, prefix=""):
        self.saved.append(path)
        np.savez(path + ".npz", **{prefix + k: v for k, v in self.p.items()})


def make_vdata(seed=0, lattice=True):
    rng = np.random.default_rng(seed)
    u8 = rng.integers(0, 256, (NLEN + 1, NVID, H, W, 3), dtype=np.uint8)
    return u8 / 127.5 - 1.0 if lattice else rng.uniform(-1, 1, u8.shape)


def reference_loop(vdata, seed, basedir):
    """train_script.py:144-203, statement by statement, with sess.run -> the oracle."""
    test = OracleModel(seed)
    lines, rows, validloss = [], [], []
    batch_size, nlen, ntrain = B, NLEN, NTRAIN
    n = vdata.shape[1]
    nvalid = n - ntrain
    validdata = vdata[:, ntrain:]
    traindata = vdata[:, :ntrain]

    def nnerr(tgt, out):
        d = np.mean((np.asarray(tgt)[:, None] - np.asarray(out)[None]) ** 2, axis=(2, 3, 4))        # [i, j]: tgt_i vs out_j (:148)
        return int(np.sum(np.abs(np.argmin(d, axis=0) - np.arange(0, batch_size) % nlen)))

    for itr in range(1, NITR):
        choicesrc = np.random.choice(ntrain, batch_size)
        choicetgt = np.random.choice(ntrain, batch_size)
        srcdata = traindata[np.arange(0, batch_size) % nlen, choicesrc]
        tgtdata = traindata[np.arange(0, batch_size) % nlen, choicetgt]
        tgtctx = traindata[0, choicetgt]
        sc = test.train_step(srcdata, tgtctx, tgtdata, 1e-4)
        if itr % 4 == 0:
            lines.append("%s %s %s %s %s %s" % (itr, sc["loss"], sc["simloss"], sc["recon1"], sc["recon2"], nnerr(tgtdata, test._last[0])))
        if itr % 40 == 0 or itr % SAVE == 0:
            choicesrc = np.random.choice(nvalid, batch_size)
            choicetgt = np.random.choice(nvalid, batch_size)
            srcdata = validdata[np.arange(0, batch_size) % nlen, choicesrc]
            tgtdata = validdata[np.arange(0, batch_size) % nlen, choicetgt]
            tgtctx = validdata[0, choicetgt]
            ev = test.evaluate(srcdata, tgtctx, tgtdata)
            loss, sim, r1, r2, err = ev["loss"], ev["simloss"], ev["recon1"], ev["recon2"], nnerr(tgtdata, ev["out"])
            lines.append("%s %s %s %s %s %s E" % (itr, loss, sim, r1, r2, err))
            validloss.append(loss)
            if itr % SAVE == 0:
                test.saved.append("%s%d/model_%d_%.2f_%.2f_%.2f_%d" % (basedir, itr, itr, loss, r1, r2, err))
                for kk in range(10):
                    choicesrc = [np.random.randint(nvalid)] * batch_size
                    choicetgt = [np.random.randint(nvalid)] * batch_size
                    srcdata = validdata[np.arange(0, batch_size) % nlen, choicesrc]
                    tgtdata = validdata[np.arange(0, batch_size) % nlen, choicetgt]
                    tgtctx = validdata[0, choicetgt]
                    clip = test.evaluate(srcdata, tgtctx, tgtdata)
                    r1, r2 = clip["recon1"], clip["recon2"]          # (:192-193: the clip fetch re-uses the names r1, r2)
            if itr >= SAVE:
                rows.append([itr, loss, sim, r1, r2, err])
    return test, lines, rows, validloss


def test_nn_err_and_lattice_helpers():
    rng = np.random.default_rng(0)
    tgt = rng.standard_normal((6, 4, 4, 3))
    out = tgt[[0, 1, 2, 0, 1, 2]] + 1e-3 * rng.standard_normal((6, 4, 4, 3))
    assert nn_err(tgt, out, 3) == 0                                # output j is nearest to tgt frame j % 3
    assert nn_err(tgt, out[::-1], 3) > 0
    u8, ok = on_u8_lattice(make_vdata())
    assert ok and u8.dtype == np.uint8 and np.array_equal(u8 / 127.5 - 1.0, make_vdata())
    assert on_u8_lattice(make_vdata(lattice=False)) == (None, False)


def test_trainer_equals_the_reference_loop(tmp_path):
    vdata = make_vdata()
    base = str(tmp_path / "run") + "/"
    np.random.seed(7)
    ref, lines, rows, validloss = reference_loop(vdata, seed=3, basedir=base)
    np.random.seed(7)
    got_lines = []
    model = OracleModel(3)
    trainer = ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", NITR, SAVE, NLEN, 1, vdata=vdata, basedir=base,
                           translator=model, log=got_lines.append)
    trainer.train()
    assert got_lines[3:] == lines                                   # after the three shape lines the reference logs too
    assert got_lines[0] == str(vdata.shape) and got_lines[1] == "%s %s" % (NTRAIN, NVID - NTRAIN)
    assert model.saved == ref.saved and len(model.saved) == 2      # itr 20 and 40
    for k in ref.p:
        np.testing.assert_array_equal(model.p[k], ref.p[k])
    assert trainer.validloss == validloss
    np.testing.assert_array_equal(np.load(base + "vdata_train.npy"), vdata[:, :NTRAIN][:, :200])
    np.testing.assert_array_equal(np.load(base + "40/validloss.npy"), validloss)
    assert sorted(f for f in os.listdir(base + "20") if f.startswith("__")) == sorted(
        [f"__{k}{t}.gif" for k in range(10) for t in ("trans", "recon")])      # savegif's names (train_script.py:193-194)
    clip = read_clip(base + "20/__0trans.gif")
    assert clip.dtype == np.uint8 and clip.shape == (NLEN, H, W, 3)
    with open(base + "progress.csv") as f:
        table = list(csv.reader(f))
    assert table[0] == ["Iteration", "Loss", "Sim", "R1", "R2", "NNErr"]
    assert [int(r[0]) for r in table[1:]] == [20, 40]
    np.testing.assert_allclose([[float(x) for x in r] for r in table[1:]], rows, rtol=1e-12)
    # np.random is left where the reference leaves it
    np.random.seed(7)
    reference_loop(vdata, seed=3, basedir=base)
    after_ref = np.random.randint(1 << 30)
    np.random.seed(7)
    ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", NITR, SAVE, NLEN, 1, vdata=vdata, basedir=str(tmp_path / "again"),
                 translator=OracleModel(3), log=lambda s: None).train()
    assert np.random.randint(1 << 30) == after_ref


def test_trainer_rejects_bad_splits(tmp_path):
    with pytest.raises(ValueError):
        ModelTrainer((H, W), NVID, NVID, B, "ContextSkipNew", 5, 5, NLEN, 1, vdata=make_vdata(), basedir=str(tmp_path),
                     translator=OracleModel(0), log=lambda s: None).train()
    with pytest.raises(ValueError):
        ModelTrainer((H, W), NVID, 4, B, "NoSuchModel", 5, 5, NLEN, 1)


@pytest.mark.gpu
def test_trainer_on_the_hip_translator_resident_and_host_paths_agree(tmp_path):
    """The real loop on a tiny ContextSkipNew: the device-resident demo tensor + device sampler (uint8-lattice demos) and the
    host-gather path (same demos, sampler switched off) must log the same numbers and write the same checkpoints; the loss
    must fall; the checkpoint must restore under the reference's extension-less name."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imitation_from_observation_amd import Translator
    Hh = Ww = 16
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (1, 6, Hh, Ww, 3))                  # videos = a base frame drifting over time: learnable
    u8 = np.clip(base + np.arange(6)[:, None, None, None, None] * 9, 0, 255).astype(np.uint8)    # T = 6 frames > nlen = 4: only t < nlen may be sampled
    vdata = u8 / 127.5 - 1.0
    runs = {}
    for mode in ("resident", "host"):
        tr = Translator(Hh, Ww, 32, 32, max_batch=8)
        tr.init_params(5)
        lines = []
        np.random.seed(11)
        t = ModelTrainer((Hh, Ww), 6, 4, 8, "ContextSkipNew", 41, 20, 4, 1, vdata=vdata, basedir=str(tmp_path / mode),
                         translator=_NoSampler(tr) if mode == "host" else tr, log=lines.append)
        t.train()
        runs[mode] = (lines, tr.get_params_flat(), sorted(os.listdir(tmp_path / mode / "40")))
        if mode == "resident":
            ck = [f for f in os.listdir(tmp_path / mode / "40") if f.startswith("model_40_")][0]
            assert ck.endswith(".npz")
            with Translator(Hh, Ww, 32, 32, max_batch=8) as again:
                again.load(str(tmp_path / mode / "40" / ck[:-4]))    # the reference's name, no extension
                np.testing.assert_array_equal(again.get_params_flat(), tr.get_params_flat())
        tr.close()
    assert runs["resident"][0] == runs["host"][0]
    np.testing.assert_array_equal(runs["resident"][1], runs["host"][1])
    assert runs["resident"][2] == runs["host"][2]
    losses = [float(ln.split()[1]) for ln in runs["resident"][0][3:] if not ln.endswith("E")]
    assert losses[-1] < losses[0]


class _NoSampler:
    """A Translator without the device sampler surface (forces ModelTrainer's host-gather path)."""

    def __init__(self, tr):
        self._tr = tr

    def __getattr__(self, name):
        if name in ("load_demos", "train_step_sampled", "eval_sampled"):
            raise AttributeError(name)
        return getattr(self._tr, name)


def test_trainer_builds_the_demo_tensor_from_decoded_videos(tmp_path):
    """ModelTrainer(videos=...): scripts/train_script.py:59-96 on decoded videos (51 frames each, any size) -- frames 1, 1 + nskip, ...
    resized to idims by the restated scipy.misc.imresize, rescaled, stacked [nlen, nvideos, h, w, 3], saved like :95 -- then the
    usual loop on the result (device sampler eligible: the tensor lies on the uint8 lattice)."""
    from imitation_from_observation_amd.demo_pipeline import transform
    rng = np.random.default_rng(9)
    videos = [rng.integers(1, 256, (51, 24, 24, 3), dtype=np.uint8) for _ in range(NVID)]
    nskip = 17                                                    # frames 1, 18, 35 -> nlen = 3
    base = str(tmp_path / "vid") + "/"
    np.random.seed(3)
    lines = []
    t = ModelTrainer((H, W), NVID, NTRAIN, B, "ContextSkipNew", 9, 8, NLEN, nskip, vdata=None, videos=videos, basedir=base,
                     translator=OracleModel(3), log=lines.append)
    t.train()
    saved = np.load(base + "vdata_strike%d.npy" % NVID)            # named after the videos looked at (train_script.py:95)
    assert saved.shape == (NLEN, NVID, H, W, 3)
    np.random.seed(3)
    order = list(range(NVID))
    np.random.shuffle(order)                                      # train_script.py:66: the list is shuffled before the loop
    np.testing.assert_array_equal(saved[1, 2], transform(videos[order[2]][18], H, W, True))
    assert on_u8_lattice(saved)[1]
    assert any(ln.endswith("E") for ln in lines)                  # the loop ran to its validation at itr 8

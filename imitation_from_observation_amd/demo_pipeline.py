"""The demo-tensor pipeline in front of the trainer: `transform` and the video loop of scripts/train_script.py:16-19, 59-96.

What the reference does per demo video (an .mp4 of 51 frames): frames 1, 1 + nskip, ... are resized to `idims` with
`scipy.misc.imresize(image, [h, w])` (default interp 'bilinear'), optionally rescaled to [-1, 1] (`/127.5 - 1`), a video with an
all-black first kept frame is dropped ("rip"), and the kept videos are stacked into `vdata[nlen, nvideos, h, w, 3]` (:92).

What is here:
  * `imresize_bilinear_u8` -- the resize itself.  `scipy.misc.imresize` (removed from scipy since 1.3; not in this image) was a thin
    wrapper: `Image.fromarray(arr).resize((w, h), resample=Image.BILINEAR)` for uint8 RGB input (`toimage` does not rescale uint8
    data).  Pillow's BILINEAR is a separable antialiased triangle filter (support = max(scale, 1) input pixels), horizontal pass then
    vertical pass, each in 22-bit fixed point with round-half-up and a clip to uint8 (libImaging/Resample.c: precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc).  Restated here in numpy integer arithmetic;
    `tests/test_demo_pipeline.py` checks it bit for bit against Pillow where Pillow is importable, and against hand cases always.
  * `transform` -- :16-19.
  * `build_vdata` -- the loop :59-96 over DECODED videos (arrays [frames, H, W, 3] uint8 or a callable that returns them): mp4
    decoding itself needs imageio + ffmpeg, which this image does not have, so the container of frames comes in from outside;
    everything after `vid.get_data(j)` is the reference's.
Host-side integer / float arithmetic only (it runs once per experiment, before training): no device code.
"""
from __future__ import annotations

import numpy as np

_PRECISION_BITS = 32 - 8 - 2          # libImaging/Resample.c: PRECISION_BITS


def _coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter over the whole axis.
    Returns (xmin[out], count[out], kk[out, ksize] int64 fixed point)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    cnt = np.zeros(out_size, np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        x0 = int(center - support + 0.5)
        x0 = max(x0, 0)
        x1 = int(center + support + 0.5)
        x1 = min(x1, in_size)
        n = x1 - x0
        x = np.arange(n)
        w = np.maximum(0.0, 1.0 - np.abs((x + x0 - center + 0.5) * ss))
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        # (int)(w < 0 ? -0.5 + w * 2^22 : 0.5 + w * 2^22): C truncation toward zero
        kk[xx, :n] = np.trunc(np.where(w < 0, -0.5, 0.5) + w * (1 << _PRECISION_BITS)).astype(np.int64)
        xmin[xx], cnt[xx] = x0, n
    return xmin, cnt, kk


def _resample_axis(img, out_size, axis):
    """One 8-bit pass along `axis` (0 = vertical, 1 = horizontal): sum of pixel * k in fixed point, + 2^21, >> 22, clip to [0, 255]."""
    in_size = img.shape[axis]
    xmin, cnt, kk = _coeffs(in_size, out_size)
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        n = int(cnt[xx])
        seg = src[xmin[xx]:xmin[xx] + n]
        acc = np.tensordot(kk[xx, :n], seg, axes=(0, 0)) + (1 << (_PRECISION_BITS - 1))
        out[xx] = acc >> _PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def imresize_bilinear_u8(image, height, width):
    """scipy.misc.imresize(image, [height, width]) for uint8 images [H, W] or [H, W, C] (train_script.py:17): Pillow's BILINEAR
    resample -- horizontal pass, then vertical pass, each rounded to uint8."""
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise TypeError("imresize_bilinear_u8 takes uint8 images [H, W] or [H, W, C] (what imageio hands the reference)")
    out = img
    if out.shape[1] != width:
        out = _resample_axis(out, int(width), 1)
    if out.shape[0] != height:
        out = _resample_axis(out, int(height), 0)
    return np.ascontiguousarray(out)


def transform(image, resize_height, resize_width, rescale):
    """train_script.py:16-19."""
    cropped_image = imresize_bilinear_u8(image, resize_height, resize_width)
    if rescale:
        return np.array(cropped_image) / 127.5 - 1.
    return cropped_image


def inverse_transform(images):
    """train_script.py:20-21."""
    return (images + 1.) / 2.


def build_vdata(videos, idims, nvideos, nlen, nskip, rescale=True, inception=False, log=None, max_fail=10, shuffle=True, return_count=False):
    """train_script.py:59-96 from `videos`, an iterable of decoded demo videos: arrays [nframes, H, W, 3] uint8 or zero-argument
    callables returning one (so that decoding errors are counted like the reference's `except:`).  Only videos of exactly 51 frames
    are used (:72); frames 1, 1 + nskip, ... < 51 are transformed (:74-75); a video whose first kept frames contain an all -1 frame
    -- black -- is dropped unless `inception` (:76-79); it must yield exactly nlen frames (:81).  Stops after `nvideos` videos were
    LOOKED AT (the reference counts every readable 51-frame video, kept or not, :87, :94) or after more than `max_fail` read errors.
    shuffle (default, as the reference): `np.random.shuffle(videos)` on the list first (:66) -- it decides which videos land in the
    train / validation split and moves the global np.random stream the trainer draws its batches from afterwards, exactly as there.
    Returns vdata [nlen, n_kept, h, w, 3] (float64 in [-1, 1] when rescale, else uint8); with return_count also `itr`, the number of
    videos looked at -- the reference names its saved tensor after it (`vdata_strike` + str(itr), :95)."""
    log = log or (lambda s: None)
    if shuffle:
        videos = list(videos)
        np.random.shuffle(videos)                              # train_script.py:66
    idata = [[] for _ in range(nlen)]
    nfail = 0
    itr = 0
    for v in videos:
        try:
            vid = v() if callable(v) else v
            if itr % 100 == 0:
                log("%s %s" % (itr, len(idata[0])))
            if len(vid) == 51:
                frames = []
                for j in range(1, 51, nskip):
                    frame = transform(vid[j], idims[0], idims[1], rescale)
                    if not inception and np.max(frame) == -1:
                        log("rip %s" % itr)
                        frames = []
                        break
                    frames.append(frame)
                if len(frames) != nlen:
                    continue                                   # (the reference `continue`s here without counting the video, :81-82)
                for j, f in enumerate(frames):
                    idata[j].append(f)
            else:
                log("%s" % len(vid))
            itr += 1
        except Exception as e:                                 # noqa: BLE001  (the reference: a bare except that logs and counts, :88-93)
            nfail += 1
            log("Unexpected error: %r" % (e,))
            if nfail > max_fail:
                break
        if itr >= nvideos:
            break
    vdata = np.array(idata)
    log(str(vdata.shape))
    return (vdata, itr) if return_count else vdata

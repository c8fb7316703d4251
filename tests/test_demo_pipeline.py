"""imitation_from_observation_amd.demo_pipeline: `transform` / the video loop of scripts/train_script.py:16-19, 59-96.
The resize is a restatement of what scipy.misc.imresize dispatched to -- Pillow's BILINEAR resample on uint8 images -- checked bit
for bit against Pillow where it is importable (it is in the build container; the test skips that part elsewhere) and against hand
cases that follow from the algorithm's definition."""
import numpy as np
import pytest

from imitation_from_observation_amd.demo_pipeline import build_vdata, imresize_bilinear_u8, inverse_transform, transform


def test_resize_hand_cases():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 10, 3), dtype=np.uint8)
    np.testing.assert_array_equal(imresize_bilinear_u8(img, 12, 10), img)              # same size: untouched
    flat = np.full((9, 7, 3), 137, np.uint8)
    assert (imresize_bilinear_u8(flat, 5, 4) == 137).all() and (imresize_bilinear_u8(flat, 20, 13) == 137).all()   # weights sum to one
    # exact 2x downscale: support 2, centre between two pixels -> weights (1, 3, 3, 1) / 8 inside the image
    row = np.array([[0, 80, 160, 240, 40, 200, 8, 16]], np.uint8)
    got = imresize_bilinear_u8(row, 1, 4)[0]
    interior = (1 * 80 + 3 * 160 + 3 * 240 + 1 * 40) / 8.0
    assert abs(int(got[1]) - interior) <= 0.5 + 1e-9
    edge = (3 * 0 + 3 * 80 + 1 * 160) / 7.0                                            # the window is cut at the border and renormalised
    assert abs(int(got[0]) - edge) <= 0.5 + 1e-9
    # 2x upscale of a ramp stays monotone and keeps the end values
    ramp = np.arange(0, 250, 25, dtype=np.uint8)[None, :]
    up = imresize_bilinear_u8(ramp, 1, 20)[0].astype(int)
    assert (np.diff(up) >= 0).all() and up[0] == 0 and up[-1] == 225
    with pytest.raises(TypeError):
        imresize_bilinear_u8(np.zeros((4, 4, 3), np.float32), 2, 2)


@pytest.mark.parametrize("shape,size", [((64, 64, 3), (48, 48)), ((128, 128, 3), (64, 64)), ((500, 500, 3), (299, 299)), ((36, 64, 3), (36, 64)),
                                         ((100, 80, 3), (36, 64)), ((33, 47), (125, 125)), ((256, 256, 3), (125, 125))])
def test_resize_equals_pillow_bit_for_bit(shape, size):
    Image = pytest.importorskip("PIL.Image")
    img = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((size[1], size[0]), resample=Image.BILINEAR))      # scipy.misc.imresize(img, size)
    np.testing.assert_array_equal(imresize_bilinear_u8(img, *size), want)


def test_transform_and_video_loop():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)
    t = transform(img, 64, 64, True)
    assert t.shape == (64, 64, 3) and t.dtype == np.float64 and t.min() >= -1 and t.max() <= 1
    np.testing.assert_array_equal(t, imresize_bilinear_u8(img, 64, 64) / 127.5 - 1.0)
    assert transform(img, 64, 64, False).dtype == np.uint8
    np.testing.assert_allclose(inverse_transform(t), imresize_bilinear_u8(img, 64, 64) / 255.0, atol=1e-12)
    # the loop: 51-frame videos only; frames 1, 3, ..., 49 (nskip 2 -> nlen 25); a black kept frame drops the video; read errors counted
    good = [rng.integers(1, 256, (51, 40, 40, 3), dtype=np.uint8) for _ in range(4)]
    short = rng.integers(1, 256, (30, 40, 40, 3), dtype=np.uint8)
    black = good[0].copy()
    black[5] = 0                                               # frame 5 is kept (odd index) and all black -> "rip"

    def broken():
        raise IOError("cannot decode")

    logs = []
    vdata = build_vdata([good[0], short, black, broken, good[1], good[2], good[3]], (32, 32), nvideos=4, nlen=25, nskip=2, log=logs.append, shuffle=False)
    # looked at: good0 (kept, 1), short (counted, not kept, 2), black (`continue`: NOT counted), broken (error), good1 (3), good2 (4) -> stop before good3
    assert vdata.shape == (25, 3, 32, 32, 3) and vdata.dtype == np.float64
    np.testing.assert_array_equal(vdata[0, 0], transform(good[0][1], 32, 32, True))
    np.testing.assert_array_equal(vdata[24, 2], transform(good[2][49], 32, 32, True))
    assert any(s.startswith("rip") for s in logs) and any("Unexpected error" in s for s in logs) and logs[-1] == str(vdata.shape)
    u8 = build_vdata(good, (40, 40), nvideos=4, nlen=25, nskip=2, rescale=False, shuffle=False)
    assert u8.dtype == np.uint8 and u8.shape == (25, 4, 40, 40, 3)
    np.testing.assert_array_equal(u8[3, 1], good[1][7])        # same size: frames pass through untouched
    # and the tensor feeds the trainer's device sampler: it lies on the uint8 lattice
    from imitation_from_observation_amd.trainer import on_u8_lattice
    k, ok = on_u8_lattice(build_vdata(good, (32, 32), nvideos=4, nlen=25, nskip=2))
    assert ok and k.dtype == np.uint8


def test_build_vdata_shuffles_like_the_reference_and_reports_the_videos_looked_at():
    """train_script.py:66 `np.random.shuffle(videos)` decides the video order (hence the train / validation split) and moves the
    global np.random stream; :95 names the saved tensor after `itr`, the videos LOOKED AT -- not the videos kept."""
    rng = np.random.default_rng(5)
    good = [rng.integers(1, 256, (51, 20, 20, 3), dtype=np.uint8) for _ in range(6)]
    short = rng.integers(1, 256, (30, 20, 20, 3), dtype=np.uint8)            # counted (itr += 1), not kept
    vids = good[:3] + [short] + good[3:]
    np.random.seed(11)
    vdata, looked_at = build_vdata(vids, (20, 20), nvideos=7, nlen=25, nskip=2, return_count=True)
    after = np.random.get_state()[1][:4].copy()
    np.random.seed(11)
    order = list(range(7))
    np.random.shuffle(order)                                                  # the same permutation the list got
    np.testing.assert_array_equal(np.random.get_state()[1][:4], after)       # the stream is left where the reference leaves it
    kept = [i for i in order if i != 3]
    assert looked_at == 7 and vdata.shape[1] == 6
    for col, i in enumerate(kept):
        np.testing.assert_array_equal(vdata[0, col], transform(vids[i][1], 20, 20, True))
    assert order != sorted(order)

"""The numpy restatement of the reference's video transform (oracle/video_ref.py) pinned bit-exactly against the
installed Pillow, which is what torchvision's ToPILImage / Resize / Grayscale call (torchvision itself is absent)."""
import numpy as np
import pytest

from oracle import video_ref as V

PIL = pytest.importorskip('PIL.Image')


def _frames(t=3, h=288, w=360, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (t, h // 8, w // 8, 3)).astype(np.uint8)
    f = np.repeat(np.repeat(base, 8, 1), 8, 2)                       # blocky image: edges exercise the filter
    noise = rng.randint(-20, 21, f.shape)
    return np.clip(f.astype(np.int64) + noise, 0, 255).astype(np.uint8)


def test_resize_and_gray_match_pillow_bit_exactly():
    for f in _frames():
        img = PIL.fromarray(f).crop(V.CROP)
        assert img.size == (136, 136)
        ref = img.resize((112, 112), PIL.BILINEAR)
        l, u, r, lo = V.CROP
        mine = V.resize_bilinear_u8(f[u:lo, l:r])
        assert np.array_equal(mine, np.asarray(ref))
        assert np.array_equal(V.rgb_to_l(mine), np.asarray(ref.convert('L')))


def test_build_tensor_layout_padding_flip_erase():
    fr = _frames(4)
    vol = V.build_tensor(fr, 6, flip=False, erase=None)
    assert vol.shape == (1, 6, 112, 112) and vol.dtype == np.float32
    assert np.all(vol[0, 4:] == 0.0)                                   # zero-initialised volume behind the clip
    flipped = V.build_tensor(fr, 6, flip=True)
    assert np.array_equal(flipped[0, :4], vol[0, :4, :, ::-1])
    er = V.build_tensor(fr, 6, erase=(-10, 60))
    assert np.all(er[0, :, 60:112, 0:46] == 0.0) and np.array_equal(er[0, :4, :60], vol[0, :4, :60])
    # value range of Normalize(0.4136, 0.17) on [0, 1]
    assert vol.min() >= (0 - V.MEAN) / V.STD - 1e-5 and vol.max() <= (1 - V.MEAN) / V.STD + 1e-5

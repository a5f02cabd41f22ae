"""The output step and the image decoder of the host layer (camera.rs:238,270 writes a PNG through the `image` crate;
texture.rs:89-93 decodes the texture file through it): the PNG writer round-trips through an independent decoder
(PIL) bit for bit, and the PNG / PPM reader returns what PIL returns for files PIL wrote, whatever colour type,
bit depth and row filters the encoder chose."""
import os

import numpy as np
import pytest
from PIL import Image

from raytracer_amd import HostBuilder
from raytracer_amd.builder import read_image, write_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _noise(h, w, seed=0):
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    img[: h // 2] = (np.add.outer(np.arange(h // 2), np.arange(w))[..., None] * np.array([1, 2, 3])) % 256  # smooth part
    return img


@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (64, 48), (225, 400)])
def test_write_png_round_trips_through_pil(tmp_path, shape):
    img = _noise(*shape)
    path = tmp_path / "out.png"
    write_png(path, img)
    back = Image.open(path)
    assert back.mode == "RGB" and back.size == (shape[1], shape[0])
    assert np.array_equal(np.asarray(back), img)
    assert np.array_equal(read_image(path), img)          # and through the library's own decoder


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
def test_read_png_equals_pil_for_every_colour_type(tmp_path, mode):
    img = Image.fromarray(_noise(37, 53, seed=3))
    if mode == "P":
        pil = img.quantize(colors=200)
    elif mode == "1":
        pil = img.convert("L").point(lambda v: 255 if v > 127 else 0).convert("1")
    else:
        pil = img.convert(mode)
    path = tmp_path / f"{mode}.png"
    pil.save(path)                                          # PIL picks the row filters (all five types occur)
    want = np.asarray(Image.open(path).convert("RGB"))
    got = read_image(path)
    assert got is not None and np.array_equal(got, want)


def test_read_png_16_bit_keeps_the_high_byte_and_low_depths_are_scaled(tmp_path):
    rs = np.random.RandomState(5)
    a16 = rs.randint(0, 65536, (20, 30)).astype(np.uint16)
    path = tmp_path / "g16.png"
    Image.fromarray(a16).save(path)                         # mode I;16: 16-bit grey
    got = read_image(path)
    assert np.array_equal(got[..., 0], (a16 >> 8).astype(np.uint8)) and np.array_equal(got[..., 0], got[..., 2])
    for bits in (2, 4):
        vals = rs.randint(0, 1 << bits, (9, 13)).astype(np.uint8)
        p = tmp_path / f"g{bits}.png"
        Image.fromarray(vals * (255 // ((1 << bits) - 1))).save(p, bits=bits)
        assert np.array_equal(read_image(p)[..., 1], vals * (255 // ((1 << bits) - 1)))


def test_read_ppm_and_refusals(tmp_path):
    ppm = os.path.join(ROOT, "assets", "earthmap.ppm")
    assert np.array_equal(read_image(ppm), np.asarray(Image.open(ppm).convert("RGB")))
    img = _noise(16, 16)
    good = tmp_path / "good.png"
    Image.fromarray(img).save(good)
    data = bytearray(good.read_bytes())
    data[len(data) // 2] ^= 0x40                            # a flipped bit inside IDAT: the chunk's CRC no longer matches
    bad = tmp_path / "bad.png"
    bad.write_bytes(bytes(data))
    assert read_image(bad) is None
    inter = tmp_path / "truncated.png"
    inter.write_bytes(good.read_bytes()[:40])               # truncated
    assert read_image(inter) is None
    assert read_image(tmp_path / "missing.png") is None
    gif = tmp_path / "x.gif"
    Image.fromarray(img).save(gif)
    assert read_image(gif) is None                          # not a format the reference's scenes use: refused


def test_image_texture_from_png_equals_the_one_from_ppm(tmp_path):
    """ImageTexture::new_or_fallback reads a PNG as it reads the PPM of the same picture; an unreadable file gives the
    reference's magenta checker (texture.rs:94-111)."""
    ppm = os.path.join(ROOT, "assets", "earthmap.ppm")
    png = tmp_path / "earthmap.png"
    Image.open(ppm).save(png)

    def texels(path):
        b = HostBuilder()
        t = b.tex_image_file(path)
        b.set_world(b.slice_of([b.sphere((0, 0, 0), 1.0, b.mat_lambertian(t))]))
        p, n = b.flat_ptr(19)
        kinds = [x.kind for x in b.flat("textures")]
        arr = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (3 * n)).from_address(p)).copy() if n else np.zeros(0, np.uint8)
        return kinds, arr

    k1, a1 = texels(ppm)
    k2, a2 = texels(png)
    assert k1 == k2 and 2 in k1 and a1.size > 0 and np.array_equal(a1, a2)    # 2 = RT_TEX_IMAGE
    k3, a3 = texels(tmp_path / "nothing.png")
    assert 2 not in k3 and 1 in k3 and a3.size == 0                            # 1 = RT_TEX_CHECKER: the fallback


def _smooth(h, w, seed=0):
    """A natural-looking image (JPEG is compared on content it is made for: smooth gradients and soft edges)."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 17.0 + seed) * np.cos(yy / 23.0), 127 + 90 * np.sin((xx + yy) / 31.0),
                    127 + 110 * np.cos(xx / 11.0) * np.sin(yy / 13.0 + 1.0)], axis=2)
    img += rs.normal(0, 4.0, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling, shape, extra", [(0, (64, 48), {}), (1, (61, 83), {}), (2, (225, 400), {}), (2, (17, 9), {}),
                                                       (2, (96, 128), {"restart_marker_blocks": 3}), (0, (40, 40), {"optimize": True})])
def test_read_jpeg_agrees_with_pil_within_decoder_tolerance(tmp_path, subsampling, shape, extra):
    """Baseline JPEG as the reference's texture files are (textures/earthmap.jpg, texture.rs:90-93): 4:4:4, 4:2:2 and
    4:2:0 chroma, sizes that are not multiples of the block size, restart markers, optimised Huffman tables.  Two
    decoders agree on a JPEG to within a level or two per sample (inverse transform and chroma filter differ in their
    rounding), not bit for bit: the bound is on the largest and the mean difference against libjpeg's result."""
    img = _smooth(*shape, seed=subsampling)
    path = tmp_path / "t.jpg"
    Image.fromarray(img).save(path, "JPEG", quality=92, subsampling=subsampling, **extra)
    want = np.asarray(Image.open(path).convert("RGB")).astype(int)
    got = read_image(path)
    assert got is not None and got.shape == want.shape
    diff = np.abs(got.astype(int) - want)
    assert diff.max() <= 4 and diff.mean() < 0.6, (diff.max(), diff.mean())
    assert np.abs(got.astype(int) - img.astype(int)).mean() < 6.0      # and it is the picture that was encoded


def test_read_jpeg_greyscale_and_unsupported_kinds(tmp_path):
    img = _smooth(50, 70, seed=4)
    g = tmp_path / "g.jpg"
    Image.fromarray(img).convert("L").save(g, "JPEG", quality=95)
    want = np.asarray(Image.open(g).convert("RGB")).astype(int)
    got = read_image(g)
    assert got is not None and np.abs(got.astype(int) - want).max() <= 2 and np.array_equal(got[..., 0], got[..., 1])
    # a progressive file is not read: ImageTexture::new_or_fallback then takes the magenta checker (texture.rs:94-111)
    p = tmp_path / "p.jpg"
    Image.fromarray(img).save(p, "JPEG", quality=90, progressive=True)
    assert read_image(p) is None
    b = HostBuilder()
    t = b.tex_image_file(str(p))
    assert t >= 0
    # truncated and corrupt files are refused, not crashed on
    data = g.read_bytes()
    for k, cut in enumerate((10, len(data) // 3, len(data) - 5)):
        q = tmp_path / f"cut{k}.jpg"
        q.write_bytes(data[:cut])
        r = read_image(q)
        assert r is None or r.shape == (50, 70, 3)
    bad = bytearray(data)
    for i in range(200, len(bad), 7):
        bad[i] ^= 0x5A
    q = tmp_path / "bad.jpg"
    q.write_bytes(bytes(bad))
    r = read_image(q)
    assert r is None or r.shape == (50, 70, 3)

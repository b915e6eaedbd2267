"""Detector::load_image's file decoder (csrc/y2_imgfile.cpp; reference yolo_v2_class.cpp:127-149 = stbi_load(.., 3)) against
Pillow's decode of the same files.  No GPU needed.  PNG and PNM are exact; baseline JPEG is compared at the tolerance two
conforming decoders agree to (IDCT and chroma-upsampling arithmetic differ by an LSB or two) -- parity with stb_image
itself is unpinned: the reference's decoder cannot be run here."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from sr_object_detection_amd import darknet

PIL = pytest.importorskip("PIL.Image")


def decode(path):
    L = darknet.lib()
    L.y2_decode_image_rgb.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_ubyte)), C.c_char_p, C.c_int]
    L.y2_free_image_rgb.argtypes = [C.POINTER(C.c_ubyte)]
    w, h, p = C.c_int(0), C.c_int(0), C.POINTER(C.c_ubyte)()
    err = C.create_string_buffer(256)
    if L.y2_decode_image_rgb(path.encode(), C.byref(w), C.byref(h), C.byref(p), err, 256) != 0:
        raise RuntimeError(err.value.decode())
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.y2_free_image_rgb(p)
    return a


def picture(w, h, seed=1):
    """smooth structure + noise, so that JPEG blocks carry real AC coefficients"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 13.0), 128 + 90 * np.cos(x / 5.0 + y / 7.0), 40 + 200.0 * x / max(w - 1, 1)], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


def stb3(im):
    """what stbi_load(file, 3 channels) yields from a Pillow image: grey replicated, alpha dropped, 16 bit -> high byte"""
    if im.mode in ("I;16", "I;16B", "I"):
        a = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
        return np.stack([a, a, a], -1)
    if im.mode == "1":
        a = np.asarray(im.convert("L"))
        return np.stack([a, a, a], -1)
    if im.mode in ("L", "LA"):
        a = np.asarray(im.convert("LA"))[..., 0]
        return np.stack([a, a, a], -1)
    if im.mode == "P":
        return np.asarray(im.convert("RGB"))
    return np.asarray(im.convert("RGBA"))[..., :3] if im.mode == "RGBA" else np.asarray(im.convert("RGB"))


@pytest.mark.parametrize("mode,size,opts", [
    ("RGB", (67, 45), {}), ("RGBA", (33, 20), {}), ("L", (50, 31), {}), ("LA", (17, 9), {}), ("P", (40, 40), {}),
    ("RGB", (129, 70), {"compress_level": 0}), ("RGB", (300, 211), {"compress_level": 9}), ("1", (37, 11), {}),
    ("I;16", (21, 14), {}), ("RGB", (8, 8), {}), ("RGB", (1, 1), {}), ("L", (3, 200), {}),
])
def test_png_is_exact(tmp_path, mode, size, opts):
    w, h = size
    rgb = picture(w, h, w * 7 + h)
    if mode == "I;16":
        im = PIL.fromarray((rgb[..., 0].astype(np.uint16) * 257 + 31).astype(np.uint16))
    elif mode == "1":
        im = PIL.fromarray(rgb[..., 0] > 128)
    elif mode == "P":
        im = PIL.fromarray(rgb).quantize(40)
    else:
        im = PIL.fromarray(rgb).convert(mode)
        if mode in ("RGBA", "LA"):
            im.putalpha(PIL.fromarray(rgb[..., 1]))
    path = str(tmp_path / "t.png")
    im.save(path, **opts)
    got = decode(path)
    want = stb3(PIL.open(path))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_interlaced_and_low_bit_depth_png_by_hand(tmp_path):
    """Adam7 and 2- / 4-bit grey written by hand (Pillow does not write them): the decoder's pass geometry and bit unpacking"""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    def write(path, w, h, depth, ctype, interlace, rows_by_pass):
        raw = b"".join(b"\x00" + r for rows in rows_by_pass for r in rows)
        open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) +
                               chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

    w, h = 13, 11
    img = picture(w, h, 5)
    xs, ys, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    passes = []
    for p in range(7):
        sub = img[ys[p]::dy[p], xs[p]::dx[p]]
        passes.append([row.tobytes() for row in sub] if sub.size else [])
    path = str(tmp_path / "adam7.png")
    write(path, w, h, 8, 2, 1, passes)
    assert np.array_equal(decode(path), img)
    for depth in (2, 4):
        vals = (img[..., 0] >> (8 - depth)).astype(np.uint8)
        rows = []
        for r in vals:
            bits = "".join(format(int(v), "0%db" % depth) for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        path = str(tmp_path / ("g%d.png" % depth))
        write(path, w, h, depth, 0, 0, [rows])
        want = (vals.astype(np.int32) * 255 // ((1 << depth) - 1)).astype(np.uint8)
        assert np.array_equal(decode(path), np.stack([want] * 3, -1))


@pytest.mark.parametrize("size", [(64, 48), (67, 45), (161, 99), (16, 16), (9, 70)])
@pytest.mark.parametrize("sub,grey,restart", [(0, False, 0), (1, False, 0), (2, False, 0), (2, False, 3), (0, True, 0), (2, False, 1)],
                         ids=["444", "422", "420", "420_rst3", "grey", "420_rst1"])
def test_baseline_jpeg_matches_pillow(tmp_path, size, sub, grey, restart):
    w, h = size
    im = PIL.fromarray(picture(w, h, w + h + sub))
    if grey:
        im = im.convert("L")
    path = str(tmp_path / "t.jpg")
    kw = dict(quality=88, subsampling=sub) if not grey else dict(quality=88)
    if restart:
        kw["restart_marker_blocks"] = restart
    im.save(path, **kw)
    got = decode(path).astype(np.int32)
    want = stb3(PIL.open(path)).astype(np.int32)
    assert got.shape == want.shape
    d = np.abs(got - want)
    assert d.max() <= 4 and d.mean() < 0.6, (int(d.max()), float(d.mean()))


def test_progressive_jpeg_and_unknown_files_are_refused(tmp_path):
    path = str(tmp_path / "p.jpg")
    PIL.fromarray(picture(40, 30)).save(path, progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        decode(path)
    junk = str(tmp_path / "junk.bin")
    open(junk, "wb").write(b"hello world, not an image")
    with pytest.raises(RuntimeError, match="unknown image format"):
        decode(junk)
    with pytest.raises(RuntimeError, match="file not found"):
        decode(str(tmp_path / "missing.png"))
    trunc = str(tmp_path / "trunc.png")
    PIL.fromarray(picture(60, 60)).save(trunc)
    data = open(trunc, "rb").read()
    open(trunc, "wb").write(data[:len(data) // 2])
    with pytest.raises(RuntimeError):
        decode(trunc)


def test_pnm(tmp_path):
    img = picture(31, 17)
    p6 = str(tmp_path / "a.ppm")
    open(p6, "wb").write(b"P6\n# a comment\n31 17\n255\n" + img.tobytes())
    assert np.array_equal(decode(p6), img)
    p5 = str(tmp_path / "a.pgm")
    open(p5, "wb").write(b"P5 31 17 255\n" + img[..., 0].tobytes())
    assert np.array_equal(decode(p5), np.stack([img[..., 0]] * 3, -1))


def test_damaged_files_are_refused_or_decoded_never_fatal(tmp_path):
    """truncations and byte flips of valid PNG / JPEG files: every one either decodes or raises; bounds are checked on every
    table, chunk, code and window reference (the same corpus runs clean under ASan + UBSan, profiles/r03_notes.md)"""
    rng = np.random.default_rng(3)
    img = (rng.random((40, 52, 3)) * 255).astype(np.uint8)
    seeds = []
    for fmt, kw in (("PNG", {}), ("JPEG", dict(quality=80)), ("JPEG", dict(quality=80, subsampling=0, restart_marker_blocks=2))):
        b = io.BytesIO()
        PIL.fromarray(img).save(b, fmt, **kw)
        seeds.append(b.getvalue())
    path = str(tmp_path / "f.bin")
    decoded = refused = 0
    for s in seeds:
        for _ in range(150):
            d = bytearray(s)
            if rng.integers(0, 4) == 0:
                d = d[:rng.integers(1, len(d))]
            else:
                for _ in range(rng.integers(1, 6)):
                    d[rng.integers(0, len(d))] = rng.integers(0, 256)
            open(path, "wb").write(bytes(d))
            try:
                a = decode(path)
                assert a.ndim == 3 and a.shape[2] == 3
                decoded += 1
            except RuntimeError:
                refused += 1
    assert decoded + refused == 450 and refused > 50

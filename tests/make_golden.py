#!/usr/bin/env python3
"""Regenerates the committed fixtures that come from the reference's own
compilable sources (run in the build container only, where /root/reference and
oracle/_ref/ref_fixtures exist):

  tests/golden/micro.comp5892mesh + micro_tex/*.png  the micro scene in the reference's file format
  tests/golden/ref_loadmodel.json   what the reference's load_baked_model (main/baked_model.cpp) reads from it
  tests/golden/png/*.png            small PNGs of every colour type / bit depth
  tests/golden/ref_stb.json         what stbi_load(path,...,4) + vertical flip (utils/vkimage.cpp:62-65) decodes them to
"""
import os
import subprocess
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkanrenderer_amd import baked_model, scenes  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_fixtures")


def write_micro():
    sc = scenes.micro_scene()
    os.makedirs(os.path.join(GOLD, "micro_tex"), exist_ok=True)
    tex = []
    for i, t in enumerate(sc.textures):
        rel = "micro_tex/t%02d.png" % i
        # stored top-down on disk; the loader flips vertically, so write the flipped image
        Image.fromarray(np.ascontiguousarray(t["rgba8"][::-1]), "RGBA").save(os.path.join(GOLD, rel), optimize=True)
        tex.append((rel, t["space"], t["channels"]))
    baked_model.write_baked_model(os.path.join(GOLD, "micro.comp5892mesh"), tex, sc.materials, sc.meshes)


def write_pngs():
    d = os.path.join(GOLD, "png")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(11)
    g = rng.integers(0, 256, (5, 7), dtype=np.uint8)
    Image.fromarray(g, "L").save(os.path.join(d, "grey8.png"))
    Image.fromarray(np.dstack([g, rng.integers(0, 256, (5, 7), dtype=np.uint8)]), "LA").save(os.path.join(d, "greyalpha8.png"))
    Image.fromarray(rng.integers(0, 256, (6, 5, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "rgb8.png"))
    Image.fromarray(rng.integers(0, 256, (4, 9, 4), dtype=np.uint8), "RGBA").save(os.path.join(d, "rgba8.png"))
    Image.fromarray((rng.integers(0, 2, (9, 13), dtype=np.uint8) * 255), "L").convert("1").save(os.path.join(d, "grey1.png"))
    p = Image.fromarray(rng.integers(0, 16, (8, 8), dtype=np.uint8), "P")
    pal = rng.integers(0, 256, 48, dtype=np.uint8).tolist()
    p.putpalette(pal + [0] * (768 - 48))
    p.save(os.path.join(d, "pal8.png"))
    p.save(os.path.join(d, "pal8_trns.png"), transparency=bytes(rng.integers(0, 256, 16, dtype=np.uint8).tolist()))
    Image.fromarray(rng.integers(0, 65536, (6, 6), dtype=np.uint16)).save(os.path.join(d, "grey16.png"))
    Image.fromarray(np.full((1, 1, 4), 255, np.uint8), "RGBA").save(os.path.join(d, "one.png"))


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))  # x0, y0, dx, dy


def encode_png(samples, depth, ctype, interlace, plte=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """A PNG writer for the fixtures (Pillow writes no interlaced files): `samples` is H x W x channels of raw sample
    values; scanlines take the filter types of `filters` in turn."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    H, W, C = samples.shape
    bpp = max(1, C * depth // 8)

    def pack_row(row):  # row: w x C samples -> bytes
        flat = row.reshape(-1).astype(np.uint32)
        if depth == 16:
            return b"".join(struct.pack(">H", int(v)) for v in flat)
        if depth == 8:
            return bytes(int(v) for v in flat)
        out, acc, n = bytearray(), 0, 0
        for v in flat:
            acc = (acc << depth) | int(v); n += depth
            if n == 8:
                out.append(acc); acc, n = 0, 0
        if n:
            out.append(acc << (8 - n))
        return bytes(out)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    def filtered(rows):  # packed rows of one (sub-)image -> filtered scanlines
        out, prev = bytearray(), None
        for y, cur in enumerate(rows):
            ft = filters[y % len(filters)]
            out.append(ft)
            for i, v in enumerate(cur):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i] if prev is not None else 0
                c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
                pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
                out.append((v - pred) & 255)
            prev = cur
        return bytes(out)

    if interlace:
        raw = b""
        for x0, y0, dx, dy in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filtered([pack_row(r) for r in sub])
    else:
        raw = filtered([pack_row(r) for r in samples])
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        png += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        png += chunk(b"tRNS", bytes(trns))
    half = len(zlib.compress(raw, 9)) // 2
    z = zlib.compress(raw, 9)
    return png + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")   # two IDAT chunks on purpose


def write_interlaced_pngs():
    """Adam7 files of every kind the decoder handles, with sizes that leave some of the seven passes empty, every
    scanline filter, and two non-interlaced files of kinds Pillow does not write (4-bit palette, 16-bit RGB with tRNS)."""
    d = os.path.join(GOLD, "png")
    rng = np.random.default_rng(12)

    def put(name, *a, **k):
        with open(os.path.join(d, name), "wb") as f:
            f.write(encode_png(*a, **k))

    put("adam7_rgba8.png", rng.integers(0, 256, (11, 13, 4)), 8, 6, True)
    put("adam7_rgb8.png", rng.integers(0, 256, (8, 8, 3)), 8, 2, True)
    put("adam7_grey1.png", rng.integers(0, 2, (5, 9, 1)), 1, 0, True)
    put("adam7_grey2.png", rng.integers(0, 4, (7, 6, 1)), 2, 0, True)
    put("adam7_pal4_trns.png", rng.integers(0, 16, (3, 7, 1)), 4, 3, True, plte=rng.integers(0, 256, 48).tolist(), trns=rng.integers(0, 256, 9).tolist())
    put("adam7_rgb16_trns.png", np.concatenate([rng.integers(0, 65536, (9, 5, 3)), np.full((1, 5, 3), 4660)]), 16, 2, True, trns=[0x12, 0x34] * 3)
    put("adam7_greyalpha8_1x1.png", rng.integers(0, 256, (1, 1, 2)), 8, 4, True)
    put("adam7_greyalpha16_2x3.png", rng.integers(0, 65536, (3, 2, 2)), 16, 4, True)
    put("adam7_grey8_3x1.png", rng.integers(0, 256, (1, 3, 1)), 8, 0, True)
    put("adam7_rgba16_33x17.png", rng.integers(0, 65536, (17, 33, 4)), 16, 6, True)
    put("pal4_trns.png", rng.integers(0, 16, (6, 5, 1)), 4, 3, False, plte=rng.integers(0, 256, 48).tolist(), trns=rng.integers(0, 256, 5).tolist())
    put("rgb16_trns.png", np.concatenate([rng.integers(0, 65536, (4, 6, 3)), np.full((1, 6, 3), 513)]), 16, 2, False, trns=[0x02, 0x01] * 3)
    put("grey4_trns.png", rng.integers(0, 16, (5, 5, 1)), 4, 0, False, trns=[0, 7])


def main():
    if "--pngs-only" not in sys.argv:
        write_micro()
        write_pngs()
    write_interlaced_pngs()
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_fixtures missing: run `make -C oracle _ref` where /root/reference exists")
    with open(os.path.join(GOLD, "ref_loadmodel.json"), "w") as f:
        subprocess.check_call([REF, "loadmodel", "tests/golden/micro.comp5892mesh"], stdout=f, cwd=ROOT)
    pngs = sorted(os.path.join("tests/golden/png", x) for x in os.listdir(os.path.join(GOLD, "png")) if x.endswith(".png"))
    with open(os.path.join(GOLD, "ref_stb.json"), "w") as f:
        subprocess.check_call([REF, "stb"] + pngs, stdout=f, cwd=ROOT)
    tex = sorted(os.path.join("tests/golden/micro_tex", x) for x in os.listdir(os.path.join(GOLD, "micro_tex")))
    with open(os.path.join(GOLD, "ref_stb_micro.json"), "w") as f:
        subprocess.check_call([REF, "stb"] + tex, stdout=f, cwd=ROOT)


if __name__ == "__main__":
    main()

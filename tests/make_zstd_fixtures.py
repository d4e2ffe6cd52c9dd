#!/usr/bin/env python3
"""Writes tests/golden/zstd/*.zst + index.json: Zstandard frames made by libzstd (through pyarrow's codec, the only
compressor in this image) at several levels over data that exercises every block and table type, with the SHA-256 and
length of what each must decode to.  The bake library's own decoder (vkr_zstd_decode) is tested against them."""
import hashlib, json, os, sys
import numpy as np
import pyarrow as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "golden", "zstd")


def main():
    os.makedirs(D, exist_ok=True)
    rng = np.random.default_rng(0x5A57)
    text = open(os.path.join(ROOT, "tests", "golden", "bake", "model.obj"), "rb").read()
    cases = {
        "obj_l1": (text, 1), "obj_l19": (text, 19), "obj_x30_l3": (text * 30, 3), "obj_x200_l22": (text * 200, 22),
        "random_l3": (rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(), 3),                  # raw blocks
        "runs_l5": (b"\x07" * 300000 + b"abc" * 1000 + b"\x00" * 150000, 5),                      # RLE blocks, long matches
        "tiny_l3": (b"hello", 3), "empty_l3": (b"", 3),
        "lowentropy_l1": ((rng.integers(0, 3, 200000, dtype=np.uint8) + 65).tobytes(), 1),        # Huffman literals, 4 streams
        "mixed_l19": (rng.integers(0, 4, 60000, dtype=np.uint8).tobytes() + text * 4 + rng.integers(0, 256, 9000, dtype=np.uint8).tobytes(), 19),
    }
    index = {}
    for name, (data, level) in cases.items():
        z = pa.Codec("zstd", compression_level=level).compress(data, asbytes=True)
        open(os.path.join(D, name + ".zst"), "wb").write(z)
        index[name] = {"level": level, "length": len(data), "sha256": hashlib.sha256(data).hexdigest(), "compressed": len(z)}
    json.dump(index, open(os.path.join(D, "index.json"), "w"), indent=1)
    print(sum(v["compressed"] for v in index.values()), "bytes of frames")


if __name__ == "__main__":
    main()

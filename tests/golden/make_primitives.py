"""Generate tests/golden/primitives.json from the REFERENCE's own code (oracle/_ref/libvcref.so =
Pilaf/image_tools.h + src/bitmap.cc compiled where they lie under /root/reference).

Run in the authoring container only (the reference does not travel):
    make -C oracle ref && python tests/golden/make_primitives.py
The output is data only: inputs and the reference's outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vc_oracle as vo  # noqa: E402

R = vo.ref()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
rng = np.random.default_rng(20131011)

out = {"source": "reference: Pilaf/image_tools.h:12-33, src/bitmap.cc:22-38", "hamming": [], "binary_to_int": [], "bitmap": []}

# compute_hamming_dist: sizes incl. non-multiples of 4 (trailing bytes ignored), all-zero / all-one / equal inputs
for nb in (4, 8, 16, 32, 64, 7, 12, 18, 3):
    cases = [np.zeros(nb, np.uint8), np.full(nb, 255, np.uint8)]
    pairs = [(cases[0], cases[1]), (cases[1], cases[1])]
    for _ in range(24):
        a = rng.integers(0, 256, nb, dtype=np.uint8)
        b = a.copy() if rng.random() < 0.2 else rng.integers(0, 256, nb, dtype=np.uint8)
        for _ in range(rng.integers(0, 6)):
            bit = rng.integers(0, nb * 8)
            b[bit // 8] ^= np.uint8(1 << (bit % 8))
        pairs.append((a, b))
    for a, b in pairs:
        d = R.vcref_hamming(a.tobytes(), b.tobytes(), nb)
        out["hamming"].append({"a": a.tobytes().hex(), "b": b.tobytes().hex(), "dist": int(d)})

# binaryToInt: every length 1..4, top byte on both sides of 0x80 (sign-extension quirk)
for ln in (1, 2, 3, 4):
    fixed = [bytes([0] * ln), bytes([0xFF] * ln), bytes([0x7F] * ln), bytes([0x80] * ln), bytes([1] + [0] * (ln - 1)),
             bytes([0] * (ln - 1) + [0x80]), bytes([0] * (ln - 1) + [0x7F])]
    rnd = [rng.integers(0, 256, ln, dtype=np.uint8).tobytes() for _ in range(40)]
    for p in fixed + rnd:
        out["binary_to_int"].append({"bytes": p.hex(), "value": int(R.vcref_binary_to_int(p, ln))})

# ImageBitmap: op traces on small bitmaps, final raw bytes + every get() result
for n_bytes in (4, 64, 1024):
    for _ in range(4):
        h = R.vcref_bitmap_new(n_bytes)
        ops, gets = [], []
        for _ in range(200):
            bit = int(rng.integers(0, n_bytes * 8))
            kind = ["set", "set", "reset", "get"][int(rng.integers(0, 4))]
            if kind == "set":
                R.vcref_bitmap_set(h, bit)
            elif kind == "reset":
                R.vcref_bitmap_reset(h, bit)
            else:
                gets.append(int(R.vcref_bitmap_get(h, bit)))
            ops.append([kind, bit])
        import ctypes
        raw = ctypes.string_at(R.vcref_bitmap_data(h), n_bytes)
        R.vcref_bitmap_free(h)
        out["bitmap"].append({"n_bytes": n_bytes, "ops": ops, "gets": gets, "raw": raw.hex()})

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "primitives.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(path, len(out["hamming"]), len(out["binary_to_int"]), len(out["bitmap"]), os.path.getsize(path))

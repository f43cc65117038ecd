#!/usr/bin/env python3
"""How much does the ONE unverifiable choice under every parity claim matter?  DEV TOOL (CPU only; uses the oracle as what it is: the checker).

The oracle restates TFLite's builtin REFERENCE kernels (variant R: RoundingDivideByPOT, ties away from zero = SURVEY.md 8(c).3's definition of "the
tflite int8 reference").  The reference's script builds tf.lite.Interpreter with default arguments (yoloface/tflite/tflite_prediction.py:23) = the default
op resolver, whose per-channel int8 CONV_2D goes through ruy: right shift ties UPWARD (U), or one single rounding on its portable path (S); with the XNNPACK
delegate on, conv / depthwise requantise in fp32 (X); U-all = ties upward in every op.  None of them can be executed here.  This prints, per input set and
variant, the distance from R -- the table of DESIGN.md section 2, pinned by tests/test_oracle.py::test_rounding_variant_exposure.

    python tools/rounding_exposure.py > profiles/r06_rounding_exposure.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, VARIANTS  # noqa: E402

NAMES = {"U": "U  ties upward, dense CONV_2D (ruy vector kernels)", "U-all": "U-all  ties upward, every op", "X": "X  fp32 requantisation, conv + depthwise (XNNPACK)",
         "S": "S  single rounding, dense CONV_2D (ruy portable path)"}


def main():
    orc = Oracle()
    g = os.path.join(ROOT, "tests", "golden")
    sets = [("the reference's 27 sample images (tests/golden/real_frames_56.bin)", np.fromfile(os.path.join(g, "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)),
            ("the 6 golden frames (tests/golden/golden_inputs.bin)", np.fromfile(os.path.join(g, "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)),
            ("4096 seeded frames, default_rng(1) uniform int8 (BASELINE configs[1]'s input)", np.random.default_rng(1).integers(-128, 128, (4096, 56, 56, 3), dtype=np.int8))]

    def boxes(h, f):
        return [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in orc.decode_py(h, f)]
    for title, x in sets:
        n = len(x)
        ref = orc.run(x, threads=8)
        rb = [boxes(ref[f], f) for f in range(n)]
        print(f"## {title}: {n} frames, {sum(1 for b in rb if b)} with at least one box, {sum(len(b) for b in rb)} boxes under R")
        print("| variant | head bytes that differ | max abs delta (LSB) | frames with a differing head byte | frames whose Python box LIST differs | ... in coordinates only | ... a box appears / disappears | max edge shift (px of 56) |")
        print("|---|---|---|---|---|---|---|---|")
        for name, v in VARIANTS.items():
            if name == "R":
                continue
            h = orc.run(x, threads=8, variant=v)
            d = h.astype(int) - ref.astype(int)
            lists = [boxes(h[f], f) for f in range(n)]
            changed = [f for f in range(n) if lists[f] != rb[f]]
            coords = [f for f in changed if [b[:3] for b in lists[f]] == [b[:3] for b in rb[f]]]
            shift = max([abs(a - b) for f in coords for p, q in zip(lists[f], rb[f]) for a, b in zip(p[3:], q[3:])], default=0)
            print(f"| {NAMES[name]} | {np.count_nonzero(d)} of {d.size} ({100 * np.count_nonzero(d) / d.size:.1f} %) | {np.abs(d).max()} | "
                  f"{int(np.count_nonzero(np.any(d.reshape(n, -1) != 0, axis=1)))} | {len(changed)} | {len(coords)} | {len(changed) - len(coords)} | {shift} |")
        print()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden heads of the OTHER published roundings of TFLite's requantisation (oracle variants, oracle/yf_oracle.h YFO_RV_*), on the inputs of golden_inputs.bin and
real_frames_56.bin.  Same honest label as make_golden.py: self-consistent, interpreter-UNVERIFIED.  They pin each variant bit for bit (a refactor of the oracle, of
the host's constants or of the second kernel set shows here) and give whoever has TensorFlow 2.10 something to compare an interpreter run with.

Files: golden_heads_variants.npz   {"U", "U_all", "S", "X"}: int8 [6][7][7][18] each (the golden inputs)
       golden_variants_meta.json   sha256 of every head of the 27 real frames per variant, and the Python-decode boxes of the golden frames per variant
"""
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Oracle, VARIANTS  # noqa: E402


def main():
    orc = Oracle()
    x = np.fromfile(os.path.join(HERE, "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)
    real = np.fromfile(os.path.join(HERE, "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    heads, meta = {}, {"label": "self-consistent, interpreter-unverified; variants of the requantisation rounding (oracle/yf_oracle.h)", "variants": {}}
    for name, v in VARIANTS.items():
        if name == "R":
            continue
        key = name.replace("-", "_")
        h = orc.run(x, variant=v)
        heads[key] = h
        hr = orc.run(real, variant=v)
        meta["variants"][key] = {"oracle_variant": v, "real_frames_head_sha256": [hashlib.sha256(a.tobytes()).hexdigest() for a in hr],
                                 "golden_detections_py": [[[int(d[1]), int(d[2]), int(d[3]), int(d[6]), int(d[7]), int(d[8]), int(d[9])] for d in orc.decode_py(h[f], f)] for f in range(len(x))]}
        print(key, "golden heads sha", hashlib.sha256(h.tobytes()).hexdigest()[:16])
    np.savez_compressed(os.path.join(HERE, "golden_heads_variants.npz"), **heads)
    json.dump(meta, open(os.path.join(HERE, "golden_variants_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

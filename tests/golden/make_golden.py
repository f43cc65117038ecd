#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle (oracle/yf_oracle.c).

HONEST LABEL: these vectors are self-consistent, TFLite-semantics-by-construction and interpreter-UNVERIFIED
(the reference holds no goldens for this path and TensorFlow cannot be run here; oracle/yf_oracle.h).

Inputs (int8 [56,56,3]):  0..2 numpy default_rng(seed).integers(-128,128), 3 all -128, 4 all 127,
                          5 reference yoloface/small_dataset/img_82.jpg -> PIL RGB -> resize 56x56 (bilinear) -> uint8-128
                            (needs /root/reference; skipped frame is kept from the committed file otherwise)
Files: golden_inputs.bin [6][9408] int8, golden_heads.bin [6][882] int8,
       golden_dump_frame0.bin [196199] int8 (every tflite op output of input 0),
       golden_meta.json (sha256 of every op output of every input, detections in both decode modes)
"""
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from oracle.np_restatement import load_yfm  # noqa: E402


def main():
    frames = [np.random.default_rng(s).integers(-128, 128, (56, 56, 3), dtype=np.int8) for s in range(3)]
    frames.append(np.full((56, 56, 3), -128, np.int8))
    frames.append(np.full((56, 56, 3), 127, np.int8))
    img = "/root/reference/yoloface/small_dataset/img_82.jpg"
    if os.path.exists(img):
        from PIL import Image
        im = Image.open(img).convert("RGB").resize((56, 56), Image.BILINEAR)
        frames.append((np.asarray(im).astype(np.int16) - 128).astype(np.int8))
    else:
        frames.append(np.fromfile(os.path.join(HERE, "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)[5])
    x = np.stack(frames)
    orc = Oracle()
    heads, dump = orc.run(x, dump=True)
    x.tofile(os.path.join(HERE, "golden_inputs.bin"))
    heads.tofile(os.path.join(HERE, "golden_heads.bin"))
    dump[0].tofile(os.path.join(HERE, "golden_dump_frame0.bin"))
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
    meta = {"label": "self-consistent, TFLite-semantics-by-construction, interpreter-unverified", "frames": []}
    names = ["rng0", "rng1", "rng2", "all_-128", "all_127", "img_82_56x56"]
    for f in range(x.shape[0]):
        off, sha = 0, []
        for n in sizes:
            sha.append(hashlib.sha256(dump[f, off:off + n].tobytes()).hexdigest()[:16])
            off += n
        keys = ["frame", "anchor", "row", "col", "q_conf", "conf", "x1", "y1", "x2", "y2"]
        py = [dict(zip(keys, [float(v) if k == "conf" else int(v) for k, v in zip(keys, d)])) for d in orc.decode_py(heads[f], f)]
        fw = [dict(zip(keys, [float(v) if k == "conf" else int(v) for k, v in zip(keys, d)])) for d in orc.decode_c(heads[f], f)]
        meta["frames"].append(dict(name=names[f], op_sha256_16=sha, head_sha256=hashlib.sha256(heads[f].tobytes()).hexdigest(),
                                   detections_py=py, detections_fw=fw))
        print(names[f], "dets py", len(py), "fw", len(fw), py[:2])
    json.dump(meta, open(os.path.join(HERE, "golden_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

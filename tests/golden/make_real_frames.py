#!/usr/bin/env python3
"""Builds real_frames_56.bin: the reference's 27 sample images (yoloface/small_dataset/*.jpg, DATA files) as the
frames the firmware would feed the network: PIL RGB -> resize 56x56 (bilinear) -> uint8 - 128, int8 [27][56][56][3],
sorted by file name; real_frames_56.json lists the names and the sha256 of the oracle's head for each frame.
Container-only (needs /root/reference); the committed .bin/.json travel to the GPU box."""
import glob
import hashlib
import json
import os
import sys
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import Oracle  # noqa: E402

files = sorted(glob.glob("/root/reference/yoloface/small_dataset/*.jpg"))
frames = np.stack([(np.asarray(Image.open(f).convert("RGB").resize((56, 56), Image.BILINEAR)).astype(np.int16) - 128).astype(np.int8)
                   for f in files])
frames.tofile(os.path.join(HERE, "real_frames_56.bin"))
heads = Oracle().run(frames)
json.dump({"label": "inputs = reference data files; heads = this repo's oracle (interpreter-unverified)",
           "files": [os.path.basename(f) for f in files],
           "head_sha256": [hashlib.sha256(h.tobytes()).hexdigest() for h in heads]},
          open(os.path.join(HERE, "real_frames_56.json"), "w"), indent=1)
print(frames.shape, "written")

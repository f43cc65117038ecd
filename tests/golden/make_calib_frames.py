#!/usr/bin/env python3
"""Builds calib_frames_56_cv.bin: the reference's representative dataset as its PTQ script feeds it to the converter
(yoloface/tflite/tflite_quantize.py:43-58): decode, BGR->RGB, cv2.resize(..., (56, 56)) -- OpenCV's INTER_LINEAR, restated
in numpy by stm32h7-yolo_amd/ptq.py::resize_linear_u8 because cv2 is not available here --, uint8 - 128 as int8
[27][56][56][3], sorted by file name.  (real_frames_56.bin, made with PIL's antialiased resize, stays the fixture of the
"what the firmware would see" tests.)  Container-only (needs /root/reference); the committed .bin travels."""
import glob
import importlib
import os
import sys
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
ptq = importlib.import_module("stm32h7-yolo_amd.ptq")
files = sorted(glob.glob("/root/reference/yoloface/small_dataset/*.jpg"))
frames = np.stack([(ptq.resize_linear_u8(np.asarray(Image.open(f).convert("RGB")), 56, 56).astype(np.int16) - 128).astype(np.int8) for f in files])
frames.tofile(os.path.join(HERE, "calib_frames_56_cv.bin"))
print(frames.shape, "written")

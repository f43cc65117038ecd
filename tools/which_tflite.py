#!/usr/bin/env python3
"""Which TFLite arithmetic does YOUR interpreter compute?  Settles the one open parity question of this project wherever TensorFlow can run (it cannot here).

Run the reference's interpreter (yoloface/tflite/tflite_prediction.py:23-41) on the six golden frames and hand the heads to this script:

    import numpy as np, tensorflow as tf
    x = np.fromfile("tests/golden/golden_inputs.bin", np.int8).reshape(-1, 1, 56, 56, 3)
    it = tf.lite.Interpreter(model_path="yoloface/tflite/yoloface_int8.tflite"); it.allocate_tensors()
    i, o = it.get_input_details()[0]["index"], it.get_output_details()[0]["index"]
    heads = []
    for f in x:
        it.set_tensor(i, f); it.invoke(); heads.append(it.get_tensor(o))
    np.concatenate(heads).astype(np.int8).tofile("my_heads.bin")            # 6 x 7 x 7 x 18 int8

    python tools/which_tflite.py my_heads.bin

It compares them with tests/golden/golden_heads.bin (the builtin REFERENCE kernels: this library's default) and with every array of
tests/golden/golden_heads_variants.npz (make_golden_variants.py), and says which rounding to select (yf_network_set_requant_rounding / $YF_REQUANT_ROUNDING) --
or that none matches, with the distance to each, which would mean the restatement is wrong somewhere else.  Host-only, no GPU, no TensorFlow needed to run IT."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
SELECT = {"R": "the default (YF_ROUND_TFLITE_REF / YF_REQUANT_ROUNDING=ref): nothing to change; the oracle is PINNED by your run",
          "U": "YF_ROUND_TIES_UP / YF_REQUANT_ROUNDING=ties_up (dense convs as ruy rounds them)",
          "U_all": "YF_ROUND_TIES_UP_ALL / YF_REQUANT_ROUNDING=ties_up_all",
          "S": "YF_ROUND_SINGLE / YF_REQUANT_ROUNDING=single (ruy's portable path)",
          "X": "no library mode: fp32 requantisation (the XNNPACK delegate) -- build the interpreter with experimental_op_resolver_type=BUILTIN_WITHOUT_DEFAULT_DELEGATES"}


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    mine = np.fromfile(sys.argv[1], np.int8)
    if mine.size != 6 * 882:
        raise SystemExit(f"{sys.argv[1]}: {mine.size} bytes, expected 6 x 882 (the heads of tests/golden/golden_inputs.bin)")
    mine = mine.reshape(6, 7, 7, 18)
    cands = {"R": np.fromfile(os.path.join(G, "golden_heads.bin"), np.int8).reshape(6, 7, 7, 18)}
    cands.update(dict(np.load(os.path.join(G, "golden_heads_variants.npz"))))
    hit = None
    for name, want in cands.items():
        d = mine.astype(int) - want.astype(int)
        print(f"{name:6s} {int(np.count_nonzero(d)):5d} of {d.size} head bytes differ, max |delta| {int(np.abs(d).max())}")
        if not d.any():
            hit = name
    print(f"\nyour interpreter computes variant {hit}: {SELECT[hit]}" if hit else
          "\nNO variant matches bit for bit: the restatement differs from your interpreter somewhere else (or the run used other inputs / another model file)")
    sys.exit(0 if hit else 1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate the committed model fixtures from the reference's DATA files -- DEV TOOL, container only.

Reads (never copies source text from) /root/reference:
  * yoloface/tflite/yoloface_int8.tflite         -> graph, weights, biases, quant params
  * stm32/X-CUBE-AI/App/network.c                -> ST blob byte offsets (network.c:3120-3263) and the
                                                    17 ST LeakyReLU LUTs (known-answer DATA, network.c:2218..2902)
  * stm32/X-CUBE-AI/App/network_data.c           -> only to ASSERT that the blob rebuilt from the .tflite
                                                    is byte-identical to ST's
Writes:
  oracle/model/yoloface_int8.yfm                 generic op-by-op model pack for the CPU oracle
  stm32h7-yolo_amd/csrc/gen/yf_model_gen.h       baked quant tables + blob offsets for the product
  stm32h7-yolo_amd/csrc/gen/yf_weights_blob_gen.c the 11304-byte weight blob in the ST layout
  stm32h7-yolo_amd/csrc/gen/yf_decode_tables_gen.h  sigmoid/exp float32 tables for the box decode
  tests/golden/st_leaky_luts.bin                 17 x 256 int8, ST's LUTs in tflite op order (2,4,7,...)
  tests/golden/decode_tables_f32.bin             2 x 256 float32 (sigmoid, exp)

.yfm layout (little endian):
  header  : 'YFM1', u32 n_tensors, u32 n_ops, u32 input_tensor, u32 output_tensor, u32 data_bytes
  tensor  : i32 shape[4], u32 type(0=i8,1=i32), i32 zero_point, u32 n_scales, u32 scales_off,
            i32 quantized_dimension, u32 data_off(0xFFFFFFFF=none), u32 data_bytes           (44 B)
  op      : u32 opcode, i32 inputs[3], i32 output, i32 padding, i32 stride_w, i32 stride_h,
            i32 filter_w, i32 filter_h, i32 depth_multiplier, i32 axis, u32 alpha_bits           (52 B)
  data    : scales (f32) and constant tensor bytes, each 4-byte aligned
"""
import os
import re
import struct
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from tflite_reader import read_tflite  # noqa: E402

REF = "/root/reference"
TFL = f"{REF}/yoloface/tflite/yoloface_int8.tflite"
NETC = f"{REF}/stm32/X-CUBE-AI/App/network.c"
NETD = f"{REF}/stm32/X-CUBE-AI/App/network_data.c"
PKG = os.path.join(ROOT, "stm32h7-yolo_amd")
OUT_ROOT = ROOT        # --out-root DIR: write everything under DIR (same relative paths) instead of into the tree (tests compare the bytes)

OPCODE = {"ADD": 0, "CONCATENATION": 2, "CONV_2D": 3, "DEPTHWISE_CONV_2D": 4, "MAX_POOL_2D": 17,
          "PAD": 34, "LEAKY_RELU": 98, "QUANTIZE": 114}


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def write_yfm(m, path):
    data = bytearray()

    def put(b):
        while len(data) % 4:
            data.append(0)
        off = len(data)
        data.extend(b)
        return off

    trecs = []
    for t in m["tensors"]:
        shape = (list(t["shape"]) + [1, 1, 1, 1])[:4] if len(t["shape"]) < 4 else list(t["shape"])
        ttype = {"INT8": 0, "INT32": 1}[t["type"]]
        zp = int(t["zero_point"][0]) if len(t["zero_point"]) else 0
        ns = len(t["scale"])
        soff = put(t["scale"].astype("<f4").tobytes()) if ns else 0
        if t["data"] is not None:
            raw = t["data"].tobytes()
            doff, dbytes = put(raw), len(raw)
        else:
            doff, dbytes = 0xFFFFFFFF, 0
        trecs.append(struct.pack("<4iIiIIiII", *shape, ttype, zp, ns, soff, t["quantized_dimension"], doff, dbytes))
    orecs = []
    for op in m["ops"]:
        o = op["options"]
        ins = (op["inputs"] + [-1, -1, -1])[:3]
        orecs.append(struct.pack("<I3ii7iI", OPCODE[op["op"]], *ins, op["outputs"][0],
                                 o.get("padding", 0), o.get("stride_w", 1), o.get("stride_h", 1),
                                 o.get("filter_w", 0), o.get("filter_h", 0), o.get("depth_multiplier", 0),
                                 o.get("axis", 0), f32bits(o.get("alpha", 0.0))))
    while len(data) % 4:
        data.append(0)
    hdr = b"YFM1" + struct.pack("<5I", len(trecs), len(orecs), m["inputs"][0], m["outputs"][0], len(data))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(hdr + b"".join(trecs) + b"".join(orecs) + bytes(data))


def st_offsets():
    """{ 'conv2d_N': (w_off, b_off) } from network_configure_weights (network.c:3120-3263)."""
    src = open(NETC, encoding="latin-1").read()
    offs = {}
    for name, kind, off in re.findall(r"(conv2d_\d+)_(weights|bias)_array\.data = AI_PTR\(weights_map\[0\] \+ (\d+)\)", src):
        offs.setdefault(name, {})[kind] = int(off)
    return {k: (v["weights"], v["bias"]) for k, v in offs.items()}


def st_luts():
    """17 ST LeakyReLU LUTs keyed by conv id (network.c:2218..2902)."""
    src = open(NETC, encoding="latin-1").read()
    out = {}
    for cid, body in re.findall(r"conv2d_(\d+)_nl_params_data\[\] = \{([^}]*)\}", src):
        v = np.array([int(x) for x in body.split(",")], dtype=np.int8)
        assert v.size == 256
        out[int(cid)] = v
    return out


def st_report_identity():
    """The identity fields ai_network_get_report / get_info return (reference network.c:38-49,3271-3361): values of the #defines in the generated
    files network.c / network.h / network_config.h, read as DATA (the library must answer with the same strings and numbers)."""
    def define(path, name, pattern=r'"([^"]*)"'):
        src = open(path, encoding="latin-1").read()
        found = re.findall(rf"#define\s+{name}\s+\(?{pattern}\)?", src)
        assert found, (path, name)
        return found[-1]
    app = os.path.dirname(NETC)
    num = r"(\d+)"
    ident = {"MODEL_NAME": define(f"{app}/network.h", "AI_NETWORK_MODEL_NAME"),
             "ORIGIN_MODEL_NAME": define(f"{app}/network.h", "AI_NETWORK_ORIGIN_MODEL_NAME"),
             "MODEL_SIGNATURE": define(NETC, "AI_NETWORK_MODEL_SIGNATURE"),
             "MODEL_DATETIME": define(NETC, "AI_TOOLS_DATE_TIME"),
             "TOOLS_REVISION_ID": define(NETC, "AI_TOOLS_REVISION_ID")}
    vers = {"TOOLS_VERSION": [int(define(f"{app}/network_config.h", f"AI_TOOLS_VERSION_{p}", num)) for p in ("MAJOR", "MINOR", "MICRO")],
            "TOOLS_API_VERSION": [int(define(f"{app}/network_config.h", f"AI_TOOLS_API_VERSION_{p}", num)) for p in ("MAJOR", "MINOR", "MICRO")],
            "PLATFORM_API_VERSION": [int(define(f"{app}/network_config.h", f"AI_PLATFORM_API_{p}", num)) for p in ("MAJOR", "MINOR", "MICRO")]}
    macc = set(re.findall(r"\.n_macc\s*=\s*(\d+)", open(NETC, encoding="latin-1").read()))
    assert len(macc) == 1
    return ident, vers, int(macc.pop())


def st_blob():
    src = open(NETD, encoding="latin-1").read()
    body = src[src.index("s_network_weights_array_u64"):]
    body = body[body.index("{") + 1: body.index("};")]
    words = [int(x.rstrip("U"), 16) for x in re.findall(r"0x[0-9a-fA-F]+U?", body)]
    return b"".join(struct.pack("<Q", w) for w in words)


def main():
    m = read_tflite(TFL)
    T = m["tensors"]
    os.makedirs(os.path.join(OUT_ROOT, "oracle", "model"), exist_ok=True)
    write_yfm(m, os.path.join(OUT_ROOT, "oracle", "model", "yoloface_int8.yfm"))

    # ---- conv table: tflite op index -> ST c-layer id -----------------------
    convs = [(i, op) for i, op in enumerate(m["ops"]) if op["op"] in ("CONV_2D", "DEPTHWISE_CONV_2D")]
    offs = st_offsets()
    blob = bytearray(11304)
    conv_rows = []
    for i, op in convs:
        cid = i  # ST names the c-layer after the tflite op index (conv2d_1, conv2d_3, ...)
        w_off, b_off = offs[f"conv2d_{cid}"]
        w, b = T[op["inputs"][1]], T[op["inputs"][2]]
        wb, bb = w["data"].tobytes(), b["data"].astype("<i4").tobytes()
        blob[w_off:w_off + len(wb)] = wb
        blob[b_off:b_off + len(bb)] = bb
        conv_rows.append((cid, op, w_off, b_off, w, b))
    ref_blob = st_blob()
    assert len(ref_blob) == 11304
    # ST pads tensors to 4/8-byte boundaries with bytes we cannot know from the tflite; they are never read.
    mism = [k for k in range(11304) if blob[k] != ref_blob[k]]
    covered = np.zeros(11304, bool)
    for cid, op, w_off, b_off, w, b in conv_rows:
        covered[w_off:w_off + w["data"].nbytes] = True
        covered[b_off:b_off + 4 * b["data"].size] = True
    assert not [k for k in mism if covered[k]], "tflite weights differ from ST blob"
    for k in mism:                       # adopt ST's alignment-pad bytes so the whole blob is identical
        blob[k] = ref_blob[k]
    assert bytes(blob) == ref_blob
    print(f"blob rebuilt from tflite == ST blob (11304 B; {len(mism)} alignment-pad bytes taken from ST)")

    gen = os.path.join(OUT_ROOT, "stm32h7-yolo_amd", "csrc", "gen")
    os.makedirs(gen, exist_ok=True)

    # ---- weight blob as a byte array (own formatting; identical bytes) ---------------------------
    with open(os.path.join(gen, "yf_weights_blob_gen.c"), "w") as f:
        f.write("/* GENERATED by tools/gen_model.py from yoloface_int8.tflite -- do not edit.\n"
                " * 11304-byte weight/bias blob laid out at the offsets the reference binds in\n"
                " * network_configure_weights (reference stm32/X-CUBE-AI/App/network.c:3120-3263);\n"
                " * byte-identical to the reference blob (network_data.c:25-380), asserted at generation. */\n"
                "#include <stdint.h>\n"
                "#if defined(__GNUC__)\n__attribute__((aligned(32)))\n#endif\n"
                "const uint8_t yf_weights_blob[11304] = {\n")
        for k in range(0, 11304, 24):
            f.write("  " + ",".join(str(x) for x in blob[k:k + 24]) + ",\n")
        f.write("};\n")

    # ---- quant tables --------------------------------------------------------------------------
    with open(os.path.join(gen, "yf_model_gen.h"), "w") as f:
        f.write("/* GENERATED by tools/gen_model.py from yoloface_int8.tflite -- do not edit.\n"
                " * Per-tensor quantisation (float32 bit patterns) indexed by tflite tensor id; equal to the\n"
                " * reference's AI_INTQ_INFO tables (network.c:663-1341).  Conv rows carry the ST blob offsets\n"
                " * (network.c:3120-3263) and the per-channel filter scales. */\n"
                "#ifndef YF_MODEL_GEN_H\n#define YF_MODEL_GEN_H\n#include <stdint.h>\n\n"
                f"#define YF_N_TENSORS {len(T)}\n#define YF_N_CONVS {len(conv_rows)}\n"
                "#define YF_WEIGHTS_BLOB_BYTES 11304\n\n")
        ident, vers, macc = st_report_identity()
        f.write("/* what ai_network_get_report / ai_network_get_info answer (reference network.c:38-49,3271-3361; network.h:29-30;\n"
                " * network_config.h:25-46): the #define values of the generated reference files */\n")
        for k, v in ident.items():
            f.write(f'#define YF_REPORT_{k} "{v}"\n')
        for k, v in vers.items():
            f.write(f"#define YF_REPORT_{k} {v[0]}, {v[1]}, {v[2]}\n")
        f.write(f"#define YF_REPORT_N_MACC {macc}u\n\n")
        f.write("static const uint32_t yf_tensor_scale_bits[YF_N_TENSORS] = {\n")
        for k in range(0, len(T), 6):
            f.write("  " + ", ".join(f"0x{f32bits(t['scale'][0]) if len(t['scale']) else 0:08x}u" for t in T[k:k + 6]) + ",\n")
        f.write("};\nstatic const int16_t yf_tensor_zero_point[YF_N_TENSORS] = {\n")
        for k in range(0, len(T), 16):
            f.write("  " + ", ".join(str(int(t['zero_point'][0]) if len(t['zero_point']) else 0) for t in T[k:k + 16]) + ",\n")
        f.write("};\n\n")
        for cid, op, w_off, b_off, w, b in conv_rows:
            f.write(f"static const uint32_t yf_conv{cid}_wscale_bits[{len(w['scale'])}] = {{"
                    + ", ".join(f"0x{f32bits(s):08x}u" for s in w["scale"]) + "};\n")
        f.write("\ntypedef struct {\n"
                "  uint8_t  tfl_op;      /* tflite operator index == ST c-layer id (conv2d_<id>) */\n"
                "  uint8_t  depthwise;   /* 1: DEPTHWISE_CONV_2D (weights 1HWC), 0: CONV_2D (weights OHWI) */\n"
                "  uint8_t  kh, kw, stride;\n"
                "  uint8_t  pad_tl;      /* explicit PAD op in front: top/left 1 (stride-2 convs) */\n"
                "  uint8_t  pad_same;    /* SAME padding (3x3 stride-1 depthwise): 1 all round */\n"
                "  uint16_t cin, cout;\n"
                "  uint16_t t_in, t_out; /* tflite tensor ids (t_in is the tensor BEFORE the PAD op) */\n"
                "  uint32_t w_off, b_off;/* byte offsets in the weight blob */\n"
                "  const uint32_t* wscale_bits;\n"
                "} yf_conv_desc;\n\n"
                "static const yf_conv_desc yf_convs[YF_N_CONVS] = {\n")
        for cid, op, w_off, b_off, w, b in conv_rows:
            o = op["options"]
            dw = op["op"] == "DEPTHWISE_CONV_2D"
            t_in = op["inputs"][0]
            pad_tl = 0
            # stride-2 convs are fed by an explicit PAD op (tfl ops 0, 9, 26)
            for pop in m["ops"]:
                if pop["op"] == "PAD" and pop["outputs"][0] == t_in:
                    pads = T[pop["inputs"][1]]["data"].reshape(4, 2).tolist()
                    assert pads == [[0, 0], [1, 0], [1, 0], [0, 0]], pads
                    t_in = pop["inputs"][0]
                    pad_tl = 1
            ws = w["shape"]
            kh, kw = ws[1], ws[2]
            cin = T[t_in]["shape"][3]
            cout = T[op["outputs"][0]]["shape"][3]
            same = 1 if (o["padding"] == 0 and kh == 3) else 0
            assert o["stride_w"] == o["stride_h"]
            f.write(f"  {{{cid}, {int(dw)}, {kh}, {kw}, {o['stride_w']}, {pad_tl}, {same}, {cin}, {cout}, "
                    f"{t_in}, {op['outputs'][0]}, {w_off}, {b_off}, yf_conv{cid}_wscale_bits}},\n")
        f.write("};\n\n#endif /* YF_MODEL_GEN_H */\n")

    # ---- ST LUTs as known-answer data (tflite LEAKY_RELU op order) -------------------------------
    luts = st_luts()
    leaky_ops = [i for i, op in enumerate(m["ops"]) if op["op"] == "LEAKY_RELU"]
    assert len(leaky_ops) == 17 and sorted(luts) == [i - 1 for i in leaky_ops]
    gold = os.path.join(OUT_ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    np.stack([luts[i - 1] for i in leaky_ops]).tofile(os.path.join(gold, "st_leaky_luts.bin"))

    # ---- decode tables: numpy float32, the arithmetic of tflite_prediction.py:42,53-55 ------------
    q = np.arange(-128, 128).astype(np.float32)
    x = (q + 15) * 0.14218327403068542          # float32 array * python float -> float32
    assert x.dtype == np.float32
    sig = 1 / (1 + np.exp(-x))
    ex = np.exp(x)
    assert sig.dtype == np.float32 and ex.dtype == np.float32
    np.stack([sig, ex]).astype("<f4").tofile(os.path.join(gold, "decode_tables_f32.bin"))
    with open(os.path.join(gen, "yf_decode_tables_gen.h"), "w") as f:
        f.write("/* GENERATED by tools/gen_model.py (numpy %s float32) -- do not edit.\n"
                " * sigmoid((q+15)*0.14218327403068542f) and exp(...) for q=-128..127, the float32 arithmetic of the\n"
                " * reference decode (yoloface/tflite/tflite_prediction.py:42,53-55).  numpy's float32 exp is not\n"
                " * correctly rounded, so these TABLES (not a formula) are the contract (SURVEY.md 8(c).5). */\n"
                "#ifndef YF_DECODE_TABLES_GEN_H\n#define YF_DECODE_TABLES_GEN_H\n#include <stdint.h>\n" % np.__version__)
        for name, arr in (("yf_sigmoid_bits", sig), ("yf_exp_bits", ex)):
            f.write(f"static const uint32_t {name}[256] = {{\n")
            for k in range(0, 256, 8):
                f.write("  " + ", ".join(f"0x{f32bits(v):08x}u" for v in arr[k:k + 8]) + ",\n")
            f.write("};\n")
        f.write("#endif\n")
    print("wrote model pack, generated headers, golden LUTs and decode tables")


if __name__ == "__main__":
    if "--out-root" in sys.argv:
        OUT_ROOT = os.path.abspath(sys.argv[sys.argv.index("--out-root") + 1])
    main()

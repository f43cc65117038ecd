"""Minimal ONNX protobuf reader (struct only) -- DEV TOOL, container only.

Parses /root/reference/yoloface/pytorch/yoloface-50k.onnx (ir 6, opset 11, 50 nodes, 48 fp32 initializers, BN folded;
SURVEY.md Appendix C lists the message fields relied on).  onnx/onnxruntime are not installable here.
"""
import struct
import numpy as np


def _varint(b, o):
    r, s = 0, 0
    while True:
        c = b[o]; o += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, o
        s += 7


def _fields(b):
    """yield (field_no, wire_type, value) for one message; length-delimited values are returned as bytes."""
    o, n = 0, len(b)
    while o < n:
        key, o = _varint(b, o)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, o = _varint(b, o)
        elif wt == 1:
            v = b[o:o + 8]; o += 8
        elif wt == 2:
            ln, o = _varint(b, o)
            v = b[o:o + ln]; o += ln
        elif wt == 5:
            v = b[o:o + 4]; o += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield fno, wt, v


def _tensor(b):
    dims, dtype, name, raw, floats = [], 0, "", None, []
    for f, wt, v in _fields(b):
        if f == 1:
            if wt == 0:
                dims.append(v)
            else:                                   # packed
                o = 0
                while o < len(v):
                    d, o = _varint(v, o); dims.append(d)
        elif f == 2: dtype = v
        elif f == 8: name = v.decode()
        elif f == 9: raw = bytes(v)
        elif f == 4:
            floats.append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(v, "<f4", 1))
    if dtype == 1:
        arr = np.frombuffer(raw, "<f4") if raw is not None else np.concatenate(floats)
    elif dtype == 7:
        arr = np.frombuffer(raw, "<i8") if raw is not None else np.zeros(0, np.int64)
    else:
        arr = np.zeros(0)
    return name, arr.reshape(dims).copy() if dims else arr.copy()


def _attr(b):
    name, val, ints = "", None, []
    for f, wt, v in _fields(b):
        if f == 1: name = v.decode()
        elif f == 2: val = struct.unpack("<f", v)[0]
        elif f == 3: val = v if v < (1 << 63) else v - (1 << 64)
        elif f == 8:
            if wt == 0:
                ints.append(v)
            else:
                o = 0
                while o < len(v):
                    d, o = _varint(v, o); ints.append(d)
    return name, (ints if ints else val)


def read_onnx(path):
    b = open(path, "rb").read()
    graph = None
    for f, wt, v in _fields(b):
        if f == 7:
            graph = v
    nodes, inits = [], {}
    for f, wt, v in _fields(graph):
        if f == 1:
            ins, outs, op, attrs = [], [], "", {}
            for f2, wt2, v2 in _fields(v):
                if f2 == 1: ins.append(v2.decode())
                elif f2 == 2: outs.append(v2.decode())
                elif f2 == 4: op = v2.decode()
                elif f2 == 5:
                    k, a = _attr(v2); attrs[k] = a
            nodes.append(dict(op=op, inputs=ins, outputs=outs, attrs=attrs))
        elif f == 5:
            name, arr = _tensor(v)
            inits[name] = arr
    return dict(nodes=nodes, initializers=inits)


if __name__ == "__main__":
    m = read_onnx("/root/reference/yoloface/pytorch/yoloface-50k.onnx")
    print(len(m["nodes"]), "nodes", len(m["initializers"]), "initializers")
    for n in m["nodes"]:
        w = [(i, m["initializers"][i].shape) for i in n["inputs"] if i in m["initializers"]]
        print(n["op"], n["inputs"][:1], "->", n["outputs"], n["attrs"], w)

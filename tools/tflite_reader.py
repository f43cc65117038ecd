"""Minimal TFLite flatbuffer reader (numpy + struct only) -- DEV TOOL, container only.

Parses `/root/reference/yoloface/tflite/yoloface_int8.tflite` (schema v3) into plain Python
dicts.  TensorFlow / flatbuffers are not installable here, so the handful of schema tables
this model uses are decoded by hand.  Field slot numbers follow the public TFLite schema
(tensorflow/lite/schema/schema.fbs, TF 2.10); SURVEY.md Appendix C lists the ones relied on.

Nothing in the product or in the GPU-side tests imports this module: it only feeds
`tools/gen_model.py`, which writes the committed data fixtures.
"""
import struct
import numpy as np

# builtin operator codes used by the model (schema.fbs BuiltinOperator)
BUILTIN = {0: "ADD", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 17: "MAX_POOL_2D",
           34: "PAD", 98: "LEAKY_RELU", 114: "QUANTIZE"}
TENSOR_TYPE = {0: "FLOAT32", 2: "INT32", 3: "UINT8", 4: "INT64", 9: "INT8"}
NP_TYPE = {"FLOAT32": np.float32, "INT32": np.int32, "UINT8": np.uint8, "INT64": np.int64, "INT8": np.int8}


class _FB:
    """Flatbuffer mechanics: tables, vtables, vectors, strings."""

    def __init__(self, buf):
        self.b = buf

    def u8(self, o): return self.b[o]
    def i8(self, o): return struct.unpack_from("<b", self.b, o)[0]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def i32(self, o): return struct.unpack_from("<i", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def i64(self, o): return struct.unpack_from("<q", self.b, o)[0]
    def f32(self, o): return struct.unpack_from("<f", self.b, o)[0]

    def root(self):
        return self.u32(0)

    def field(self, table, slot):
        """Absolute offset of field `slot` of `table`, or None when absent (default)."""
        vt = table - self.i32(table)
        vt_len = self.u16(vt)
        pos = 4 + 2 * slot
        if pos >= vt_len:
            return None
        off = self.u16(vt + pos)
        return table + off if off else None

    def indirect(self, o):
        return o + self.u32(o)

    def vec(self, table, slot):
        """(start, length) of the vector in field `slot` or (None, 0)."""
        f = self.field(table, slot)
        if f is None:
            return None, 0
        v = self.indirect(f)
        return v + 4, self.u32(v)

    def table_vec(self, table, slot):
        s, n = self.vec(table, slot)
        return [self.indirect(s + 4 * i) for i in range(n)]

    def np_vec(self, table, slot, dtype):
        s, n = self.vec(table, slot)
        if s is None:
            return np.zeros(0, dtype)
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=s).copy()

    def string(self, table, slot):
        s, n = self.vec(table, slot)
        return bytes(self.b[s:s + n]).decode("utf-8") if s is not None else ""

    def scalar(self, table, slot, kind, default=0):
        f = self.field(table, slot)
        if f is None:
            return default
        return getattr(self, kind)(f)


def read_tflite(path):
    buf = open(path, "rb").read()
    fb = _FB(buf)
    model = fb.root()
    version = fb.scalar(model, 0, "u32")
    opcodes = []
    for t in fb.table_vec(model, 1):
        dep = fb.scalar(t, 0, "i8")
        new = fb.scalar(t, 3, "i32")
        opcodes.append(max(dep, new))
    buffers = []
    for t in fb.table_vec(model, 4):
        s, n = fb.vec(t, 0)
        buffers.append(bytes(buf[s:s + n]) if s is not None else b"")
    subgraphs = fb.table_vec(model, 2)
    assert len(subgraphs) == 1
    sg = subgraphs[0]
    tensors = []
    for t in fb.table_vec(sg, 0):
        shape = fb.np_vec(t, 0, np.int32).tolist()
        ttype = TENSOR_TYPE[fb.scalar(t, 1, "u8")]
        bidx = fb.scalar(t, 2, "u32")
        name = fb.string(t, 3)
        q = fb.field(t, 4)
        scale = np.zeros(0, np.float32)
        zp = np.zeros(0, np.int64)
        qdim = 0
        if q is not None:
            qt = fb.indirect(q)
            scale = fb.np_vec(qt, 2, np.float32)
            zp = fb.np_vec(qt, 3, np.int64)
            qdim = fb.scalar(qt, 6, "i32")
        data = None
        if buffers[bidx]:
            data = np.frombuffer(buffers[bidx], dtype=NP_TYPE[ttype]).reshape(shape).copy()
        tensors.append(dict(name=name, shape=shape, type=ttype, buffer=bidx, scale=scale, zero_point=zp,
                            quantized_dimension=qdim, data=data))
    ops = []
    for t in fb.table_vec(sg, 3):
        code = opcodes[fb.scalar(t, 0, "u32")]
        name = BUILTIN[code]
        ins = fb.np_vec(t, 1, np.int32).tolist()
        outs = fb.np_vec(t, 2, np.int32).tolist()
        opt = fb.field(t, 4)
        o = {}
        if opt is not None:
            ot = fb.indirect(opt)
            if name == "CONV_2D":
                o = dict(padding=fb.scalar(ot, 0, "i8"), stride_w=fb.scalar(ot, 1, "i32"),
                         stride_h=fb.scalar(ot, 2, "i32"), fused_act=fb.scalar(ot, 3, "i8"),
                         dil_w=fb.scalar(ot, 4, "i32", 1), dil_h=fb.scalar(ot, 5, "i32", 1))
            elif name == "DEPTHWISE_CONV_2D":
                o = dict(padding=fb.scalar(ot, 0, "i8"), stride_w=fb.scalar(ot, 1, "i32"),
                         stride_h=fb.scalar(ot, 2, "i32"), depth_multiplier=fb.scalar(ot, 3, "i32"),
                         fused_act=fb.scalar(ot, 4, "i8"), dil_w=fb.scalar(ot, 5, "i32", 1),
                         dil_h=fb.scalar(ot, 6, "i32", 1))
            elif name == "MAX_POOL_2D":
                o = dict(padding=fb.scalar(ot, 0, "i8"), stride_w=fb.scalar(ot, 1, "i32"),
                         stride_h=fb.scalar(ot, 2, "i32"), filter_w=fb.scalar(ot, 3, "i32"),
                         filter_h=fb.scalar(ot, 4, "i32"), fused_act=fb.scalar(ot, 5, "i8"))
            elif name == "LEAKY_RELU":
                o = dict(alpha=fb.scalar(ot, 0, "f32"))
            elif name == "CONCATENATION":
                o = dict(axis=fb.scalar(ot, 0, "i32"), fused_act=fb.scalar(ot, 1, "i8"))
            elif name == "ADD":
                o = dict(fused_act=fb.scalar(ot, 0, "i8"))
        ops.append(dict(op=name, inputs=ins, outputs=outs, options=o))
    inputs = fb.np_vec(sg, 1, np.int32).tolist()
    outputs = fb.np_vec(sg, 2, np.int32).tolist()
    return dict(version=version, tensors=tensors, ops=ops, inputs=inputs, outputs=outputs,
                description=fb.string(model, 3))


if __name__ == "__main__":
    import sys
    m = read_tflite(sys.argv[1] if len(sys.argv) > 1 else
                    "/root/reference/yoloface/tflite/yoloface_int8.tflite")
    print("version", m["version"], "tensors", len(m["tensors"]), "ops", len(m["ops"]),
          "in", m["inputs"], "out", m["outputs"])
    for i, op in enumerate(m["ops"]):
        t = m["tensors"]
        outs = [(o, t[o]["shape"], float(t[o]["scale"][0]) if len(t[o]["scale"]) else None,
                 int(t[o]["zero_point"][0]) if len(t[o]["zero_point"]) else None) for o in op["outputs"]]
        print(i, op["op"], op["inputs"], outs, op["options"])

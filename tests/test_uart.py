"""Wire format of the detections: the firmware's UART text, byte for byte, from the library's C formatter
(yf_network_format_uart) and its Python mirror, read back with the reference monitor's three regular expressions; and the
firmware decode's float -> int conversions at the edge of int32.  CPU only (the formatter needs no GPU)."""
import importlib
import re

import numpy as np

# The monitor's patterns, restated from the reference (上位机/IAP/main.py:325, 333-334, 362): a parser, not arithmetic.
RE_FRAME = r'=== Frame (\d+) ==='
RE_FACE = r'\[Face\s+(\d+)\]\s+BBox:\s*\[(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\],\s*Conf:\s*([\d\.]+)'
RE_TOTAL = r'Total faces detected:\s*(\d+)'


def parse_frame_data(data_lines):
    """上位机/IAP/main.py:317-369 (parse_frame_data) without the GUI object."""
    faces, frame_num, face_count = [], 0, 0
    for line in data_lines:
        m = re.search(RE_FRAME, line)
        if m:
            frame_num = int(m.group(1))
        m = re.search(RE_FACE, line)
        if m:
            faces.append(dict(id=int(m.group(1)), x1=int(m.group(2)), y1=int(m.group(3)), x2=int(m.group(4)), y2=int(m.group(5)),
                              confidence=float(m.group(6))))
        m = re.search(RE_TOTAL, line, re.IGNORECASE)
        if m:
            face_count = int(m.group(1))
    return frame_num, faces, face_count


def _records(yf, rows):
    d = np.zeros(len(rows), yf.DET_DTYPE)
    for i, (x1, y1, x2, y2, conf) in enumerate(rows):
        d[i]["x1"], d[i]["y1"], d[i]["x2"], d[i]["y2"], d[i]["conf"] = x1, y1, x2, y2, conf
    return d


def test_uart_text_is_byte_exact_and_parses_with_the_monitor_regexes(yf):
    binding = importlib.import_module("stm32h7-yolo_amd.binding")
    ip = importlib.import_module("stm32h7-yolo_amd.interpreter")
    rows = [(12, 30, 64, 98, 0.8175744), (0, 0, 110, 110, 0.7005672), (46, 2, 70, 40, 0.9925)]
    dets = _records(yf, rows)
    text = binding.format_uart(7, dets)
    dashes = b"-" * 40
    assert text == (b"=== Frame 7 ===\r\n" + dashes + b"\r\n"                       # stm32/User/main.c:46
                    b"[Face 1] BBox: [12, 30, 64, 98], Conf: 0.82\r\n"              # yoloface.c:148
                    b"[Face 2] BBox: [0, 0, 110, 110], Conf: 0.70\r\n"
                    b"[Face 3] BBox: [46, 2, 70, 40], Conf: 0.99\r\n"
                    + dashes + b"\r\n[INFO] Total faces detected: 3\r\n")            # main.c:53
    assert ip.format_uart(7, dets).encode() == text                                  # the Python mirror
    frame, faces, total = parse_frame_data(text.decode().split("\r\n"))
    assert frame == 7 and total == 3 and [f["id"] for f in faces] == [1, 2, 3]
    for f, (x1, y1, x2, y2, conf) in zip(faces, rows):
        assert (f["x1"], f["y1"], f["x2"], f["y2"]) == (x1, y1, x2, y2)
        assert f["confidence"] == float("%.2f" % np.float32(conf))                   # conf to 2 decimals
    # no faces; a capacity smaller than the count (lines for the stored records, the total line carries the count)
    empty = binding.format_uart(123456, dets[:0])
    assert empty == b"=== Frame 123456 ===\r\n" + dashes + b"\r\n" + dashes + b"\r\n[INFO] Total faces detected: 0\r\n"
    assert parse_frame_data(empty.decode().split("\r\n")) == (123456, [], 0)
    capped = binding.format_uart(1, dets[:2], count=3)
    assert capped.count(b"[Face ") == 2 and capped.endswith(b"Total faces detected: 3\r\n")
    assert ip.format_uart(1, dets[:2], count=3).encode() == capped


def test_formatter_truncates_safely(yf):
    import ctypes
    binding = importlib.import_module("stm32h7-yolo_amd.binding")
    lib = binding.load()
    dets = _records(yf, [(1, 2, 3, 4, 0.9)] * 5)
    need = lib.yf_network_format_uart(9, dets.ctypes.data, 5, 5, None, 0)
    full = binding.format_uart(9, dets)
    assert need == len(full)
    for room in (1, 2, 17, 60, need, need + 1):
        buf = ctypes.create_string_buffer(b"\xAA" * (room + 8), room + 8)
        assert lib.yf_network_format_uart(9, dets.ctypes.data, 5, 5, buf, room) == need
        stored = buf.raw[:room].split(b"\0")[0]
        assert stored == full[:room - 1] and buf.raw[room:] == b"\xAA" * 8            # NUL-terminated, nothing past the buffer


def test_firmware_decode_at_the_edge_of_int32(oracle):
    """yoloface.c:135-138 assign float box edges to int.  On the Cortex-M7 that is VCVT.S32.F32 (saturating); the same
    file compiled for an x86-64 host converts with cvttss2si (out of range -> INT32_MIN).  With w = h logits of 127,
    exp((127+15)*0.1421...) * anchor is ~5e9: edges beyond +-2^31."""
    head = np.full((7, 7, 18), -128, np.int8)
    head[2, 3, 0:6] = [0, 0, 127, 127, 127, -128]            # anchor 0 of cell (row 2, col 3): conf fires, w and h explode
    fw = oracle.decode_c(head)                               # the firmware (MCU)
    host = oracle.decode_c(head, host_x86=True)              # yoloface.c built for a PC
    assert len(fw) == len(host) == 1 and fw[0][1:4] == host[0][1:4] == (0, 2, 3)
    # MCU: y1 = x2 = INT32_MAX, x1 = y2 = INT32_MIN; clamps: x1 < 0 -> 0, x2 > 55 -> 55; printed doubled with wrap-around:
    #      x1*2 = 0, y1*2 = INT32_MAX*2 = -2, x2*2 = 110, y2*2 = INT32_MIN*2 = 0
    assert fw[0][6:10] == (0, -2, 110, 0)
    # x86-64: all four conversions give INT32_MIN: x1, y1 < 0 -> 0; x2, y2 stay INT32_MIN (not > 55) -> doubled = 0
    assert host[0][6:10] == (0, 0, 0, 0)
    # in range the two conventions agree
    head[2, 3, 2:4] = [-20, -10]
    assert oracle.decode_c(head) == oracle.decode_c(head, host_x86=True)

#!/usr/bin/env python3
"""Per-stage instruction counts of the fused fp16 kernel.  Needs a library built with -DYF16_STAGEPMC
(make -C stm32h7-yolo_amd/csrc OUT=../lib_f16stage EXTRA_FP16FLAGS=-DYF16_STAGEPMC): every frame is abandoned behind barrier number
YF16_STOP_STAGE, one launch per value; run under
   rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -o p -- python3 tools/fp16_stage_pmc.py run
then   python3 tools/fp16_stage_pmc.py report <dir> [<dir2> ...]   prints the differences between consecutive launches per frame
(counters of several directories = several --pmc passes are joined).  DEV TOOL."""
import csv, glob, importlib, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 4096
# barrier-delimited front stages in launch order (stop = 1 ... 12: every frame is abandoned behind that barrier, no tail phase), then the full kernel
# (stop = 0): its difference to stop = 12 is the tail phase (conv2d_29 .. conv2d_53, one frame per wave) + the per-batch arena clear / weight fetch
STAGES = [("input staging + halo fills", 0), ("conv2d_1", 6272), ("conv2d_3 (dw)", 6272),
          ("conv2d_5 -> conv2d_6", 3136 + 14112), ("pool_8 h | conv2d_10 (dw)", 3528), ("pool_8 v | conv2d_12", 1176), ("conv2d_13", 7056),
          ("conv2d_15 (dw)", 7056), ("conv2d_17+add", 1176), ("conv2d_19", 3528), ("conv2d_23", 4704),
          ("pool_25 | conv2d_27 (dw) -> park", 1176),
          ("tail phase: conv2d_29 .. conv2d_53", 392 + 1960 * 3 + 392 * 2 + 1960 * 2 + 1176 + 1960 + 1568 + 882)]


def run():
    import numpy as np, torch
    yf = importlib.import_module("stm32h7-yolo_amd")
    x = (np.random.default_rng(3).integers(0, 256, (N, 56, 56, 3)).astype(np.float32) / 255.0).astype(np.float16)
    net = yf.Network().init(); net.fp16_init()
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((N, 7, 7, 18), dtype=torch.float32, device="cuda")
    for k in list(range(1, len(STAGES))) + [0]:      # stops 1 .. 12, then everything
        os.environ["YF16_STOP_STAGE"] = str(k)
        net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), N)
    torch.cuda.synchronize()


def report(roots):
    per = defaultdict(dict)
    for root in roots:
        rows = []
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        acc = defaultdict(lambda: defaultdict(float))
        for r in rows:
            if "f16_fused" in r["Kernel_Name"]: acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        ids = sorted(acc)
        assert len(ids) == len(STAGES), (root, len(ids))
        for i, d in enumerate(ids): per[i].update(acc[d])
    ctrs = sorted(per[0])
    floor = lambda i: STAGES[i][1] / 64.0 * 2.5 if STAGES[i][1] else None      # v_mul + v_max per output, one v_cvt_pk per two   # noqa: E731
    has_valu = "SQ_INSTS_VALU" in ctrs
    print(f"{'stage':32s}" + "".join(f"{c[3:] if c.startswith('SQ_') else c:>18s}" for c in ctrs)
          + (f"{'conv outputs':>14s}{'VALU floor':>12s}{'VALU/floor':>12s}" if has_valu else "") + "   (per frame)")
    prev = {c: 0.0 for c in ctrs}; tot_floor = 0.0
    for i in range(len(STAGES)):
        cur = per[i]
        line = f"{STAGES[i][0]:32s}" + "".join(f"{(cur[c] - prev[c]) / N:18.1f}" for c in ctrs)
        if has_valu:
            fl = floor(i); v = (cur["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]) / N; tot_floor += fl or 0.0
            line += f"{STAGES[i][1]:14d}{fl:12.0f}{v / fl:12.2f}" if fl else f"{'-':>14s}{'-':>12s}{'-':>12s}"
        print(line); prev = cur
    print(f"{'TOTAL':32s}" + "".join(f"{prev[c] / N:18.1f}" for c in ctrs)
          + (f"{sum(o for _, o in STAGES):14d}{tot_floor:12.0f}{prev['SQ_INSTS_VALU'] / N / tot_floor:12.2f}" if has_valu else ""))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2:])

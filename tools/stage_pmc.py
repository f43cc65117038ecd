#!/usr/bin/env python3
"""Per-stage instruction counts of the fused kernel: one launch per stop_stage (debug build), meant to run under
   rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d gpurun_out/prof/stage_pmc -o p -- python3 tools/stage_pmc.py run
and afterwards   python3 tools/stage_pmc.py report gpurun_out/prof/stage_pmc   prints the per-stage differences."""
import csv, glob, importlib, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.stage_profile_names import NAMES  # noqa: E402

N = 4096
# requantised outputs per frame of each barrier-delimited stage (tools/stage_profile_names.py order); x2 where a residual add
# requantises the sum again; 0 = no convolution in the stage
OUTPUTS = [0, 6272, 6272, 3136, 14112, 0, 3528, 1176, 7056, 7056, 2 * 1176, 3528, 4704, 1176, 392, 1960, 1960, 2 * 392, 1960, 1960,
           2 * 392, 1176, 1960, 1960, 1568, 882]


def run():
    import numpy as np, torch
    yf = importlib.import_module("stm32h7-yolo_amd")
    x = np.random.default_rng(1).integers(-128, 128, (N, 56, 56, 3), dtype=np.int8)
    net = yf.Network().init()
    d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((N, 7, 7, 18), dtype=torch.int8, device="cuda")
    for k in list(range(1, 26)) + [0]:
        net.time_stages(d_in.data_ptr(), d_out.data_ptr(), N, 1, k)
    torch.cuda.synchronize()


def report(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = defaultdict(dict)
    for r in rows:
        if "yoloface56_fused" not in r["Kernel_Name"]: continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    assert len(ids) == 26, len(ids)
    ctrs = sorted(per[ids[0]])
    # VALU floor of a convolution stage = requantised outputs / 64 lanes x (4 requantisation ops + 0.75 pack ops); the
    # residual-add stages requantise twice.  Pools and the input staging have no such floor.
    floor = lambda i: OUTPUTS[i] / 64.0 * 4.75 if OUTPUTS[i] else None          # noqa: E731
    has_valu = "SQ_INSTS_VALU" in ctrs
    print(f"{'stage':30s}" + "".join(f"{c[3:] if c.startswith('SQ_') else c:>18s}" for c in ctrs)
          + (f"{'conv outputs':>16s}{'VALU floor':>12s}{'VALU/floor':>12s}" if has_valu else "") + "   (per frame)")
    prev = {c: 0.0 for c in ctrs}
    tot_floor = 0.0
    for i, d in enumerate(ids):
        cur = per[d]
        line = f"{NAMES[i]:30s}" + "".join(f"{(cur[c]-prev[c])/N:18.1f}" for c in ctrs)
        if has_valu:
            fl = floor(i)
            v = (cur["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]) / N
            tot_floor += fl or 0.0
            line += f"{OUTPUTS[i]:16d}{fl:12.0f}{v / fl:12.2f}" if fl else f"{'-':>16s}{'-':>12s}{'-':>12s}"
        print(line)
        prev = cur
    print(f"{'TOTAL':30s}" + "".join(f"{prev[c]/N:18.1f}" for c in ctrs)
          + (f"{sum(OUTPUTS):16d}{tot_floor:12.0f}{prev['SQ_INSTS_VALU'] / N / tot_floor:12.2f}" if has_valu else ""))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])

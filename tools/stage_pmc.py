#!/usr/bin/env python3
"""Per-stage instruction counts of the fused kernel: one launch per stop_stage (debug build), meant to run under
   rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d gpurun_out/prof/stage_pmc -o p -- python3 tools/stage_pmc.py run
and afterwards   python3 tools/stage_pmc.py report gpurun_out/prof/stage_pmc   prints the per-stage differences."""
import csv, glob, importlib, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.stage_profile_names import NAMES  # noqa: E402

N = 4096


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
    print(f"{'stage':30s}" + "".join(f"{c[3:] if c.startswith('SQ_') else c:>18s}" for c in ctrs) + "   (per frame)")
    prev = {c: 0.0 for c in ctrs}
    for i, d in enumerate(ids):
        cur = per[d]
        print(f"{NAMES[i]:30s}" + "".join(f"{(cur[c]-prev[c])/N:18.1f}" for c in ctrs))
        prev = cur
    print(f"{'TOTAL':30s}" + "".join(f"{prev[c]/N:18.1f}" for c in ctrs))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])

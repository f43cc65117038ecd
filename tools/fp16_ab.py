#!/usr/bin/env python3
"""Interleaved timing of the fused fp16 kernel of several library builds (one process per library and round) and a comparison of their
   logits with the first library's.  usage: fp16_ab.py <libdir> <libdir> [...] [rounds]   (directories under stm32h7-yolo_amd/).  DEV TOOL."""
import subprocess, sys, os, re, statistics
import numpy as np
args = sys.argv[1:]
rounds = int(args.pop()) if args and args[-1].isdigit() else 3
libs = args
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
res = {l: [] for l in libs}
notes = {l: "" for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, YF_LIB_PATH=os.path.join(root, "stm32h7-yolo_amd", l, "libyf_network.so"))
        dump = os.path.join(root, "gpurun_out", f"f16ab_{l}.npy")
        cmd = [sys.executable, os.path.join(root, "tools", "fp16_bench.py")] + (["--dump", dump] if r == 0 else [])
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        m = re.search(r"([\d.]+) ms per", p.stdout)
        if not m:
            notes[l] = "FAILED: " + (p.stderr.strip().splitlines() or ["?"])[-1]
            continue
        res[l].append(float(m.group(1)) * 1e3)
        if r == 0:
            notes[l] = p.stdout.strip().splitlines()[-1]
base = None
for l in libs:
    if not res[l]:
        print(f"{l:16s} {notes[l]}"); continue
    out = np.load(os.path.join(root, "gpurun_out", f"f16ab_{l}.npy"))
    if base is None: base = out
    d = np.abs(out - base)
    med = statistics.median(res[l])
    print(f"{l:16s} median {med:7.2f} us ({4096 / med:.2f} M frames/s)  vs first: identical {bool(np.array_equal(out, base))} max abs diff {d.max():.5f}   {notes[l]}   {[round(x, 1) for x in res[l]]}", flush=True)

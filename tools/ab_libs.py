#!/usr/bin/env python3
"""Interleaved timing of the SHIPPED kernel of several library builds (one process per library and round; network + fused decode, HBM-resident
   rotating inputs).  usage: ab_libs.py <libdir> <libdir> [...] [rounds]      (directories under stm32h7-yolo_amd/).  DEV TOOL."""
import subprocess, sys, os, re, statistics
args = sys.argv[1:]
rounds = int(args.pop()) if args and args[-1].isdigit() else 4
libs = args
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, YF_LIB_PATH=os.path.join(root, "stm32h7-yolo_amd", l, "libyf_network.so"))
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "probe", "decode_cost.py")], env=env, capture_output=True, text=True, timeout=300).stdout
        m = re.search(r"network only ([\d.]+) us per launch, network \+ fused decode ([\d.]+) us", out)
        res[l].append((float(m.group(1)), float(m.group(2))))
for l in libs:
    print(f"{l:16s} network only median {statistics.median(x[0] for x in res[l]):7.2f} us   with decode {statistics.median(x[1] for x in res[l]):7.2f} us   ({res[l]})")

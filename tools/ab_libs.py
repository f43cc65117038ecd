#!/usr/bin/env python3
"""Interleaved timing of the SHIPPED kernel of several library builds (one process per library and round; network + fused decode, HBM-resident
   rotating inputs).  usage: ab_libs.py <libdir>[@rounding] <libdir>[@rounding] [...] [rounds]      (directories under stm32h7-yolo_amd/; @ties_up etc. sets
   YF_REQUANT_ROUNDING for that library's runs).  DEV TOOL."""
import subprocess, sys, os, re, statistics
args = sys.argv[1:]
rounds = int(args.pop()) if args and args[-1].isdigit() else 4
libs = args
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in libs}
notes = {l: "" for l in libs}
for r in range(rounds):
    for l in libs:
        ldir, _, rounding = l.partition("@")         # "lib@ties_up": that library with YF_REQUANT_ROUNDING=ties_up (same kernels, other constants)
        path = os.path.join(root, "stm32h7-yolo_amd", ldir, "libyf_network.so")
        if not os.path.exists(path):
            notes[l] = f"FAILED: {path} does not exist"
            continue
        env = dict(os.environ, YF_LIB_PATH=path, **({"YF_REQUANT_ROUNDING": rounding} if rounding else {}))
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "probe", "decode_cost.py")], env=env, capture_output=True, text=True, timeout=300)
        m = re.search(r"network only ([\d.]+) us per launch, network \+ fused decode ([\d.]+) us", p.stdout)
        if p.returncode != 0 or not m:       # a missing library, a child that died on the GPU, a changed output line: say which, with the child's last words
            notes[l] = f"FAILED: exit code {p.returncode}: " + ((p.stderr.strip() or p.stdout.strip()).splitlines() or ["no output"])[-1]
            continue
        res[l].append((float(m.group(1)), float(m.group(2))))
for l in libs:
    if not res[l]:
        print(f"{l:16s} {notes[l]}"); continue
    if notes[l]:
        print(f"{l:16s} (some rounds: {notes[l]})")
    print(f"{l:16s} network only median {statistics.median(x[0] for x in res[l]):7.2f} us   with decode {statistics.median(x[1] for x in res[l]):7.2f} us   ({res[l]})")

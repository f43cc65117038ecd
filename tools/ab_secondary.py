#!/usr/bin/env python3
"""Interleaved timing of one side configuration of the bench line for several library builds, IN THE BENCH'S OWN REGIME: every sample is a fresh
`python bench.py --only-secondary <section>` (same clock settle, same timed region as the `secondary` entry of the full line) with YF_LIB_PATH pointing at
the build.  usage: ab_secondary.py <fp16_56x56|int8_160x160|camera_rgb565_112x112> <libdir> <libdir> [...] [rounds]   (directories under stm32h7-yolo_amd/).
A build that fails says why (exit code and the child's last line).  DEV TOOL."""
import json, os, statistics, subprocess, sys
args = sys.argv[1:]
section = args.pop(0)
rounds = int(args.pop()) if args and args[-1].isdigit() else 4
libs = args
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res, notes = {l: [] for l in libs}, {l: "" for l in libs}
for r in range(rounds):
    for l in libs:
        path = os.path.join(root, "stm32h7-yolo_amd", l, "libyf_network.so")
        if not os.path.exists(path):
            notes[l] = f"FAILED: {path} does not exist"
            continue
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--only-secondary", section], env=dict(os.environ, YF_LIB_PATH=path), capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            notes[l] = f"FAILED: exit code {p.returncode}: " + ((p.stderr.strip() or p.stdout.strip()).splitlines() or ["no output"])[-1]
            continue
        res[l].append(json.loads(lines[-1])["secondary"][section]["ms_per_step"] * 1e3)
base = None
for l in libs:
    if not res[l]:
        print(f"{l:16s} {notes[l]}"); continue
    med = statistics.median(res[l])
    base = base or med
    print(f"{l:16s} median {med:8.2f} us per step ({100 * (med / base - 1):+5.1f} % vs {libs[0]})   {[round(x, 1) for x in res[l]]}   {notes[l]}", flush=True)

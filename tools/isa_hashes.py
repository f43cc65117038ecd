#!/usr/bin/env python3
"""Instruction-stream identity of the device code inside a built library.  DEV TOOL (container or GPU box; needs /opt/rocm/lib/llvm/bin).

Unbundles the gfx950 code objects (yf_engine.o, yf_fp16.o) out of libyf_network.so's fat binary, disassembles them (no raw bytes, no comments) and prints one line per
kernel: sha256[:16] of its instruction text, its instruction count, its mangled name.  The int8 kernels are FROZEN since round 4;
`profiles/isa_hashes_frozen.txt` is the list of the round-5 product library, and tests/test_abi.py::test_frozen_kernels_are_instruction_identical compares a
fresh build with it -- a host-side edit of a device source file (round 6: yf_engine.hip, yf_stream_scratch.h, yf_fused56.hip.h for the laboratory's dump
build) must leave every line unchanged.

    python tools/isa_hashes.py [library.so]              print
    python tools/isa_hashes.py --check [library.so]      compare with profiles/isa_hashes_frozen.txt, exit 1 on any difference
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FROZEN = os.path.join(ROOT, "profiles", "isa_hashes_frozen.txt")


def hashes(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()                       # one bundle per device object linked into the library (yf_engine.o, yf_fp16.o), back to back
        magic, dis = b"__CLANG_OFFLOAD_BUNDLE__", ""
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, at in enumerate(starts):
            part = os.path.join(tmp, f"bundle{k}.bin")
            open(part, "wb").write(blob[at:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                                  stderr=subprocess.DEVNULL)
            dis += subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    out, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and ln[:1] in " \t" and ln.strip():           # instruction lines are indented; "file format" / "Disassembly of section" lines are not
            out[cur].append(re.sub(r"\s*//.*$", "", ln.strip()))
    return sorted((name, hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], len(body)) for name, body in out.items() if name.startswith("_Z"))


def main():
    args = [a for a in sys.argv[1:] if a != "--check"]
    lib = args[0] if args else os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "libyf_network.so")
    lines = [f"{h} {n} {name}" for name, h, n in hashes(lib)]
    if "--check" in sys.argv:
        want = [ln.strip() for ln in open(FROZEN) if ln.strip() and not ln.startswith("#")]
        diff = sorted(set(lines) ^ set(want))
        print("\n".join(diff) if diff else f"{len(lines)} kernels: instruction streams identical to {os.path.relpath(FROZEN, ROOT)}")
        sys.exit(1 if diff else 0)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

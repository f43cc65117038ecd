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
    out, cur, getpc = {}, None, -9
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and ln[:1] in " \t" and ln.strip():           # instruction lines are indented; "file format" / "Disassembly of section" lines are not
            ins = re.sub(r"\s*//.*$", "", ln.strip())
            # the address of a global (decode tables, ...) is formed pc-relatively: s_getpc_b64, then s_add_u32 / s_addc_u32 with the RESOLVED distance as a literal.
            # That literal moves whenever another kernel is added to the code object; it is a relocation, not an instruction of the kernel: normalised.
            if ins.startswith("s_getpc_b64"):
                getpc = len(out[cur])
            elif len(out[cur]) - getpc <= 4 and re.match(r"s_addc?_u32 s\d+, s\d+, 0x[0-9a-f]+$", ins):
                ins = re.sub(r"0x[0-9a-f]+$", "<pc-relative>", ins)
            out[cur].append(ins)
    for name, body in out.items():                          # what follows a kernel's last s_endpgm is alignment padding up to the next kernel (none behind the last one)
        ends = [i for i, ins in enumerate(body) if ins.startswith("s_endpgm")]
        if ends:
            del body[ends[-1] + 1:]
    return sorted((name, hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], len(body)) for name, body in out.items() if name.startswith("_Z"))


def main():
    args = [a for a in sys.argv[1:] if a != "--check"]
    lib = args[0] if args else os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "libyf_network.so")
    lines = [f"{h} {n} {name}" for name, h, n in hashes(lib)]
    if "--check" in sys.argv:
        want = [ln.strip() for ln in open(FROZEN) if ln.strip() and not ln.startswith("#")]
        missing = sorted(set(want) - set(lines))            # a frozen kernel that is gone or disassembles differently
        extra = sorted(set(lines) - set(want))              # kernels added since (round 6: the second kernel set with the sign-free dense epilogue)
        if missing:
            print("CHANGED or missing:\n" + "\n".join(missing) + "\nnow:\n" + "\n".join(ln for ln in lines if ln.split()[2] in {m.split()[2] for m in missing}))
        print(f"{len(want) - len(missing)} of {len(want)} frozen kernels: instruction streams identical to {os.path.relpath(FROZEN, ROOT)}; {len(extra)} kernels beside them")
        sys.exit(1 if missing else 0)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

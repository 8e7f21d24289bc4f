#!/usr/bin/env python3
"""Fail when any kernel of the built library uses scratch memory (register spills or a stack): reads the AMDGPU metadata notes of
sdnq_amd/libsdnq_hip.so (every embedded gfx950 code object) with llvm-readelf and checks `.private_segment_fixed_size` and
`.vgpr_spill_count` / `.sgpr_spill_count` of every kernel.  Run by __graft_entry__.build().   usage: tools/check_spills.py [lib] [-v]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib):
    """[(kernel name, private bytes, vgpr spills, sgpr spills, vgprs)] of every kernel in the fat binary `lib`."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        # the device code objects sit in the .hip_fatbin section as a clang offload bundle
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        # every embedded ELF starts with \x7fELF; cut them out (the bundle header holds offsets, but scanning is enough here)
        starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
        for n, st in enumerate(starts):
            end = starts[n + 1] if n + 1 < len(starts) else len(data)
            co = os.path.join(tmp, f"co{n}.elf")
            open(co, "wb").write(data[st:end])
            r = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True)
            if r.returncode != 0:
                continue
            cur = {}
            for line in r.stdout.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "name" and "cur_name" not in cur:
                    cur["cur_name"] = v
                if k in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "symbol"):
                    cur[k] = v
                if k == "wavefront_size":  # last key of a kernel record
                    if "symbol" in cur:
                        out.append((cur.get("symbol", "?"), int(cur.get("private_segment_fixed_size", 0)), int(cur.get("vgpr_spill_count", 0)),
                                    int(cur.get("sgpr_spill_count", 0)), int(cur.get("vgpr_count", 0))))
                    cur = {}
    return out


def waterfall_loops(lib):
    """Number of readfirstlane "waterfall" loops in front of memory instructions, over the whole library: a buffer descriptor (or any
    scalar operand of a memory instruction) that the compiler could not keep in SGPRs is fed through `v_readfirstlane ... v_cmp_eq ...
    s_and_saveexec ... <memory instruction> ... s_cbranch_execnz`.  Every descriptor of this library is wave-uniform by construction, so
    the expected count is 0 (round 3: 6-24 such loops per GEMM kernel, the K loop's LDS-DMAs included, went unnoticed for a while)."""
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
        for i, st in enumerate(starts):
            co = os.path.join(tmp, f"co{i}.elf")
            open(co, "wb").write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
            r = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
            n += len(re.findall(r"v_readfirstlane_b32[^\n]*\n(?:[^\n]*\n){0,6}?[^\n]*v_cmp_eq_u64[^\n]*\n(?:[^\n]*\n){0,4}?[^\n]*s_and_saveexec_b64", r.stdout))
    return n


def main():
    lib = next((a for a in sys.argv[1:] if not a.startswith("-")), os.path.join(ROOT, "sdnq_amd", "libsdnq_hip.so"))
    res = kernel_resources(lib)
    if not res:
        print("no kernels found in", lib)
        return 2
    bad = [r for r in res if r[1] > 0 or r[2] > 0]  # SGPR spills go to VGPR lanes (no scratch memory): reported with -v only
    sg = [r for r in res if r[3] > 0 and r not in bad]
    if "-v" in sys.argv:
        for r in sorted(res, key=lambda r: -r[4])[:25]:
            print(f"{r[4]:4d} vgprs  scratch {r[1]:5d}  {r[0][:150]}")
    print(f"{len(res)} kernels, {len(bad)} with scratch memory / vector-register spills ({len(sg)} more keep spilled SGPRs in VGPR lanes)")
    for r in bad:
        print(f"  scratch {r[1]} B, vgpr spills {r[2]}, sgpr spills {r[3]}: {r[0][:200]}")
    if "--waterfalls" in sys.argv:
        wf = waterfall_loops(lib)
        print(f"{wf} readfirstlane waterfall loops in front of memory instructions")
        if wf:
            return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

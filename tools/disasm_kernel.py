#!/usr/bin/env python3
"""Development aid: disassemble the kernels of sdnq_amd/libsdnq_hip.so whose (mangled) name contains every given substring.
usage: tools/disasm_kernel.py <substr> [<substr> ...] [--count]   (--count: instruction histogram instead of the listing)"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.environ.get("SDNQ_HIP_LIB") or os.path.join(ROOT, "sdnq_amd", "libsdnq_hip.so")
subs = [a for a in sys.argv[1:] if not a.startswith("--")]
count = "--count" in sys.argv
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    for n, st in enumerate(starts):
        co = os.path.join(tmp, f"co{n}.elf")
        open(co, "wb").write(data[st:(starts[n + 1] if n + 1 < len(starts) else len(data))])
        r = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
        cur, keep, hist = None, False, collections.Counter()
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if keep and count:
                    print(cur); [print(f"  {k:40s} {v}") for k, v in hist.most_common(40)]
                cur, hist = m.group(1), collections.Counter()
                keep = all(s in cur for s in subs)
                if keep and not count:
                    print(line)
                continue
            if keep:
                if count:
                    t = line.split()
                    if t: hist[t[0]] += 1
                else:
                    print(line)
        if keep and count:
            print(cur); [print(f"  {k:40s} {v}") for k, v in hist.most_common(40)]

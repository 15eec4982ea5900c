#!/usr/bin/env python3
"""Scans the gfx950 code objects of libparesis_hip.so for the store-data hazard found in round 6 (gpurun_out/r6s33): a buffer store of
more than 64 bits whose scalar-offset operand is a REGISTER, followed at once by a vector-ALU instruction that overwrites one of the
store's data registers.  The compiler pads that case only when the scalar offset is an immediate (GCNHazardRecognizer: "this hazard only
exists if the instruction is not using a register in the soffset field"); on gfx950 the upper half of the data is then read too late
in lanes 12-15 of every row of 16 -- silently wrong bytes in memory.  Returns the offending sites; tests/test_host_cpu.py keeps the
list empty.   python tools/check_store_hazard.py [lib]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+)\b)")
WAIT_STATES = 2          # instructions looked at behind the store (an s_nop N counts N + 1)


def sites(lib=None):
    lib = lib or os.path.join(ROOT, "paresis_amd", "libparesis_hip.so")
    tmp = tempfile.mkdtemp(prefix="psx_hz_")
    out = []
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            kernel = "?"
            lines = [l.split("//")[0].rstrip() for l in txt.splitlines()]
            for i, l in enumerate(lines):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
                if m:
                    kernel = m.group(1)
                    continue
                m = STORE.match(l)
                if not m:
                    continue
                lo, hi, soff = int(m.group(2)), int(m.group(3)), m.group(5)
                if not re.match(r"^s\d+$", soff):          # immediate / off: the compiler pads these itself
                    continue
                slots, j = 0, i + 1
                while slots < WAIT_STATES and j < len(lines):
                    n = lines[j].strip()
                    j += 1
                    if not n or n.endswith(":"):
                        continue
                    mn = re.match(r"s_nop (\d+)", n)
                    if mn:
                        slots += int(mn.group(1)) + 1
                        continue
                    d = VDST.match(n)
                    if d:
                        a = int(d.group(1) if d.group(1) is not None else d.group(3))
                        b = int(d.group(2) if d.group(2) is not None else d.group(3))
                        if a <= hi and b >= lo:
                            out.append((kernel, l.strip(), n))
                            break
                    slots += 1
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    found = sites(sys.argv[1] if len(sys.argv) > 1 else None)
    for k, s, n in found:
        print(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110])
        print("    ", s)
        print("    ", n)
    print(len(found), "site(s)")

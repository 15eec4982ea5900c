#!/usr/bin/env python3
"""Register / scratch use of every kernel in libparesis_hip.so, read from the code-object notes (no GPU needed).

    python tools/kernel_resources.py [filter]       # prints: vgprs agprs spills scratch-bytes sgprs lds name

`kernels(path)` returns the same as a list of dicts (tests/test_host_cpu.py checks that no shipped kernel spills).
The gfx950 code objects are unbundled from the host library with llvm-objdump --offloading in a scratch directory.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "sgpr_count",
          "group_segment_fixed_size")


def kernels(lib=None):
    lib = lib or os.path.join(ROOT, "paresis_amd", "libparesis_hip.so")
    tmp = tempfile.mkdtemp(prefix="psx_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True).stdout
            for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                block = ".agpr_count:" + block
                e = {}
                for k in FIELDS:
                    m = re.search(r"\.%s:\s+(\d+)" % k, block)
                    e[k] = int(m.group(1)) if m else None
                m = re.search(r"\.name:\s+(\S+)", block)
                e["symbol"] = m.group(1) if m else "?"
                out.append(e)
        names = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] + [e["symbol"] for e in out], capture_output=True,
                               text=True).stdout.splitlines() if out and os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else []
        for e, n in zip(out, names):
            e["name"] = n
        for e in out:
            e.setdefault("name", e["symbol"])
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print("vgpr agpr vspill sspill scratch sgpr   lds  kernel")
    for e in kernels():
        if pat and pat not in e["name"]:
            continue
        print("%4d %4d %6d %6d %7d %4d %6d  %s" % (e["vgpr_count"], e["agpr_count"] or 0, e["vgpr_spill_count"],
                                                 e["sgpr_spill_count"] or 0, e["private_segment_fixed_size"], e["sgpr_count"],
                                                 e["group_segment_fixed_size"] or 0, e["name"][:150]))

#!/usr/bin/env python3
"""Diagnostic: bench.py with a dummy device allocation of PAD KiB made BEFORE the library (and its code objects) is loaded --
shifts where the runtime places the kernels' code and every later buffer.     python tools/pad_bench.py PAD_KIB [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pad_kib = int(sys.argv[1])
torch.cuda.init()
keep = torch.empty(max(1, pad_kib) << 10, dtype=torch.uint8, device="cuda") if pad_kib else None
print("pad %d KiB at %x" % (pad_kib, keep.data_ptr() if keep is not None else 0), file=sys.stderr)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()

import ctypes, os, sys, time, types
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from paresis_amd import _lib, ops
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()
lib = _lib.lib()
N, ov = 16384, 4
n = N // ov
imgs = [torch.rand((N, N), device="cuda") for _ in range(3)]
det = ops.DetectorPlan(N, N, ov, n, n, 10.0 * 3.6 / 141.6 / 6.0 * ov / 2.355, 1.2)
outs = [torch.empty((n, n), dtype=torch.float32, device="cuda") for _ in range(3)]
def f():
    for i in range(3): det.detect(imgs[i], out=outs[i])
f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize()
print("detect 16384^2 -> 4096^2: %.3f ms per image" % ((time.perf_counter() - t0) / 15 * 1e3))
lib.psx_profile_enable(1); f(); torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16); lib.psx_profile_summary(buf, len(buf)); print(buf.value.decode())

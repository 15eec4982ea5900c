#!/usr/bin/env python3
"""Diagnostic (never a timed run): where does a workgroup of the LDS Fresnel row pass spend its wall time?"""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paresis_amd import _lib, ops
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # with PSX_STAMP_PASS1=1: distances sharing pass 1
lib = _lib.lib()
plan = ops.FresnelPlan(N, N, max_dist=ND)
w = (torch.randn(N, N, device="cuda") + 1j * torch.randn(N, N, device="cuda")).to(torch.complex64)
out = torch.empty((N, N), dtype=torch.float32, device="cuda")
du = 2 * np.pi / (N * 2.9e-6)
outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in range(ND)]
aa = [3e-12 * (k + 1) for k in range(ND)]
def run():
    plan.propagate(aa, [0.0] * ND, (du, du), wave_in=w, want_wave=[False] * ND, inten_out=outs)
for _ in range(3):
    run()
nblk = N
buf = torch.zeros((nblk, 32), dtype=torch.int64, device="cuda")
lib.psx_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.psx_debug_stamps(None)
lib.psx_profile_enable(1)
for _ in range(20):
    run()
torch.cuda.synchronize()
pbuf = ctypes.create_string_buffer(1 << 14)
lib.psx_profile_summary(pbuf, len(pbuf))
lib.psx_profile_enable(0)
print("event pairs, 20 un-stamped calls (name launches total_ms):", pbuf.value.decode().replace("\n", "; "))
s = buf.cpu().numpy().astype(np.float64)
s = s[s[:, 13] > 0]
names = ["prologue: first fetch + spread (once per launch)", "-> round 2 start (round 1 not stamped)", "fwd A", "barrier 1", "fwd B", "wave sync",
         "C + xH + invC", "wave sync", "inv B", "barrier 2", "inv A: LDS reads + barrier 3", "twiddle, butterfly, stores",
         "barrier 4 (next group spread by the loaders)"]
d = np.diff(s[:, :14], axis=1) * 10.0   # 100 MHz ticks -> ns
rnd = d[:, 2:].sum(1).mean()
print("workgroups:", len(s), " steady-state round %.2f us" % (rnd / 1e3))
# the launch as the chip sees it (the 100 MHz counter is global): first workgroup's first stamp (after its twiddle tables are
# in LDS) -> last workgroup's last stamp of the stamped round; against the launch's event-pair time this separates dispatch +
# table fill + drain from the rounds themselves (where do the ~40 us of a 2-round launch go?)
print("  stamped span: first stamp of any workgroup -> last stamp of the stamped round, all workgroups: %.2f us; per workgroup "
      "stamp 0 -> 13: mean %.2f, max %.2f us; start skew (stamp 0, max - min) %.2f us; end skew (stamp 13) %.2f us"
      % ((s[:, 13].max() - s[:, 0].min()) / 100.0, (s[:, 13] - s[:, 0]).mean() / 100.0, (s[:, 13] - s[:, 0]).max() / 100.0,
         (s[:, 0].max() - s[:, 0].min()) / 100.0, (s[:, 13].max() - s[:, 13].min()) / 100.0))
for n, v in zip(names, d.mean(0)):
    print("  %-52s %7.2f us  %5.1f %%" % (n, v / 1e3, 100 * v / rnd))
ld = s[:, 16:21]
base = s[:, 4:5]      # engine passed barrier 1
print("loader wave 12 (relative to the engine leaving barrier 1):")
for n, c in zip(["fetch start", "fetch issued", "spread start (barrier 3 passed)", "spread done", "barrier 4 passed"], range(5)):
    print("  %-34s %7.2f us" % (n, ((ld[:, c:c+1] - base) * 10.0).mean() / 1e3))
if np.any(s[:, 21] > 0):       # partitioned / DIF loaders: the window moves in two halves between barriers (3) and (4)
    print("  first half spread %.2f, second half fetched (issued) %.2f us after barrier 3" %
          tuple(((s[:, c:c+1] - ld[:, 2:3]) * 10.0).mean() / 1e3 for c in (21, 22)))
if np.any(s[:, 23] > 0):       # DIF, round O: inside "twiddle, butterfly, stores"
    print("  round O after barrier 3: butterfly done %.2f, recombination twiddles done %.2f, ye added %.2f, stores issued %.2f us" %
          tuple(((s[:, c:c+1] - s[:, 11:12]) * 10.0).mean() / 1e3 for c in (23, 14, 15, 12)))
print("engine: barrier 2 passed %.2f, barrier 3 passed %.2f, stores issued %.2f, barrier 4 passed %.2f us" % tuple(((s[:, c:c+1] - base) * 10.0).mean() / 1e3 for c in (10, 11, 12, 13)))

import os, sys, ctypes, time
sys.path.insert(0, "/root/repo")
import torch
from paresis_amd import ops, _lib
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()
lib = _lib.lib()
N = 16384
plan = ops.FresnelPlan(N, N, max_dist=1)
u = (torch.randn(N, N, device="cuda") * 0.1 + 1.0).to(torch.complex64)
inten = torch.zeros((N, N), dtype=torch.float32, device="cuda")
f = lambda: plan.propagate([2e-12], [0.1], (3e5, 3e5), wave_in=u, want_wave=[False], inten_out=[inten])
f(); torch.cuda.synchronize()
lib.psx_profile_enable(1)
for _ in range(3): f()
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.psx_profile_summary(buf, len(buf)); print(buf.value.decode())
print("plan bytes", plan.nbytes if hasattr(plan, "nbytes") else "")

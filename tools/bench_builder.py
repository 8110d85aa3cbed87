"""Developer script: bandwidth of the device-side parameter builder (HBM-bound gather kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from cmpc_amd.solver import DeviceRecordBuilder
sc = wl.scene(); spec = ProblemSpec(N=20)
bld = DeviceRecordBuilder(sc, device="cuda:0")
for B in (65536, 1 << 20):
    t = torch.randint(0, sc.t_max(20), (B,), dtype=torch.int32, device="cuda:0")
    state = torch.randn((B, 16), dtype=torch.float64, device="cuda:0")
    out = torch.empty((B, spec.nrec), dtype=torch.float64, device="cuda:0")
    for _ in range(3): bld.build(spec, t, state, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): bld.build(spec, t, state, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = B * spec.nrec * 8 / 1e9
    print(f"B={B}: {ms:.3f} ms per call, {gb / (ms * 1e-3):.0f} GB/s written ({gb:.3f} GB), {B / (ms * 1e-3) / 1e6:.1f} M records/s")

"""Diagnostic: run the 256-instance rollout of tests/test_gpu_walk.py and dump the records / warm starts of the
ticks at which instances stop (gpurun_out/rollout_diag.npz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from cmpc_amd.rollout import BatchedRollout
scene = wl.scene()
hwm = np.loadtxt(os.path.join(ROOT, "tests", "golden", "measured_hw_cuhw.txt"))
spec = ProblemSpec(N=10)
B, t0, ticks = 256, 150, 380
rng = np.random.default_rng(22)
com, dcom = scene.nominal_state(np.full(B, t0))
com = com + rng.uniform(-0.003, 0.003, size=(B, 3))
dcom = dcom + rng.normal(0, 0.01, size=(B, 3))
ro = BatchedRollout(scene, spec, B, device="cuda:0", hw_measured=hwm, hw_offset=rng.normal(0, 0.05, size=(B, 3)))
ro.reset(t0, com, dcom)
dump = []
for i in range(ticks):
    was = ro.alive.clone(); warm = None if ro.warm is None else ro.warm.clone()
    ro.step()
    died = (was & ~ro.alive).nonzero().flatten().tolist()
    for b in died:
        dump.append((i, b, int(ro.last_status[b]), int(ro.last_iters[b]), ro.last_records[b].cpu().numpy(),
                     None if warm is None else warm[b].cpu().numpy()))
        print("tick", t0 + i, "instance", b, "status", int(ro.last_status[b]), "iters", int(ro.last_iters[b]))
print("alive", int(ro.alive.sum()), "of", B)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "rollout_diag.npz"), tick=[d[0] + t0 for d in dump], inst=[d[1] for d in dump],
         status=[d[2] for d in dump], rec=np.array([d[4] for d in dump]), warm=np.array([d[5] for d in dump]))

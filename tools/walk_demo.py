"""Developer script (GPU box): BASELINE config 1 end to end.
 (a) the drop-in class with the HIP solver through the whole shipped walk, t = 0 ... 1900 (N = 10, measured momentum
     from the reference's recording), one instance per tick as simulation.py does;
 (b) the same walk as B parallel closed loops (BatchedRollout, per-instance plans, perturbed initial states and
     momentum offsets) -- what the batched solver is for."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cmpc_amd
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from cmpc_amd.rollout import BatchedRollout
from test_walk import build_walk, measured_hw
import centroidal_mpc_vertices

hw = measured_hw()
walk, planner, ftg, mpc, sc = build_walk(centroidal_mpc_vertices.centroidal_mpc, N=10, hw=hw)
T = min(1900, sc.t_max(10))
t0 = time.perf_counter()
log = walk.run(T)
dt = time.perf_counter() - t0
st, it = log['status'], log['iterations']
ref = sc.com_tab[log['t'] + 1, 0:3]
print(f"(a) drop-in class, t = 0 ... {T}: {T} ticks, status counts {np.bincount(st, minlength=4).tolist()}, iterations mean {it.mean():.1f} "
      f"max {it.max()}, {1e3 * dt / T:.1f} ms per tick (host loop + one B = 1 launch), write-backs {int(log['counter'].sum())}, "
      f"max |com - ref| {np.abs(log['com'] - ref).max(0).round(4).tolist()}, max |hw_des| {np.abs(log['hw_des']).max():.2f}")

B, start, ticks = 4096, 150, 600
spec = ProblemSpec(N=10)
rng = np.random.default_rng(7)
com, dcom = sc.nominal_state(np.full(B, start))
com = com + rng.uniform(-0.003, 0.003, size=(B, 3)); dcom = dcom + rng.normal(0, 0.01, size=(B, 3))
ro = BatchedRollout(sc, spec, B, device="cuda:0", hw_measured=hw, hw_offset=rng.normal(0, 0.05, size=(B, 3)))
ro.reset(start, com, dcom)
torch.cuda.synchronize(); t0 = time.perf_counter()
its = []
for i in range(ticks):
    ro.step(); its.append(float(ro.last_iters.double().mean()))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
alive = int(ro.alive.sum())
moved = (ro.plan_pos.cpu().numpy() != sc.plan_pos[None]).any(axis=2).sum(axis=1)
print(f"(b) {B} parallel walks, t = {start} ... {start + ticks}: {alive} of {B} alive after {ticks} ticks, mean iterations {np.mean(its):.1f}, "
      f"{1e3 * dt / ticks:.1f} ms per tick = {B * ticks / dt:.0f} closed-loop solves/s (builder + solve + write-back + advance), "
      f"plan entries rewritten per walk: min {moved.min()} max {moved.max()}")

# (c) the same B closed loops with the whole-body QP inside every tick (code/simulation.py:193-232 per robot: the MPC, then
# get_joint_torques on the MPC's desired CoM).  The rigid-body model is the synthetic stand-in of the GPU test (fixed task
# Jacobians per robot, the CoM task's feed-forward acceleration = the MPC's CoM_acc): what is timed is the whole tick on
# the device -- records, MPC solve, plan write-back, QP assembly of the stand-in, QP solve.
from cmpc_amd import wbc
ro = BatchedRollout(sc, spec, B, device="cuda:0", hw_measured=hw, hw_offset=rng.normal(0, 0.05, size=(B, 3)))
Hq0, Fq0, Mm, hh, Jc = (torch.from_numpy(np.ascontiguousarray(np.tile(a, (B // 256,) + (1,) * (a.ndim - 1)))).cuda() for a in wl.wbc_synthetic(256, seed=77))
Jcom = torch.from_numpy(np.tile(rng.normal(0, 0.4, size=(256, 3, 30)), (B // 256, 1, 1))).cuda()
Jcom[:, :, 3:6] += torch.eye(3, dtype=torch.float64, device="cuda:0")
Hq = (Hq0 + 50.0 * Jcom.transpose(1, 2) @ Jcom).contiguous()


def model(rollout, desired):
    g = torch.cat([desired["gamma_l"][:, None].expand(-1, 6), desired["gamma_r"][:, None].expand(-1, 6)], dim=1)
    Fq = (Fq0 - 50.0 * torch.einsum("brn,br->bn", Jcom, desired["com_acc"])).contiguous()
    return Hq, Fq, Mm, hh, (Jc * g[:, :, None]).contiguous()


ro.attach_whole_body(wbc.BatchedInverseDynamicsQP(foot_size=0.1, mu=0.5, device="cuda:0"), model)
ro.reset(start, com, dcom)
for i in range(3):
    ro.step()
ro.reset(start, com, dcom)
torch.cuda.synchronize(); t0 = time.perf_counter()
okq = 0
for i in range(ticks):
    ro.step(); okq += int((ro.last_wbc[3] == 0).sum())
torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print(f"(c) the same {B} walks with the whole-body QP inside every tick: {1e3 * dt2 / ticks:.1f} ms per tick = {B * ticks / dt2:.0f} whole "
      f"ticks/s (MPC + QP; {1e3 * (dt2 - dt) / ticks:.1f} ms of it for the stand-in model's assembly and the QP), QPs converged "
      f"{okq / (B * ticks):.4f}, {int(ro.alive.sum())} of {B} alive")

"""Developer script: iterations per tick of the closed-loop flat-ground walk (perfect tracking, measured angular
momentum) with the solver state resumed at barrier level MU_WARM -- the C oracle compiled with -DMU_WARM=<level> into
/tmp.  usage: python tools/warm_walk.py N MU_WARM [ticks]      e.g.  python tools/warm_walk.py 10 1e-7"""
import sys, ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, cmpc_amd, subprocess
from cmpc_amd import workloads as wl
from cmpc_amd.problem import ProblemSpec
from oracle import oracle_lib as ol
HW = np.loadtxt(os.path.join(ROOT, 'tests', 'golden', 'measured_hw_cuhw.txt'))
N = int(sys.argv[1]); mw = sys.argv[2]; T1 = int(sys.argv[3]) if len(sys.argv) > 3 else 1900
so = f"/tmp/oracle_mw_{mw}.so"
subprocess.check_call(["gcc","-O3","-march=x86-64-v3","-fopenmp","-fPIC","-shared",f"-DMU_WARM={mw}","-o",so,os.path.join(ROOT, "oracle", "cmpc_oracle.c"),"-lm"])
ol._LIB = ctypes.CDLL(so)
spec = ProblemSpec(N=N); sc = wl.scene()
cs = ol.default_spec(N=N, nv=4, tol=1e-8, max_iter=100)
com, dcom = sc.nominal_state(np.array([0])); com = com[0]; dcom = dcom[0]
theta = np.zeros(3); warm = None; state = None; its = []; sts = []
for t in range(0, T1):
    rec = sc.build_records(spec, np.array([t]), com[None], dcom[None], HW[t][None], theta[None], np.zeros(1), np.zeros(1), np.full(1, wl.HRP4_MASS), np.full(1, 0.5))
    sol, state, st, it, kkt = ol.solve_batch_state(cs, rec, warm=warm, state=state, verbose=(t == int(os.environ.get("VT", -1))))
    its.append(it[0]); sts.append(st[0])
    X = sol[0, :20*(N+1)].reshape(N+1, 20)
    com, dcom, theta = X[1, 0:3].copy(), X[1, 3:6].copy(), X[1, 9:12].copy()
    warm = sol
    if st[0] not in (0, 3): print("FAIL at", t, st); break
its = np.array(its); sts = np.array(sts)
print(f"N={N} MU_WARM={mw} ticks {len(its)}: mean {its.mean():.2f} median {np.median(its):.0f} p90 {np.percentile(its,90):.0f} max {its.max()} status3 {(sts==3).sum()}")
print(its[200:420])

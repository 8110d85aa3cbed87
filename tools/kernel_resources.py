"""Developer script: cross-compile the HIP sources and print registers / spills / scratch / LDS of every kernel
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).  usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd")
for name in ("cmpc_hip.hip", "wbc_qp.hip"):
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", os.path.join(PKG, "csrc", name)] + sys.argv[1:],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(1)
    cur = None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]+ (?:Function )?Name: (\S+)", ln) or re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::|\(.*", "", cur)
            print(f"{cur}:", end="")
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and cur:
            print(f"  {m.group(1).split(' [')[0]} {m.group(2)}", end="")
            if m.group(1).startswith("LDS"):
                print()

#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy figures of the kernels of one source file (compiler remarks).
    python tools/kernel_regs.py attention.hip [substring]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "peneo_amd", "csrc", sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] if "pair_bwd" in src else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
                    "--cuda-device-only", "-c", src, "-o", "/dev/null"] + extra, capture_output=True, text=True)
cur = None
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
    if k == "Function Name":
        if cur and want in cur["name"]:
            print(f"vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} sgpr {cur.get('SGPRs'):>4} scratch {cur.get('ScratchSize [bytes/lane]'):>5} occ {cur.get('Occupancy [waves/SIMD]')}  {cur['name'][:120]}")
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
    elif cur is not None:
        cur[k] = v
if cur and want in cur["name"]:
    print(f"vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} sgpr {cur.get('SGPRs'):>4} scratch {cur.get('ScratchSize [bytes/lane]'):>5} occ {cur.get('Occupancy [waves/SIMD]')}  {cur['name'][:120]}")

#!/bin/bash
# usage: tools/pmc.sh <kernel-name-substring> <python script> ; collects a few PMC groups (separate passes) and prints per-launch averages
K="$1"; shift
export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES_EQ_64"; do
  rm -rf gpurun_out/pmc_tmp
  rocprofv3 --pmc $C -d gpurun_out/pmc_tmp -o p -f csv -- python3 "$@" > /dev/null 2>&1
  python3 - "$K" <<'PY'
import csv, glob, collections, sys
k = sys.argv[1]
fs = glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv")
if not fs:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if k in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in acc.items():
    print(f"{c:34s} {v / n[c]:16.1f}   (n={n[c]})")
PY
done

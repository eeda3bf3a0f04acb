#!/bin/bash
# usage: bash tools/pmc_ta.sh <kernel-name-substring> <python script> : vector-memory path counters of one kernel (separate passes,
# --pmc only): texture-addresser busy, its stalls behind the L1 (TCP), L1 -> L2 read requests and their accumulated latency.
K="$1"; shift
export TMPDIR=/tmp
for C in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf gpurun_out/pmc_tmp
  rocprofv3 --pmc $C -d gpurun_out/pmc_tmp -o p -f csv -- python3 "$@" > /dev/null 2>&1
  python3 - "$K" <<'PY'
import csv, glob, collections, sys
k = sys.argv[1]
fs = glob.glob("gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if k in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in acc.items():
    print(f"{c:38s} {v / n[c]:18.1f}   (n={n[c]})")
PY
done

#!/bin/bash
# usage: tools/pmc_cycles.sh <kernel-name-substring> <python script>: shader cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs) next to the
# kernel's duration in the same pass -> the clock the kernel actually ran at (the pair kernels sit at the package power limit)
K="$1"; shift
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_tmp -o p -f csv -- python3 "$@" > /dev/null 2>&1
python3 - "$K" <<'PY'
import csv, glob, sys
k = sys.argv[1]
cyc = [float(r["Counter_Value"]) for f in glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv") for r in csv.DictReader(open(f)) if k in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for f in glob.glob("gpurun_out/pmc_tmp/*kernel_trace.csv") for r in csv.DictReader(open(f)) if k in r["Kernel_Name"]]
if cyc: print(f"{k}: {sum(cyc) / len(cyc) / 8 / 1e6:.3f} M cycles per launch (n={len(cyc)})", end="")
if dur: print(f"   {sum(dur) / len(dur):.1f} us per launch under the counter pass -> {sum(cyc) / len(cyc) / 8 / (sum(dur) / len(dur)) / 1e3:.2f} GHz" if cyc else f"{sum(dur)/len(dur):.1f} us")
else: print()
PY

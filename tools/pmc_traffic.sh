#!/bin/bash
# usage: bash tools/pmc_traffic.sh <kernel-substring> <python script...>
# HBM traffic per launch of one kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass on
# gfx950).  Units are KB; per /opt/skills/guides/MI355X_MICROARCH.md the gfx950 FETCH_SIZE of wide coalesced reads counts
# half the bytes, so reads are doubled; WRITE_SIZE is reported as-is (uncalibrated).
K="$1"; shift
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_tmp
  rocprofv3 --pmc $C -d gpurun_out/pmc_tmp -o p -f csv -- python3 "$@" > /dev/null 2>&1
  python3 - "$K" "$C" <<'PY'
import csv, glob, sys
k, c = sys.argv[1], sys.argv[2]
fs = glob.glob("gpurun_out/pmc_tmp/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); sys.exit(0)
tot, n = 0.0, 0
for r in csv.DictReader(open(fs[0])):
    if k in r["Kernel_Name"] and r["Counter_Name"] == c:
        tot += float(r["Counter_Value"]); n += 1
if n:
    kb = tot / n
    corr = 2.0 if c == "FETCH_SIZE" else 1.0
    print(f"{c}: {kb:.1f} KB/launch raw over {n} launches -> {kb * corr * 1024 / 1e6:.3f} MB/launch after gfx950 correction x{corr:g}")
else:
    print(f"{c}: kernel '{k}' not found")
PY
done

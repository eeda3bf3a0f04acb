#!/bin/bash
# which kernels the vendor library (torch.mm -> hipBLASLt / rocBLAS) runs on the model's GEMM shapes, and with what resources:
# kernel name (macro-tile, wave tiling, depth-U, stream-k / split flags are encoded in it), grid, workgroup, LDS, VGPR / AGPR / scratch,
# average duration -> gpurun_out/$1/vendor_gemm_configs.txt   (inspect the opponent)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-vendor}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python3 tools/run_blas_ref.py > $OUT/run.log 2>&1
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$T" > $OUT/vendor_gemm_configs.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if not (n.startswith("Cijk") or "gemm" in n.lower() or "splitk" in n.lower()):
        continue
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"),
           r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("Scratch_Size", "?"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
print("calls   avg_us   grid  wg  lds  vgpr  agpr  scratch  kernel")
for (n, g, w, l, v, a_, s), (c, t) in agg.items():
    print(f"{c:5d} {t / c:8.1f} {g:>7} {w:>4} {l:>6} {v:>4} {a_:>4} {s:>4}  {n[:400]}")
PY
rm -rf $OUT/prof
cat $OUT/run.log | grep -v amdgpu.ids

#!/bin/bash
# L2 / texture-path counters of the GEMM kernels (8-wave core, 4-wave fine schedule, library) at the gate|up shape.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm_l2
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "(TCC|TCP|TA|TD)_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
i=0
for v in 7 260; do
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  PM_GEMM_VARIANT=$v PM_PROBE_LIB=$([ $v = 7 ] && echo 1 || echo 0) timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python tools/microbench.py gemmprobe > $OUT/p$i.log 2>&1
done
done
OUT="$OUT" python - <<'PYEOF' | tee $OUT/summary.txt
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(m in k for m in ("gemm4_kernel", "gemm_kernel", "Cijk")):
            continue
        acc[k[:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PYEOF
wc -l $OUT/counters_available.txt; grep -E "TCC_HIT|TCC_MISS|TA_BUSY|TA_TA|TCP_TCC_READ|PENDING" $OUT/counters_available.txt | head -20

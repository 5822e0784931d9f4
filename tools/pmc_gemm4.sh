#!/bin/bash
# PMC passes for the GEMM kernels (8-wave core, one-wave-per-SIMD kernel, library) at the gate|up shape: cycles, matrix-pipe busy, waits, sustained clock.
# usage (GPU box, repo root): bash tools/pmc_gemm4.sh   -> gpurun_out/pmc_gemm4/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm4
rm -rf $OUT && mkdir -p $OUT
i=0
for v in 7 68; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  PM_GEMM_VARIANT=$v PM_PROBE_LIB=$([ $v = 7 ] && echo 1 || echo 0) timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python tools/microbench.py gemmprobe > $OUT/p$i.log 2>&1
done
done
OUT="$OUT" python - <<'PYEOF' | tee $OUT/summary.txt
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.environ["OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(m in k for m in ("gemm4_kernel", "gemm_kernel", "Cijk")):
            continue
        acc[k[:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.environ["OUT"] + "/p*/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if any(m in k for m in ("gemm4_kernel", "gemm_kernel", "Cijk")):
            dur[k[:100]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    print(k)
    if dur.get(k):
        v = sorted(dur[k]); print(f"   duration_us                  n={len(v):3d} median={v[len(v)//2]:.1f} min={v[0]:.1f}")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
PYEOF

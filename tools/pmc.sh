#!/bin/bash
# PMC passes over any command (one counter group per run; --pmc is never combined with tracing, as gpurun requires).
# usage (on the GPU box, from the repo root):  bash tools/pmc.sh <out-name> <kernel-name-substring> <command...>
# Writes gpurun_out/pmc_<out-name>/summary.txt: per matching kernel, the mean of every counter over its launches.
NAME=$1; MATCH=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$NAME
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- "$@" > $OUT/p$i.log 2>&1
done
MATCH="$MATCH" OUT="$OUT" python - <<'PYEOF' | tee $OUT/summary.txt
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ["OUT"] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if not any(m in k for m in os.environ["MATCH"].split("|")):  # "a|b": any of the substrings
            continue
        acc[k[:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
PYEOF

#!/bin/bash
# Sustained shader clock of the attention kernels (GRBM_GUI_ACTIVE per dispatch / dispatch duration), shipped builds vs the round-4 variants.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4ao
rm -rf $OUT && mkdir -p $OUT
PM_ATTN_VARIANTS= timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/a -o a -- python tools/microbench.py attn16k > $OUT/a.log 2>&1
PM_ATTN_VARIANTS=0:47,1:4 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/b -o b -- python tools/microbench.py attn16k > $OUT/b.log 2>&1
OUT="$OUT" python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
for tag in ("a", "b"):
    act, dur, name = {}, {}, {}
    for f in glob.glob(os.environ["OUT"] + f"/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "attn" in row["Kernel_Name"]:
                act[row["Dispatch_Id"]] = float(row["Counter_Value"]); name[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0][:60]
    for f in glob.glob(os.environ["OUT"] + f"/{tag}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
    per = collections.defaultdict(list)
    for d, a in act.items():
        if d in dur and dur[d] > 150:  # the 16 384-token launches only
            per[name[d]].append((dur[d], a))
    for k, v in per.items():
        v.sort()
        med = v[len(v) // 2]
        print(f"{tag} {k:58s} n={len(v):2d} median {med[0]:7.1f} us  GUI_ACTIVE {med[1]:.4g}  clock {med[1] / med[0] / 1e3:.3f} GHz")
PY

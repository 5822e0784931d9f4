#!/bin/bash
# Sustained shader clock per kernel class INSIDE the bench step: GRBM_GUI_ACTIVE per dispatch (summed over 8 XCDs) / dispatch duration from the same run's kernel trace.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4ap
rm -rf $OUT && mkdir -p $OUT
timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/a -o a -- python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-parity > $OUT/a.log 2>&1
tail -1 $OUT/a.log | cut -c1-200
OUT="$OUT" python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
act, dur, name = {}, {}, {}
for f in glob.glob(os.environ["OUT"] + "/a/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[row["Dispatch_Id"]] = float(row["Counter_Value"]); name[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0][:70]
for f in glob.glob(os.environ["OUT"] + "/a/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
per = collections.defaultdict(list)
for d, a in act.items():
    if d in dur and dur[d] > 80:
        per[name[d]].append((dur[d], a))
rows = []
for k, v in per.items():
    tot_us = sum(x[0] for x in v); tot_act = sum(x[1] for x in v)
    rows.append((tot_us, k, len(v), tot_act / tot_us / 8 / 1e3))
for tot_us, k, n, ghz in sorted(rows, reverse=True)[:22]:
    print(f"{k:72s} n={n:5d} total {tot_us / 1e3:8.1f} ms  clock {ghz:.3f} GHz")
PY
rm -rf $OUT/a

"""Turn a rocprofv3 `--kernel-trace --stats` results.db (rocpd sqlite) into a per-kernel summary (markdown + csv).

    python tools/rocprof_summary.py gpurun_out/prof_mb8/r01_results.db profiles/r01_bench_mb8 "command line that was profiled" [--window]

--window: count only the dispatches between the FIRST and the LAST launch of the marker kernel `mfma_probe_kernel` (tools/microbench.py brackets its timed
steps with `trace_mark()`): model construction, weight init and the warm-up step stay out of the summary (VERDICT r5 'next' 4c).
"""

import glob
import sqlite3
import sys


def main():
    window = "--window" in sys.argv
    argv = [a for a in sys.argv if a != "--window"]
    db, out, cmd = argv[1], argv[2], (argv[3] if len(argv) > 3 else "")
    con = sqlite3.connect(glob.glob(db)[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    where, note = "", ""
    if window:
        marks = [r[0] for r in con.execute(f"select d.start from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%mfma_probe_kernel%' order by d.start")]
        if len(marks) >= 2:
            where = f" where d.start > {marks[0]} and d.start < {marks[-1]} "
            note = f" (window between the marker launches: {(marks[-1] - marks[0]) / 1e6:.1f} ms of wall time)"
        else:
            note = " (--window: fewer than two marker launches found; whole trace)"
    tot = con.execute(f"select sum(d.end-d.start) from {kd} d {where}").fetchone()[0]
    rows = list(con.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        f"from {kd} d join {ks} s on d.kernel_id=s.id {where} group by s.kernel_name order by 3 desc"))
    cmd = cmd + note
    with open(out + ".csv", "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,arch_vgpr,accum_vgpr,sgpr,lds_bytes\n")
        for n, c, t, a, mn, mx, vg, ag, sg, lds in rows:
            f.write(f"\"{n}\",{c},{t},{a:.0f},{mn},{mx},{100*t/tot:.3f},{vg},{ag},{sg},{lds}\n")
    with open(out + ".md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\ntotal kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches\n\n")
        f.write("| % | total ms | calls | avg us | min us | max us | VGPR | AGPR | LDS B | kernel |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for n, c, t, a, mn, mx, vg, ag, sg, lds in rows[:45]:
            f.write(f"| {100*t/tot:.1f} | {t/1e6:.2f} | {c} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {vg} | {ag} | {lds} | `{n[:120]}` |\n")
    print("wrote", out + ".md")


if __name__ == "__main__":
    main()

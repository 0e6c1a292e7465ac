#!/usr/bin/env python3
"""Aggregate a rocprofv3 rocpd sqlite database into the per-kernel stats CSV kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, header, tail=0.0, windows=None):
    """tail > 0: only the launches that start in the last ``tail`` fraction of the trace's span (e.g. a tool's timed part);
    windows: candidate (start, end) pairs in ns on different clocks -- the one that lies inside the trace's span is used"""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    where = ""
    if tail > 0:
        t0, t1 = list(cur.execute(f"select min(start), max(end) from {kd}"))[0]
        where = "where d.start >= %d " % int(t1 - tail * (t1 - t0))
    if windows:
        t0, t1 = list(cur.execute(f"select min(start), max(end) from {kd}"))[0]
        ok = [(a, b) for a, b in windows if t0 <= a <= b <= t1 + 10**9]
        if not ok:
            raise SystemExit("no window on the tracer's clock: trace %d..%d, windows %r" % (t0, t1, windows))
        where = "where d.start >= %d and d.start < %d " % ok[0]
        header += "\\nlaunches inside the window %d..%d ns (%.1f ms)" % (ok[0][0], ok[0][1], (ok[0][1] - ok[0][0]) / 1e6)
    rows = list(cur.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
        f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id=s.id {where}"
        f"group by s.kernel_name order by 6 desc"))
    tot = sum(r[5] for r in rows)
    with open(out_path, "w") as f:
        for h in header.split("\\n"):
            f.write("# " + h + "\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPRs,SGPRs,LDS\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%.2f,%d,%d,%s,%s,%s\n' % (r[0], r[1], r[5], r[2], 100 * r[5] / tot, r[3], r[4], r[6], r[7], r[8]))
    for r in rows[:12]:
        print("%-50s n=%6d avg=%9.2f us  %5.1f%%" % (r[0][:50], r[1], r[2] / 1e3, 100 * r[5] / tot))


if __name__ == "__main__":
    tail_, win_ = 0.0, None
    if len(sys.argv) > 4:
        if "," in sys.argv[4]:  # "a,b a,b ..." : a CHUNK_WINDOW_NS line's pairs
            win_ = [tuple(int(x) for x in w.split(",")) for w in sys.argv[4].split()]
        else:
            tail_ = float(sys.argv[4])
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "", tail_, win_)

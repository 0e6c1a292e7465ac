#!/usr/bin/env python3
"""Timeline figures of a rocprofv3 kernel trace (rocpd sqlite): span, time with at least one kernel running, idle gaps, and
a text dump of the launches of a few steps around the middle (or behind the given fraction of the trace, from the first launch
of a named kernel there).   python tools/rocprof_timeline.py run.db [n_dump [fraction [kernel]]]"""
import sqlite3
import sys


def main(db_path, n_dump=60, frac=0.5, anchor=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute(f"select d.start, d.end, s.kernel_name, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    rows = rows[len(rows) // 3:]  # skip set-up and warm-up
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end, gaps = 0, rows[0][0], []
    for s, e, _, _ in rows:
        if s > cur_end:
            gaps.append(s - cur_end)
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    print("span %.1f ms, some kernel running %.1f ms (%.0f%%), sum of kernel durations %.1f ms" % (
        (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(r[1] - r[0] for r in rows) / 1e6))
    gaps.sort()
    if gaps:
        print("idle gaps: n=%d total %.1f ms, median %.1f us, p90 %.1f us, max %.1f us" % (
            len(gaps), sum(gaps) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
    mid = int(len(rows) * frac)
    if anchor:  # the dump starts a few launches in front of the first launch of that kernel behind the given place
        for i in range(mid, len(rows)):
            if anchor in rows[i][2]:
                mid = max(i - 8, 0)
                break
    base = rows[mid][0]
    for s, e, n, q in rows[mid:mid + n_dump]:
        short = n.split("(")[0]
        short = short[short.find("k_"):][:28] if "k_" in short else short[:28]
        print("%9.1f %8.1f  q%-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, q, short))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60, float(sys.argv[3]) if len(sys.argv) > 3 else 0.5,
         sys.argv[4] if len(sys.argv) > 4 else None)

#!/usr/bin/env python3
"""Timeline of a window of kernel dispatches from a rocprofv3 rocpd sqlite database (start/end in us relative to the first
dispatch shown): which launches overlap, where the gaps are.  usage: rocprof_timeline.py run.db first_index count"""
import sqlite3
import sys


def main(db_path, first, count):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.{q} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    rows = rows[first:first + count]
    t0 = rows[0][1]
    for name, st, en, qid in rows:
        short = name.split("(")[0].replace("_Z", "")[:28]
        print("%-28s q=%-4s start %9.1f  end %9.1f  dur %8.1f" % (short, qid, (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))

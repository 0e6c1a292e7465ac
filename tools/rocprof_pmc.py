#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd sqlite database (one --pmc pass)."""
import json
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    info = [t for t in tabs if "info_pmc" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({pmc})")]
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value), sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, i.name")
    out = {}
    for name, ctr, n, avg, tot in cur.execute(q):
        out.setdefault(name, {})[ctr] = {"launches": n, "avg": avg, "sum": tot}
    for name in sorted(out, key=lambda k: -max(v["sum"] for v in out[k].values()))[:8]:
        print(name[:60], {c: round(v["avg"], 1) for c, v in out[name].items()}, "n=%d" % list(out[name].values())[0]["launches"])
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])

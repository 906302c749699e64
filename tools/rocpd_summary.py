#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (ROCm 7.2 default output of `-o name`) into the small CSV
summaries that are committed under profiles/.

    python tools/rocpd_summary.py stats  <results.db>                  > profiles/<round>_kernel_stats.csv
    python tools/rocpd_summary.py pmc    <fetch.db> [<write.db> ...]   > profiles/<round>_pmc.csv

`stats`: one row per kernel name: calls, total / average / min / max duration (ns), share.
`pmc`  : one row per kernel name per counter: calls, average counter value per dispatch and the
         average dispatch duration of that (profiled) pass.  FETCH_SIZE / WRITE_SIZE are in KiB as
         rocprofv3 reports them; the gfx950 corrections (MI355X_MICROARCH.md, HBM section: double
         FETCH_SIZE for 16-B-per-lane streaming reads) are applied by the reader, not here.
"""
import csv
import sqlite3
import sys


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


def stats(db: str, out):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([short(name), calls, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])


def pmc(dbs, out):
    w = csv.writer(out)
    w.writerow(["Name", "Counter", "Calls", "AvgValuePerDispatch", "AvgDurationNs", "GridX", "GridY", "GridZ"])
    for db in dbs:
        con = sqlite3.connect(db)
        rows = con.execute(
            "select kernel_name, counter_name, count(*), avg(value), avg(duration), "
            "avg(grid_size_x), avg(grid_size_y), avg(grid_size_z) "
            "from counters_collection group by kernel_name, counter_name order by sum(duration) desc").fetchall()
        for name, ctr, calls, val, dur, gx, gy, gz in rows:
            w.writerow([short(name), ctr, calls, round(val, 3), round(dur, 1), int(gx), int(gy), int(gz)])


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.stdout)
    elif mode == "pmc":
        pmc(sys.argv[2:], sys.stdout)
    else:
        raise SystemExit(__doc__)

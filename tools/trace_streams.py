"""Per-stream busy segments of the last `window_ms` of a rocprofv3 kernel trace (rocpd .db): shows what overlapped."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); win = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
rows = db.execute("select name,start,end,duration,queue_id,stream_id from kernels order by start").fetchall()
tend = rows[-1][2]; rows = [r for r in rows if r[1] >= tend - win * 1e6]; T0 = rows[0][1]
seg = collections.defaultdict(list)
for name, s, e, d, q, st in rows:
    L = seg[(st, q)]
    if L and s - L[-1][1] < 200e3:
        L[-1][1] = max(e, L[-1][1]); L[-1][2] += 1; L[-1][3] += d
    else:
        L.append([s, e, 1, d, name.split("(")[0][-40:]])
for k, L in sorted(seg.items()):
    print("stream,queue", k)
    for a, b, n, busy, first in L:
        print(f"   {(a - T0) / 1e6:8.2f} -> {(b - T0) / 1e6:8.2f} ms  {n:5d} kernels, busy {busy / 1e6:7.2f} ms   first: {first}")

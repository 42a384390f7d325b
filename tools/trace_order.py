"""Print the kernel sequence of the last few steps from a rocprofv3 --kernel-trace database (which launches -- copies and
fills included -- a step really consists of, in stream order, with durations and the gaps between them).
    python tools/trace_order.py <dir with *_results.db> [n_rows]"""
import glob, os, sqlite3, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))[0]
c = sqlite3.connect(f)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
rows = list(c.execute("select name, start, end from %s order by start" % view[0]))
rows = rows[-n:]
prev = None
for name, s, e in rows:
    print("%-60s %8.2f us   gap %6.2f us" % (name[:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e

"""Summarise a rocprofv3 --pmc results .db: per kernel name, mean of each counter over dispatches."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("no counters view; tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]); sys.exit()
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
rows = cur.execute(f"select kernel_name, counter_name, value from {view}") if "kernel_name" in cols else None
if rows is None:
    print(cols); sys.exit()
acc = defaultdict(lambda: defaultdict(list))
for k, c, v in rows:
    k = re.sub(r"\(.*", "", k).replace("void ", "").replace("mst::", "")[:40]
    acc[k][c].append(v)
names = sorted({c for k in acc for c in acc[k]})
print("| kernel | n | " + " | ".join(names) + " |")
for k in sorted(acc, key=lambda k: -sum(acc[k][names[0]])):
    n = len(acc[k][names[0]])
    print(f"| {k} | {n} | " + " | ".join(f"{sum(acc[k][c]) / max(len(acc[k][c]), 1):.3g}" for c in names) + " |")

"""Sum a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE results .db over the dispatches of the timed steps."""
import json, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
counter = sys.argv[2]
steps_total = int(sys.argv[3])          # warmup + steps + 1 stamped step
rows = list(cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"))
per = {}
tot = 0.0
for k, c, v, d in rows:
    if c != counter: continue
    if not ("mst::" in k): continue      # hot-path kernels only (torch fills / copies are negligible)
    n = re.sub(r"\(.*", "", k).replace("void ", "").replace("mst::", "")
    per[n] = per.get(n, 0.0) + v
    tot += v
print(json.dumps({"counter": counter, "total": tot, "per_step": tot / steps_total,
                  "per_kernel_per_step": {k: v / steps_total for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]}}))

"""FETCH_SIZE / WRITE_SIZE per known byte (tools/ubench/hbm_counters.hip): python tools/hbm_calib.py <fetch.db> <write.db> [<trace.db>]"""
import json, re, sqlite3, sys
import os
BYTES = float(int(os.environ.get('CALIB_MIB', '1024')) << 20)
out = {}
for path, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    cur = sqlite3.connect(path).cursor()
    acc = {}
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c != counter: continue
        n = re.sub(r"\(.*", "", k)
        acc.setdefault(n, []).append(v)
    for n, vs in acc.items():
        vs = vs[1:] if len(vs) > 1 else vs  # first repetition: cold
        out.setdefault(n, {})[counter + "_KB"] = sum(vs) / len(vs)
        out[n][counter + "_per_true_byte"] = sum(vs) / len(vs) * 1024.0 / BYTES
if len(sys.argv) > 3:
    cur = sqlite3.connect(sys.argv[3]).cursor()
    acc = {}
    for name, s, e in cur.execute("select name, start, end from kernels"):
        acc.setdefault(re.sub(r"\(.*", "", name), []).append((e - s) / 1e3)
    for n, d in acc.items():
        if n in out and len(d) > 1:
            out[n]["us"] = sum(d[1:]) / (len(d) - 1)
            out[n]["true_GBs"] = BYTES / (out[n]["us"] * 1e-6) / 1e9
print(json.dumps(out, indent=1))

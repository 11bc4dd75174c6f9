"""Start / end of every kernel of ONE mid-run step (relative us) from a rocprofv3 --kernel-trace results .db - shows overlap across streams.
usage: python tools/ktimeline.py <results.db> [steps-from-end=6]"""
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_stft3_fwd" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
i0, i1 = idx[-back], idx[-back + 1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    n = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("mst::", "")
    print(f"{n[:36]:36s} q {str(r[3])[-6:]:>6s}  start {(r[1] - t0) / 1e3:8.1f}  end {(r[2] - t0) / 1e3:8.1f}  dur {(r[2] - r[1]) / 1e3:7.1f}")
print(f"step wall {(rows[i1][1] - t0) / 1e3:.1f} us")

"""Condense a rocprofv3 results .db (kernel trace) into a short per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("mst::", "")
    if "at::native" in name or "elementwise" in name:
        name = "torch:" + re.sub(r".*native::", "", name)[:40]
    return name[:64]


def main(db_path, steps=None):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        c, t, p = agg.get(k, (0, 0.0, 0.0))
        agg[k] = (c + calls, t + total, p + pct)
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for k, (c, t, p) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {c} | {t:.1f} | {t / c:.2f} | {p:.2f} |")
    tot = sum(v[1] for v in agg.values())
    print(f"\ntotal kernel time {tot:.1f} us" + (f"; per step {tot / steps:.1f} us over {steps} steps" if steps else ""))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)

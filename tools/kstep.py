"""Per-kernel listing of ONE mid-run step from a rocprofv3 --kernel-trace results .db.
usage: python tools/kstep.py <results.db> [anchor-kernel-substring=k_prep(] [steps-from-end=8]"""
import re
import sqlite3
import sys


def main(path, anchor="k_prep(", back=8):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x, vgpr_count, lds_size from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    i0, i1 = idx[-back], idx[-back + 1]
    print(f"step wall {(rows[i1][1] - rows[i0][1]) / 1e3:.1f} us")
    tot = 0.0
    for r in rows[i0:i1]:
        n = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("mst::", "")
        d = (r[2] - r[1]) / 1e3
        tot += d
        print(f"{n[:34]:34s} {d:7.1f} us  grid {r[3] // max(r[5], 1)}x{r[4]} wg {r[5]} vgpr {r[6]} lds {r[7]}")
    print(f"sum of kernel durations {tot:.1f} us")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_prep(", int(a[3]) if len(a) > 3 else 8)

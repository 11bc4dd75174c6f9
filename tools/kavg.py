"""Average duration per kernel from rocprofv3 --kernel-trace results .db files: python tools/kavg.py <pattern> <db>..."""
import re, sqlite3, sys
from collections import defaultdict
pat = sys.argv[1]
for path in sys.argv[2:]:
    cur = sqlite3.connect(path).cursor()
    acc = defaultdict(list)
    for name, s, e, vg, lds, gx, wx in cur.execute("select name, start, end, vgpr_count, lds_size, grid_x, workgroup_x from kernels"):
        n = re.sub(r"\(.*", "", name).replace("void ", "").replace("mst::", "")
        acc[(n, vg, lds, gx // max(wx, 1), wx)].append((e - s) / 1e3)
    print("==", path.split("/")[-2])
    for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if re.search(pat, k[0]):
            d = d[3:] if len(d) > 3 else d
            print(f"  {k[0][:44]:44s} vgpr {k[1]:4d} lds {k[2]:6d} grid {k[3]:5d} x{k[4]:4d}  avg {sum(d)/len(d):7.1f} us")

"""Per-kernel averages of the controller kernels (csrc/mst_ctrl.hip) in one cfg #5 step of the trace written by tools/cfg5_prof.sh."""
import csv, collections, sys
f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cfg5_prof/r_kernel_trace.csv"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
spec = [i for i, r in enumerate(rows) if "k_spectrogram" in r["Kernel_Name"]]
st = rows[spec[-4]:spec[-2]]
d = collections.defaultdict(list)
for r in st:
    n = r["Kernel_Name"]
    if "ctrl::" in n:
        d[n.split("ctrl::")[1].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} n={len(v):4d} avg {sum(v) / len(v):7.1f} us  min {min(v):6.1f} max {max(v):6.1f} total {sum(v):8.1f}")
    tot += sum(v)
print(f"controller kernels: {tot / 1e3:.2f} ms per step")

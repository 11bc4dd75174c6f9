"""Idle-gap attribution of one cfg #5 step from the rocprofv3 kernel trace of tools/cfg5_prof.sh: python tools/cfg5_gaps.py [trace.csv]"""
import csv, collections, sys
f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cfg5_prof/r_kernel_trace.csv"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
spec = [i for i, r in enumerate(rows) if "k_spectrogram" in r["Kernel_Name"]]
st = rows[spec[-4]:spec[-2]]
t0, t1 = int(st[0]["Start_Timestamp"]), int(st[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
lib = sum(1 for r in st if "mst::" in r["Kernel_Name"])
print(f"one step: {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, {len(st)} kernels ({lib} of the library)")
gap, cnt, dur = collections.defaultdict(float), collections.Counter(), collections.defaultdict(float)
prev_end, prev = int(st[0]["End_Timestamp"]), "-"
big = []
for r in st[1:]:
    g = max(0, int(r["Start_Timestamp"]) - prev_end)
    fam = r["Kernel_Name"].replace("void ", "").replace("mst::", "").split("<")[0].split("(")[0][:44]
    gap[fam] += g; cnt[fam] += 1; dur[fam] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if g > 30000: big.append((g / 1e3, prev, fam))
    prev_end = max(prev_end, int(r["End_Timestamp"])); prev = fam
print(f"idle {sum(gap.values()) / 1e6:.2f} ms; gap in front of:")
for k, v in sorted(gap.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {k:46s} n={cnt[k]:4d} gap {v / 1e3:8.1f} us  busy {dur[k] / 1e3:8.1f} us")
print("gaps > 30 us (us, after, before):")
for b in sorted(big, reverse=True)[:25]: print("  %.0f  %s -> %s" % b)

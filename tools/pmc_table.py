"""Per-kernel counter table from the passes of tools/pmc_passes.sh.

usage: python tools/pmc_table.py <tag> <steps_executed> [--md out.md] [--json out.json]

Reads gpurun_out/<tag>_{stats,sq1,sq2,fetch,write}/.  Durations come from the un-instrumented kernel-trace
pass (counter passes serialise dispatches and run at a lower clock), counters are means over dispatches.
Derived columns (MI355X: 256 CUs x 4 SIMD-32, 2.4 GHz nominal; SQ_* "cycles" are quad-cycles summed over waves):
  VGPR        = 2 x rocprofv3's vgpr_count (it reports half the allocation of a wave64 kernel, see counters())
  waves/SIMD  = resident waves per SIMD if the whole grid is resident (min with the VGPR / LDS / 8-wave limits)
  VALU issue  = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x duration x 2.4 GHz)   (a wave64 VALU op occupies a SIMD-32 for 2 cycles)
  VALUBusy    = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES-normalised) is NOT used: the gfx94x formula is not valid here
  wait        = SQ_WAIT_ANY / SQ_WAVE_CYCLES   (share of wave lifetime parked on s_waitcnt / s_barrier)
  stall       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  LDS conf    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  GB/s        = (2 x FETCH_SIZE? no: raw FETCH_SIZE + WRITE_SIZE, KB) / duration   (raw counters, see profiles/*_traffic.json)
"""
import csv
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void\s+", "", name).replace("mst::", "")
    if "at::native" in name:
        name = "torch:" + re.sub(r".*native::", "", name)[:30]
    return name[:48]


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    if not os.path.exists(path):
        return acc, meta
    cur = sqlite3.connect(path).cursor()
    q = ("select kernel_name, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, "
         "workgroup_size, grid_size from counters_collection")
    for k, c, v, vg, ag, sg, lds, wg, grid in cur.execute(q):
        k = short(k)
        acc[k][c].append(v)
        # rocprofv3 7.2 reports HALF the per-lane VGPR allocation of wave64 gfx950 kernels (k_stft2_fwd<8192>: 64 here, 128 in
        # hipcc -Rpass-analysis=kernel-resource-usage and in the code-object metadata; same factor for every kernel checked)
        meta[k] = dict(vgpr=2 * vg, agpr=2 * ag, sgpr=sg, lds=lds, wg=wg, grid=grid)
    return acc, meta


def main(tag, steps, md=None, js=None):
    dur = defaultdict(list)
    with open(os.path.join(G, f"{tag}_stats", "r_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    acc, meta = {}, {}
    for part in ("sq1", "sq2", "fetch", "write"):
        a, m = counters(os.path.join(G, f"{tag}_{part}", "r_results.db"))
        for k in a:
            acc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in a[k].items()})
        meta.update(m)
    rows = []
    for k, d in dur.items():
        if k not in acc:
            continue
        c, m = acc[k], meta[k]
        us = sum(d) / len(d)
        per_step = sum(d) / steps
        waves_wg = max(1, m["wg"] // 64)
        n_wg = m["grid"] // max(m["wg"], 1)
        alloc = -(-(m["vgpr"] + m["agpr"]) // 8) * 8
        by_vgpr = min(8, 512 // max(alloc, 8))
        by_lds = (160 * 1024 // m["lds"]) * waves_wg / 4.0 if m["lds"] else 8.0
        resident = min(8.0, by_vgpr, by_lds, n_wg * waves_wg / 1024.0)
        cyc = us * 1e-6 * 2.4e9
        g = lambda n: c.get(n, float("nan"))
        rows.append(dict(
            kernel=k, calls=len(d), us=us, us_per_step=per_step, wgs=n_wg, wg=m["wg"], vgpr=m["vgpr"] + m["agpr"], lds=m["lds"],
            waves_per_simd=resident, valu_insts=g("SQ_INSTS_VALU"), valu_issue=g("SQ_INSTS_VALU") * 2 / (1024 * cyc),
            wait=g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), stall=g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
            lds_insts=g("SQ_INSTS_LDS"), lds_conflict=g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
            vmem_insts=g("SQ_INSTS_VMEM"), fetch_mb=g("FETCH_SIZE") / 1e3, write_mb=g("WRITE_SIZE") / 1e3,
            gbs=(g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / (us * 1e-6) / 1e9,
        ))
    rows.sort(key=lambda r: -r["us_per_step"])
    lines = ["| kernel | calls | avg us | us/step | WGs x lanes | VGPR | LDS B | waves/SIMD | VALU insts (wave) | VALU issue % | wait % | stall % | LDS conflict % | FETCH MB | WRITE MB | raw GB/s |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        if r["us_per_step"] < 0.5:
            continue
        lines.append(
            f"| {r['kernel']} | {r['calls']} | {r['us']:.1f} | {r['us_per_step']:.1f} | {r['wgs']} x {r['wg']} | {r['vgpr']} | {r['lds']} | "
            f"{r['waves_per_simd']:.1f} | {r['valu_insts']:.3g} | {100 * r['valu_issue']:.0f} | {100 * r['wait']:.0f} | {100 * r['stall']:.0f} | "
            f"{100 * r['lds_conflict']:.0f} | {r['fetch_mb']:.1f} | {r['write_mb']:.1f} | {r['gbs']:.0f} |")
    tot_us = sum(r["us_per_step"] for r in rows)
    tot_valu = sum(r["valu_insts"] * r["calls"] for r in rows if r["valu_insts"] == r["valu_insts"]) / steps
    lines.append(f"\nsum of kernel time {tot_us:.1f} us per step; VALU wave-instructions per step {tot_valu:.3g} "
                 f"(= {tot_valu * 64:.3g} lane-instructions; at 2 cycles per wave-instruction on 1024 SIMD-32s: "
                 f"{tot_valu * 2 / 1024 / 2.4e3:.1f} us of pure issue)")
    out = "\n".join(lines)
    print(out)
    if md:
        open(md, "w").write(out + "\n")
    if js:
        json.dump(dict(steps=steps, rows=rows, sum_us_per_step=tot_us, valu_wave_insts_per_step=tot_valu), open(js, "w"), indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    md = a[a.index("--md") + 1] if "--md" in a else None
    js = a[a.index("--json") + 1] if "--json" in a else None
    main(a[0], int(a[1]), md, js)

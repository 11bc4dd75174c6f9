"""Turn the round-2 evidence run (gpurun_out/r2_*: tools/pmc_passes.sh r2 + pytest + bench) into the committed artefacts:

  profiles/round2_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary`
  profiles/round2_agent_info.csv
  profiles/round2_counters.md / .json      per-kernel table: duration, grid, VGPR, LDS, waves/SIMD, VALU issue %, wait %, stall %,
                                           LDS bank-conflict %, FETCH / WRITE (tools/pmc_table.py)
  profiles/round2_traffic.json             HBM bytes per step (raw FETCH_SIZE + WRITE_SIZE) and VALU lane-instructions per step
  profiles/round2_bench.json               the bench line of the same build
  profiles/parity_r02.json                 measured parity numbers of the -m gpu tests

usage: python tools/make_profile_r2.py <steps_executed_in_pmc_runs>     (warmup + steps + 3 stage-split steps = 15)
"""
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counter_sum(db, counter):
    cur = sqlite3.connect(db).cursor()
    tot, per = 0.0, {}
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c != counter or "mst::" not in k:
            continue
        tot += v
        per[k.split("(")[0].replace("void ", "").replace("mst::", "")] = per.get(k.split("(")[0].replace("void ", "").replace("mst::", ""), 0.0) + v
    return tot, per


def main(steps):
    shutil.copy(os.path.join(G, "r2_stats", "r_kernel_stats.csv"), os.path.join(P, "round2_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "r2_stats", "r_agent_info.csv"), os.path.join(P, "round2_agent_info.csv"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), "r2", str(steps), "--md",
                    os.path.join(P, "round2_counters.md"), "--json", os.path.join(P, "round2_counters.json")], check=True)
    f, fper = counter_sum(os.path.join(G, "r2_fetch", "r_results.db"), "FETCH_SIZE")
    w, wper = counter_sum(os.path.join(G, "r2_write", "r_results.db"), "WRITE_SIZE")
    v, _ = counter_sum(os.path.join(G, "r2_sq1", "r_results.db"), "SQ_INSTS_VALU")
    out = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE / the SQ sets of tools/pmc_passes.sh) -- "
                   "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary",
        "steps_counted": steps,
        "fetch_kb_per_step": f / steps, "write_kb_per_step": w / steps,
        "hbm_bytes_per_step_raw": (f + w) * 1024.0 / steps,
        "algorithmic_bytes_per_step": 218103808,
        "valu_wave_instructions_per_step": v / steps,
        "valu_lane_instructions_per_step": 64.0 * v / steps,
        "note": "raw FETCH_SIZE / WRITE_SIZE (KB) summed over the mst:: kernels of the run and divided by the steps executed (the one "
                "no-grad console forward that builds the reference mix is included).  The x2 read correction of the MI355X guide "
                "applies to fully coalesced 16-B-per-lane streams only and is NOT applied (round-1 calibration: 32-B-per-lane streams "
                "count ~1.0x, the 64-B-per-lane-row slab pattern of the EQ kernels ~0.65-0.8x).",
        "per_kernel_fetch_kb_per_step": {k: x / steps for k, x in sorted(fper.items(), key=lambda kv: -kv[1])[:14]},
        "per_kernel_write_kb_per_step": {k: x / steps for k, x in sorted(wper.items(), key=lambda kv: -kv[1])[:14]},
    }
    json.dump(out, open(os.path.join(P, "round2_traffic.json"), "w"), indent=1)
    print(f"FETCH {f / steps / 1e3:.0f} MB + WRITE {w / steps / 1e3:.0f} MB = {(f + w) * 1024 / steps / 1e9:.3f} GB per step; "
          f"VALU {64 * v / steps:.3g} lane-instructions per step")
    for name in ("parity_r02.json",):
        if os.path.exists(os.path.join(G, name)):
            shutil.copy(os.path.join(G, name), os.path.join(P, name))
    log = os.path.join(G, "r2_bench.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                json.dump(json.loads(line), open(os.path.join(P, "round2_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 15)

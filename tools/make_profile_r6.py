"""Turn the round-6 evidence run (gpurun_out/r6_*: tools/pmc_passes.sh r5 + pytest + bench) into the committed artefacts:

  profiles/round6_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph`
  profiles/round6_agent_info.csv
  profiles/round6_counters.md / .json      per-kernel table: duration, grid, VGPR, LDS, waves/SIMD, VALU issue %, wait %, stall %,
                                           LDS bank-conflict %, FETCH / WRITE (tools/pmc_table.py)
  profiles/round6_traffic.json             HBM bytes per step (raw FETCH_SIZE + WRITE_SIZE) and VALU lane-instructions per step
  profiles/round6_bench.json               the bench line of the same build
  profiles/parity_r06.json                 measured parity numbers of the -m gpu tests

usage: python tools/make_profile_r6.py <steps_executed_in_pmc_runs>     (warmup + steps + 3 stage-split steps = 15)
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
    shutil.copy(os.path.join(G, "r6_stats", "r_kernel_stats.csv"), os.path.join(P, "round6_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "r6_stats", "r_agent_info.csv"), os.path.join(P, "round6_agent_info.csv"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_table.py"), "r6", str(steps), "--md",
                    os.path.join(P, "round6_counters.md"), "--json", os.path.join(P, "round6_counters.json")], check=True)
    f, fper = counter_sum(os.path.join(G, "r6_fetch", "r_results.db"), "FETCH_SIZE")
    w, wper = counter_sum(os.path.join(G, "r6_write", "r_results.db"), "WRITE_SIZE")
    v, _ = counter_sum(os.path.join(G, "r6_sq1", "r_results.db"), "SQ_INSTS_VALU")
    out = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE / the SQ sets of tools/pmc_passes.sh) -- "
                   "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph",
        "steps_counted": steps,
        "fetch_kb_per_step": f / steps, "write_kb_per_step": w / steps,
        "hbm_bytes_per_step_raw": (f + w) * 1024.0 / steps,
        # corrected with the known-bytes calibration of profiles/round4_hbm_calibration.md: FETCH_SIZE counts half of the bytes read in
        # every access pattern of these kernels, WRITE_SIZE all bytes written; Infinity-Cache hits are counted (an upper bound on HBM)
        "hbm_bytes_per_step": (2.0 * f + w) * 1024.0 / steps,
        "algorithmic_bytes_per_step": 218103808,
        "valu_wave_instructions_per_step": v / steps,
        "valu_lane_instructions_per_step": 64.0 * v / steps,
        "note": "FETCH_SIZE / WRITE_SIZE (KB) summed over the mst:: kernels of the run and divided by the steps executed (the one no-grad "
                "console forward that builds the reference mix is included).  hbm_bytes_per_step = 2 x FETCH + WRITE (calibrated: "
                "profiles/round4_hbm_calibration.md); rounds 1-3 reported the raw sum (round 3 corrected: 1.79 GB).",
        "per_kernel_fetch_kb_per_step": {k: x / steps for k, x in sorted(fper.items(), key=lambda kv: -kv[1])[:14]},
        "per_kernel_write_kb_per_step": {k: x / steps for k, x in sorted(wper.items(), key=lambda kv: -kv[1])[:14]},
    }
    json.dump(out, open(os.path.join(P, "round6_traffic.json"), "w"), indent=1)
    print(f"FETCH {f / steps / 1e3:.0f} MB + WRITE {w / steps / 1e3:.0f} MB = {(f + w) * 1024 / steps / 1e9:.3f} GB per step; "
          f"VALU {64 * v / steps:.3g} lane-instructions per step")
    for name in ("parity_r06.json",):
        if os.path.exists(os.path.join(G, name)):
            shutil.copy(os.path.join(G, name), os.path.join(P, name))
    log = os.path.join(G, "r6_bench.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                json.dump(json.loads(line), open(os.path.join(P, "round6_bench.json"), "w"), indent=1)


def round2_us():  # (kept name: the comparison column now holds ROUND 3)
    """us per step of the round-2 kernels (profiles/round2_counters.json) for the comparison column"""
    try:
        rows = json.load(open(os.path.join(P, "round5_counters.json")))
        rows = rows["rows"] if isinstance(rows, dict) else rows
        return {k["kernel"]: k["us_per_step"] for k in rows}
    except (OSError, KeyError, ValueError):
        return {}


# kernels that replaced a round-2 kernel of another name
SAME_AS_ROUND2 = {}  # (round 6: the kernel names of the comparison round are this round's)


def summary():
    b = json.load(open(os.path.join(P, "round6_bench.json")))
    t = json.load(open(os.path.join(P, "round6_traffic.json")))
    rows = json.load(open(os.path.join(P, "round6_counters.json")))
    rows = rows["rows"] if isinstance(rows, dict) else rows
    R2 = round2_us()
    r = b["roofline"]
    out = ["# Round 6 profile - `python bench.py` on one MI355X (final build of the round)", "",
           "Made by `tools/round_end_r6.sh main` on the GPU box (`smoke()`, `tools/pmc_passes.sh r6`, `bench.py`; the `-m gpu` suite ran separately: 132 passed) and "
           "`tools/make_profile_r6.py` here.  Raw: `round6_bench_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of "
           "`bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-graph`), `round6_counters.md/json` (separate --pmc passes), "
           "`round6_traffic.json`, `round6_bench.json` (the un-profiled bench line), `parity_r06.json` (measured parity numbers).", "",
           f"**Bench line: {b['value']:.0f} mixes/s, {b['ms_per_step']:.4f} ms per step** (median of per-step HIP events "
           f"{r['gpu_ms_per_step_median']:.4f} ms), {r['achieved']:.0f} GB/s algorithmic = {100 * r['frac']:.2f} % of 8 TB/s; L2 <-> fabric "
           f"traffic (2 x FETCH_SIZE + WRITE_SIZE, calibrated; Infinity-Cache hits included) {t['hbm_bytes_per_step'] / 1e9:.3f} GB per step = "
           f"{t['hbm_bytes_per_step'] / t['algorithmic_bytes_per_step']:.1f}x the algorithmic bytes (raw sum {t['hbm_bytes_per_step_raw'] / 1e9:.3f} GB; "
           f"round 5: 1.554 GB corrected); VALU {t['valu_lane_instructions_per_step']:.3g} lane-instructions per step "
           f"(issue share {100 * (r['valu'].get('issue_frac') or 0):.0f} %).  Driver-measured earlier rounds: round 5 16988 mixes/s, 0.4709 ms; round 4 15041, 0.532; round 3 13066, 0.612; round 2 12363; round 1 9938.", "",
           "Stages (ms): " + ", ".join(f"{k.replace('_ms', '').replace('_', ' ')} {v:.3f}" for k, v in r["stages"].items()) + ".", "",
           "| secondary line | ms per step (median) | mixes/s (encoder line: signals/s) |", "|---|---:|---:|"]
    for s_ in b.get("secondary", []):
        rate = s_.get("mixes_per_s", s_.get("signals_per_s", 0.0))
        peak_name = {"bf16": "bf16 MFMA peak", "bf16x3": "bf16 MFMA peak / 3 (three matrix instructions per block)",
                     "bf16x6": "bf16 MFMA peak / 6 (six matrix instructions per block)"}.get(s_.get("precision"), "fp32 MFMA peak")
        extra = (f" ({s_['conv_TFLOPs_per_s']:.0f} TFLOP/s of convolution = {100 * s_['frac_of_dense_mfma_peak_of_that_dtype']:.1f} % of the dense "
                 f"{peak_name})") if "conv_TFLOPs_per_s" in s_ else ""
        if "hipgraph_replay" in s_:
            extra += f" (replayed as one hipGraph: {s_['hipgraph_replay']['ms_per_step_median']:.4f} ms)"
        out.append(f"| {s_['workload'][:150]}{' [' + (s_.get('precision') or s_.get('encoder_precision')) + ']' if (s_.get('precision') or s_.get('encoder_precision')) else ''}{extra} | {s_['ms_per_step_median']:.3f} | {rate:.0f} |")
    cb = b.get("cpu_baseline") or {}
    if cb:
        out += ["", f"CPU baseline (oracle, `kind: {cb.get('kind')}`): {cb.get('value'):.2f} mixes/s on {cb.get('cores')} threads; by thread count: "
                + ", ".join(f"{k}: {v['value']:.2f}" for k, v in (cb.get("by_threads") or {}).items()) + "."]
    traced = sum(k["us_per_step"] for k in rows)
    scale = 1e3 * r["gpu_ms_per_step"] / traced if traced else 1.0
    out += ["", f"**Trace inflation.**  The per-kernel durations below come from `rocprofv3 --kernel-trace`, under which every kernel runs longer "
            f"(lower clocks, per-dispatch instrumentation): they add up to {traced:.1f} us per step, the un-profiled step (HIP events on the launch "
            f"stream, same build, same box class) is {1e3 * r['gpu_ms_per_step']:.1f} us.  The second column scales every kernel by that ratio "
            f"({scale:.3f}) - the un-inflated estimate to use against the step time; kernel-to-kernel proportions are the trace's."]
    out += ["", "| kernel | us per step (traced) | us per step (un-inflated) | round 5 (traced) | VGPR | waves/SIMD | VALU issue % | wait % |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for k in rows[:26]:
        name = k["kernel"]
        r1 = R2.get(SAME_AS_ROUND2.get(name, name))
        r1 = None if r1 is None else round(r1, 1)
        out.append(f"| {name} | {k['us_per_step']:.1f} | {k['us_per_step'] * scale:.1f} | {'' if r1 is None else r1} | {k['vgpr']} | {k['waves_per_simd']:.1f} | "
                   f"{100 * k['valu_issue']:.0f} | {100 * k['wait']:.0f} |")
    out += ["", "VGPR = 2 x rocprofv3's `vgpr_count` (it reports half the wave64 allocation; checked against "
            "`-Rpass-analysis=kernel-resource-usage`).  What the numbers mean and what was done with them: DESIGN.md sections 8-12.  "]
    open(os.path.join(P, "round6_summary.md"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 15)
    summary()

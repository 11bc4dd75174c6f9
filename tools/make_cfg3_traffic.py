"""profiles/round6_cfg3_traffic.json + round6_cfg3_kernels.txt from `tools/round_end_r6.sh cfg3` (gpurun_out/cfg3_{FETCH,WRITE}_SIZE.json, cfg3_prof.log)."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
f = json.load(open(os.path.join(G, "cfg3_FETCH_SIZE.json"))); w = json.load(open(os.path.join(G, "cfg3_WRITE_SIZE.json")))
alg = 32 * (8 * 262144 * (16 + 2) + 24 * 262144)
ms = [l.strip() for l in open(os.path.join(G, "cfg3_prof.log")) if "cfg3:" in l]
try:
    sec = [s for s in json.load(open(os.path.join(P, "round6_bench.json")))["secondary"] if s["workload"].startswith("cfg #3")][0]["ms_per_step_median"]
except Exception:
    sec = None
raw, cor = (f["per_step"] + w["per_step"]) * 1024, (2 * f["per_step"] + w["per_step"]) * 1024
out = {"command": "tools/round_end_r6.sh cfg3: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/cfg3_bench.py",
       "workload": "cfg #3: AdvancedMixConsole 16 tracks x 262144, batch 32, AudioFeatureLoss, lean console; " + (ms[-1] if ms else "") +
                   " under the tracer" + (f", {sec:.3f} ms un-profiled (profiles/round6_bench.json)" if sec else ""),
       "steps_counted": 8, "fetch_kb_per_step": f["per_step"], "write_kb_per_step": w["per_step"],
       "hbm_bytes_per_step_raw": raw, "hbm_bytes_per_step": cor, "algorithmic_bytes_per_step": alg,
       "ratio_raw_over_algorithmic": raw / alg, "ratio_corrected_over_algorithmic": cor / alg,
       "round3_raw": {"hbm_bytes_per_step_raw": 8309447060.0, "ratio_raw_over_algorithmic": 5.896},
       "note": "corrected = 2 x FETCH_SIZE + WRITE_SIZE (profiles/round4_hbm_calibration.md); the one no-grad console forward that builds the reference mix is "
               "included in the 8 steps' sums",
       "per_kernel_fetch_kb_per_step": f["per_kernel_per_step"], "per_kernel_write_kb_per_step": w["per_kernel_per_step"]}
json.dump(out, open(os.path.join(P, "round6_cfg3_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(G, "r6_cfg3_prof.txt"), os.path.join(P, "round6_cfg3_kernels.txt"))
print(out["workload"], f"raw {raw / 1e9:.2f} GB = {raw / alg:.2f}x, corrected {cor / 1e9:.2f} GB = {cor / alg:.2f}x")

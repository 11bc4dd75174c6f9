"""profiles/round3_encoder_counters.md from the evidence run (gpurun_out/r3_encoder_prof.txt, r3_encoder_pmc.txt, r3_bench.log):
python tools/make_encoder_counters.py"""
import ast, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
prof = [l.rstrip("\n") for l in open(os.path.join(G, "r3_encoder_prof.txt")) if not l.startswith(("W2026", "E2026"))]
pmc = {}
for l in open(os.path.join(G, "r3_encoder_pmc.txt")):
    m = re.match(r"(pmc[12]) (.*?) (\{.*\})\s*$", l)
    if m:
        pmc.setdefault(m.group(2).strip(), {}).update({k: float(v) for k, v in ast.literal_eval(m.group(3)).items()})
bench = None
for l in open(os.path.join(G, "r3_bench.log")):
    if l.startswith("{"):
        bench = json.loads(l)
enc = next(s for s in bench["secondary"] if s["workload"].startswith("SpectrogramEncoder"))
out = ["# Round 3 - spectrogram encoder (SpectrogramEncoder + Cnn14) on one MI355X: kernel table and MFMA / LDS counters", "",
       "Made by `tools/round_end.sh` + `tools/make_encoder_counters.py` (`tools/encoder_prof.sh 16 262144 bf16`: rocprofv3 --kernel-trace --stats of "
       "`tools/encoder_bench.py` - 2 + 3 forward passes and 2 + 3 forward+backward passes of 16 signals x 262144 samples, bf16 operands; "
       "`tools/encoder_pmc.sh`: two separate --pmc passes of the same script with 2 timed iterations).  bench.py's secondary line (34 signals): "
       f"{enc['ms_per_step_median']:.1f} ms per forward+backward = {enc['conv_TFLOPs_per_s']:.0f} TFLOP/s of convolution = "
       f"{100 * enc['frac_of_dense_bf16_mfma_peak']:.1f} % of the 2.5 PFLOP/s dense bf16 MFMA peak (mid-round, register-staged kernels: 56.0 ms, 9.1 %).", "", "```"]
out += [l for l in prof if l.startswith("encoder ")]
out += ["```", "", "| kernel (all launches of the run) | calls | avg us | share of GPU time |", "|---|---:|---:|---:|"]
for l in prof:
    m = re.match(r"(.*?)\s+calls\s+(\d+) avg\s+([0-9.]+) us\s+([0-9.]+) %", l)
    if m:
        out.append(f"| `{m.group(1).replace('void ', '').replace('mst::', '')[:64]}` | {m.group(2)} | {m.group(3)} | {m.group(4)} % |")
out += ["", "Counters (summed over every launch of the kernel in the counter run).  MFMA busy = `SQ_VALU_MFMA_BUSY_CYCLES` / (1024 SIMDs x "
        "`GRBM_GUI_ACTIVE` / 8 XCDs): the share of the kernel's SIMD-cycles in which the matrix pipe works - the dense bf16 peak corresponds to "
        "100 %.  `SQ_INSTS_VALU_MFMA_MOPS_BF16` / `SQ_INSTS_MFMA` = 32 (512-flop units of a 16x16x32 instruction) checks the instruction mix.", "",
        "| kernel | MFMA instructions | MFMA busy | other VALU instructions per MFMA | LDS instructions per MFMA | LDS bank-conflict cycles | wave-cycles waiting on an operand |",
        "|---|---:|---:|---:|---:|---:|---:|"]
for k, d in sorted(pmc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0)):
    n = d.get("SQ_INSTS_MFMA", 0)
    if n <= 0 or "SQ_WAVE_CYCLES" not in d:
        continue
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
    out.append(f"| `{k}` | {n:.3g} | {100 * busy:.1f} % | {(d['SQ_INSTS_VALU'] - n) / n:.1f} | {d['SQ_INSTS_LDS'] / n:.2f} | {d['SQ_LDS_BANK_CONFLICT']:.3g} | "
               f"{100 * d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.0f} % |")
out += ["", "Reading: with LDS-direct loads (DESIGN.md section 9.3) the large-layer kernels keep the matrix pipe busy 29-37 % of the time "
        "(register-staged predecessors: 12-21 %, weight gradient 10-17 %), no kernel has LDS bank conflicts, and the weight gradient needs ~1 LDS "
        "instruction per MFMA (transposing 8-byte reads; the 2-byte gathers needed 4.5-9).  What is left is the L2 -> LDS fill rate (~20 B/clk/CU) on the "
        "64-channel layers and the BatchNorm / ReLU / pool passes, which run at HBM speed and are ~30 % of the step."]
open(os.path.join(ROOT, "profiles", "round3_encoder_counters.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[-14:]))

"""Format the raw outputs of tools/final_profile.sh (gpurun_out/final/) into the committed profiles/r01_*.txt files."""
import json
import os

F = "gpurun_out/final"
P = "profiles"


def lines(name):
    path = os.path.join(F, name)
    return open(path).read().splitlines() if os.path.exists(path) else []


line = lines("bench_line.json")[-1]
open(os.path.join(P, "r01_bench_line.json"), "w").write(line + "\n")
st2 = json.loads(lines("bench_under_rocprof.json")[-1])
s2 = st2["roofline"]["stage_ms"]
out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2   (cfg3, 1x MI355X, round 1 final; tools/final_profile.sh)",
       "# bench.py HIP-event durations in the same run: mtfft %.3f ms, fused (+combine) %.3f ms, 2x measure %.3f ms; step %.2f ms"
       % (s2["mtfft_fused"], s2["fused_csm_absim"], s2["measure_epilogue"], st2["ms_per_step"]),
       "# (the fused stage of bench.py = fused_csm_absim_kernel + fused_combine_kernel; averages below include the 2 warm-up launches)"]
out += lines("kernel_stats.txt")[:6]
open(os.path.join(P, "r01_bench_kernel_stats.txt"), "w").write("\n".join(out) + "\n")

pm = lines("pmc_sq.txt")
vals = {}
for l in pm:
    parts = l.split()
    if len(parts) >= 4 and "fused_csm" in parts[0]:
        vals[parts[1]] = float(parts[-1].split("=")[1])
busy, valu, mf = vals.get("SQ_BUSY_CYCLES", 1), vals.get("SQ_INSTS_VALU", 0), vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
hdr = ["# rocprofv3 --kernel-trace --pmc (two passes of 8 SQ counters), cfg3, 1x MI355X, round 1 final kernels (tools/final_profile.sh)",
       "# averages per dispatch PER SHADER ENGINE (32 SEs, 8 CUs = 32 SIMDs each).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles",
       "# (a wave64 VALU instruction = 1 unit = 4 cycles of its SIMD); SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_* are cycles.",
       "# fused kernel: VALU busy = %.3g x 4 / 32 SIMDs = %.3g of %.3g busy cycles = %.0f %%; matrix pipe = %.4g / 32 SIMDs = %.3g cycles = %.0f %%"
       % (valu, valu * 4 / 32, busy, 100 * valu * 4 / 32 / busy, mf, mf / 32, 100 * mf / 32 / busy),
       "#   (= 216 x 16 + 80 x 32 MFMA cycles per chunk and SIMD x 772 chunks, as designed); busy cycles / kernel time = effective clock under this load."]
open(os.path.join(P, "r01_pmc_fused_mtfft.txt"), "w").write("\n".join(hdr + pm) + "\n")

ab = ["# Ablations of the two hot kernels with parts switched off by a debug mask (results WRONG when set), cfg3, 1x MI355X, round 1 final.",
      "# tools/fused_ablation.py (SC_FUSED_DEBUG: 1 = CSM waves skip their MFMAs, 2 = abs waves skip theirs, 8 = no HBM loads after chunk 0)"]
ab += lines("fused_ablation.txt")
ab += ["# tools/mtfft_ablation.py (SC_MTFFT_DEBUG: 1 = no HBM stores, 2 = skip both radix-16 passes, 4 = skip the split/store loop)"]
ab += lines("mtfft_ablation.txt")
open(os.path.join(P, "r01_ablation.txt"), "w").write("\n".join(ab) + "\n")

tr = ["# tools/fused_trace.py: library rebuilt with -DFU_TRACE, one workgroup per bin (SC_FUSED_SPLIT=1), shader-clock cycles per chunk of",
      "# 32 observation rows summed over the 219 chunks of workgroup 0.  The timers themselves cost ~450 cycles per tick (4 ticks per chunk),",
      "# so the absolute numbers are inflated by ~15 %; what matters is the balance between the three groups of waves.",
      "# CSM waves 0-3: staging of quads 0-3 after their products; abs waves 4-7 (block set 0): staging of quads 4-7 before theirs; waves 8-11: set 1."]
tr += lines("fused_trace.txt")
tr += ["", "# tools/mtfft_trace.py: library rebuilt with -DMT_TRACE, wave 0 of one workgroup of mtfft16_kernel<8> (16 waves per CU share the SIMDs,",
       "# so a phase's elapsed cycles include the time its instructions wait behind the other workgroups')."]
tr += lines("mtfft_trace.txt")
open(os.path.join(P, "r01_fused_trace.txt"), "w").write("\n".join(tr) + "\n")

hw = ["# tools/hbm_write_bench.cpp on MI355X: pure HBM streams (6 GiB) and the store pattern of mtfft16_kernel with the arithmetic removed",
      "# (workgroup (c-tile, trial, window) walks 7 tapers; per taper 129 frequency rows x 256 B / 512 B / 1 KB).  The 'f-major walk' line touches a",
      "# small region only (cache resident) and is not an HBM number.  Conclusion: writes sustain 5.2-5.9 TB/s, reads 6.3 TB/s; the kernel's",
      "# store pattern alone takes 1.18 ms for the 6.47 GB of cfg3 spectra."]
hw += lines("hbm_write.txt")
open(os.path.join(P, "r01_hbm_write.txt"), "w").write("\n".join(hw) + "\n")

al = ["# tools/abs_loop_bench.cpp on MI355X: the inner loop of the |Im| role in isolation -- per (block,row) step one v_mfma_f32_32x32x16_bf16",
      "# (C = 0) and 16 x v_add_f32 acc, acc, |d| -- for 1..3 such waves per SIMD, alone and next to a wave issuing back-to-back",
      "# v_mfma_f32_16x16x32_bf16 (5.4 per step).  ns per step per SIMD at the clock the part sustains under the load.",
      "# ('16 adds only' replaces the MFMA by 16 v_mov of zeros, hence slower than the real loop; 'mfma only' keeps d alive with an empty asm.)"]
al += lines("abs_loop.txt")
open(os.path.join(P, "r01_abs_loop.txt"), "w").write("\n".join(al) + "\n")

tv = {}
for l in lines("hbm_traffic.txt"):
    parts = l.split()
    if len(parts) >= 4:
        tv[(parts[0][:24], parts[1])] = float(parts[-1].split("=")[1])
print("HBM KB per dispatch:", {k: v for k, v in tv.items() if "fused" in k[0] or "mtfft16" in k[0]})
d = json.loads(line)
print("step %.3f ms, value %.4g, stages %s" % (d["ms_per_step"], d["value"], d["roofline"]["stage_ms"]))

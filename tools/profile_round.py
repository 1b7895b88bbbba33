"""Turn the raw outputs of tools/profile_round.sh (gpurun_out/$ROUND/) into the committed profiles/$ROUND_* files, including
profiles/$ROUND_hbm_traffic.json -- the PMC-derived HBM bytes per launch that bench.py reports as roofline.traffic."""
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")
F = os.path.join(ROOT, "gpurun_out", ROUND)
P = os.path.join(ROOT, "profiles")


def lines(name):
    path = os.path.join(F, name)
    return open(path).read().splitlines() if os.path.exists(path) else []


def kernel_source_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "spectral_connectivity_amd", "csrc")
    for name in ("sc_fused.hip", "sc_fused2.hip", "sc_fused_common.h", "sc_mtfft.hip", "sc_mtfft_long.hip", "sc_mtfft_bfly.h", "sc_measure.hip", "sc_stage.h", "sc_common.h",
                 "sc_wilson_pair.hip", "sc_wilson_fft.h"):   # = bench.py
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc(name, counter):
    """{kernel name prefix: average counter value per dispatch} from a rocpd_summary PMC listing."""
    out = {}
    for l in lines(name):
        m = re.match(r"\s+(\S+)\s+(\S+)\s+n=\s*(\d+)\s+avg=(\S+)", l)
        if m and m.group(2) == counter:
            out[m.group(1)] = float(m.group(4))
    return out


def find(d, prefix):
    for k, v in d.items():
        if k.startswith(prefix) or prefix in k:
            return v
    return None


bench = [l for l in lines("bench_line.json") if l.startswith("{")]
final = os.path.join(ROOT, "gpurun_out", "final_bench_line.json")      # a bench run AFTER the traffic JSON below was written
if os.path.exists(final) and os.path.getmtime(final) > os.path.getmtime(os.path.join(F, "bench_line.json")):
    bench = [l for l in open(final).read().split("\n") if l.startswith("{")] or bench
if bench:
    open(os.path.join(P, ROUND + "_bench_line.json"), "w").write(bench[-1] + "\n")
under = [l for l in lines("bench_under_rocprof.json") if l.startswith("{")]
hdr = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 2 --timed-only   (cfg3, 1x MI355X, round %s;" % ROUND.lstrip("r0") + " tools/profile_round.sh;",
       "# --timed-only: nothing but the warm-up and the timed steps, so every average below is over the 42 launches of the step loop)"]
if under:
    st = json.loads(under[-1])
    sm = st["roofline"]["stage_ms"]
    hdr.append("# stage times of the same run from the library's own hipEvent timers (sc_last_timing): " +
               ", ".join(f"{k} {v:.3f} ms" for k, v in sm.items()) + f"; step {st['ms_per_step']:.2f} ms")
    hdr.append("# (fused_stage_b = fused2_kernel; its split-bin partial records are summed by measure_tile_multi_kernel; averages below include the 2 warm-up launches, which run before the clock has settled)")
open(os.path.join(P, ROUND + "_bench_kernel_stats.txt"), "w").write("\n".join(hdr + lines("kt.txt")[:12]) + "\n")

fetch, write = pmc("fetch.txt", "FETCH_SIZE"), pmc("write.txt", "WRITE_SIZE")
rows = [("fused2_kernel", "_Z13fused2_kernel", True), ("fused_csm_absim_kernel", "_Z22fused_csm_absim", True),
        ("fused_combine_kernel", "_Z20fused_combine", True), ("planes_absmax_kernel", "_Z20planes_absmax", True),
        ("mtfft_long_kernel", "_Z17mtfft_long", False), ("mtfft16_kernel", "_Z14mtfft16", False), ("measure_tile_multi_kernel", "measure_tile_multi", False)]
txt = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md), python bench.py --steps 2",
       "# --warmup 1 --timed-only (cfg3, 1x MI355X), round " + ROUND[1:].lstrip("0") + " (tools/profile_round.sh).  Counter values are KB per dispatch.  gfx950 correction: FETCH_SIZE",
       "# reports half of a wide (16 B / lane) coalesced read stream -> doubled for the kernels whose reads are such streams (marked x2);",
       "# WRITE_SIZE as is.",
       f"{'kernel':28s} {'FETCH_SIZE[KB]':>15s} {'WRITE_SIZE[KB]':>15s} {'HBM bytes (corrected)':>24s}"]
traffic = {}
for name, prefix, wide in rows:
    f_kb, w_kb = find(fetch, prefix), find(write, prefix)
    if f_kb is None or w_kb is None:
        continue
    total = (2 if wide else 1) * f_kb * 1024 + w_kb * 1024
    traffic[name] = total
    txt.append(f"{name:28s} {f_kb:15.4g} {w_kb:15.4g} {total / 1e9:20.3f} GB{'  (x2)' if wide else ''}")
open(os.path.join(P, ROUND + "_hbm_traffic.txt"), "w").write("\n".join(txt) + "\n")
stage_b = "fused2_kernel" if "fused2_kernel" in traffic else "fused_csm_absim_kernel"
if stage_b in traffic:
    rec = {"kernel_source_hash": kernel_source_hash(),
           "source": "profiles/" + ROUND + "_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 on gfx950)",
           "cfg3": {"fused_stage_b": traffic[stage_b] + traffic.get("fused_combine_kernel", 0.0),
                    "planes_scales": traffic.get("planes_absmax_kernel"),
                    "mtfft_fused": traffic.get("mtfft_long_kernel", traffic.get("mtfft16_kernel")),
                    "measure_epilogue": traffic.get("measure_tile_multi_kernel")}}
    # BASELINE configs[3]: the kernels of the resident pairwise Granger (sc_wilson_pair.hip), summed over the entry point
    f4, w4 = pmc("fetch4.txt", "FETCH_SIZE"), pmc("write4.txt", "WRITE_SIZE")
    g_rows = [("wilson_pair_kernel", "wilson_pair_kernel"), ("pair_lag0_kernel", "pair_lag0"), ("pair_granger_kernel", "pair_granger"),
              ("pair_fill_nan_kernel", "pair_fill_nan"), ("pair_consts_kernel", "pair_consts")]
    g_txt, g_total = [], 0.0
    for name, key in g_rows:
        fk, wk = find(f4, key), find(w4, key)
        if fk is None or wk is None:
            continue
        tot = fk * 1024 + wk * 1024                  # (scattered 4- / 8-byte reads of the records: no wide-stream correction)
        g_total += tot
        g_txt.append(f"{name:28s} {fk:15.4g} {wk:15.4g} {tot / 1e9:20.3f} GB")
    if g_txt:
        open(os.path.join(P, ROUND + "_hbm_traffic.txt"), "a").write(
            "# BASELINE configs[3] (python bench.py --config cfg4 --steps 2 --warmup 1): pairwise Granger, resident 2 x 2 Wilson kernel\n"
            + "\n".join(g_txt) + f"\n{'entry point total':28s} {'':15s} {'':15s} {g_total / 1e9:20.3f} GB\n")
        rec["cfg4"] = {"granger_pairwise": g_total}
    json.dump(rec, open(os.path.join(P, ROUND + "_hbm_traffic.json"), "w"), indent=1)

sq = lines("sq.txt")
if sq:
    vals = {}
    for l in sq:
        m = re.match(r"\s+(\S+)\s+(\S+)\s+n=\s*(\d+)\s+avg=(\S+)", l)
        if m and ("fused2_kernel" in m.group(1) or "fused_csm_absim" in m.group(1)):
            vals[m.group(2)] = float(m.group(4))
    busy, valu, mf = vals.get("SQ_BUSY_CYCLES", 1.0), vals.get("SQ_INSTS_VALU", 0.0), vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    h2 = ["# rocprofv3 --kernel-trace --pmc (one pass of 8 SQ counters), cfg3, 1x MI355X, round " + ROUND[1:].lstrip("0") + " (tools/profile_round.sh).",
          "# Averages per dispatch PER SHADER ENGINE (32 SEs x 8 CUs = 32 SIMDs each): SQ_INSTS_* are wave instructions, SQ_BUSY_CYCLES and",
          "# SQ_VALU_MFMA_BUSY_CYCLES cycles, SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles.",
          "# fused kernel: VALU issue = %.3g instr x 4 cycles / 32 SIMDs = %.3g of %.3g busy cycles = %.0f %%; matrix pipe = %.4g / 32 SIMDs = %.3g cycles = %.0f %%"
          % (valu, valu * 4 / 32, busy, 100 * valu * 4 / 32 / busy, mf, mf / 32, 100 * mf / 32 / busy)]
    open(os.path.join(P, ROUND + "_pmc_fused.txt"), "w").write("\n".join(h2 + sq) + "\n")

for src, dst, head in (("mvar_64ch.txt", ROUND + "_mvar_64ch.txt", "# tools/mvar_time.py 64 1792 256: full 64 x 64 Wilson factorisation + DTF, 7 windows x 256 bins (round " + ROUND[1:].lstrip("0") + ")"),
                       ("mvar_128ch.txt", ROUND + "_mvar_128ch.txt", "# tools/mvar_time.py 128 1792 256: full 128 x 128 Wilson factorisation + DTF, 7 windows x 256 bins (round " + ROUND[1:].lstrip("0") + ")"),
                       ("engine_time.txt", ROUND + "_engine_time.txt", "# float32 and float64 engine on the BASELINE configurations (round " + ROUND[1:].lstrip("0") + ")"),
                       ("stage_a.txt", ROUND + "_stage_a.txt", "# tools/stage_a_breakdown.py: stage A per window length, cfg3 data volume (round " + ROUND[1:].lstrip("0") + ")"),
                       ("plane_pass.txt", ROUND + "_plane_pass.txt", "# tools/plane_pass_time.py: stage B per plane family, cfg3 data volume, round " + ROUND[1:].lstrip("0")),
                       ("shape_sweep.txt", ROUND + "_shape_sweep.txt", "# tools/shape_sweep.py, round " + ROUND[1:].lstrip("0")),
                       ("fused_ablation.txt", ROUND + "_fused_ablation.txt", "# tools/fused_ablation.py (the complex64 kernel of sc_fused.hip; SC_FUSED_DEBUG; results WRONG when set)"),
                       ("fused2_ablation.txt", ROUND + "_fused2_ablation.txt", "# tools/fused2_time.py 0 1 2 3 8 9 10 11 64: stage B on the planes format (sc_fused2.hip) under SC_FUSED_DEBUG (1 = CSM waves skip their MFMAs, 2 = |Im s| waves skip theirs, 8 = no HBM loads after the first chunk, 64 = no intermediate folds), next to the complex64 kernel"),
                       ("stage_a_planes_ab.txt", ROUND + "_stage_a_planes_ab.txt", "# tools/stage_a_planes_ab.py: stage A into the planes format against the complex64 output, SC_MTFFT_DEBUG switches, scale pre-pass alone"),
                       ("stage_a_planes_check.txt", ROUND + "_stage_a_planes_check.txt", "# tools/stage_a_planes_check.py: stage A into the planes format and into complex64, both against the float64 transform of the same samples (loud channel x300 next to channel 0, quiet channel x1e-3 at C/2)"),
                       ("sq2.txt", ROUND + "_pmc_fused_waits.txt", "# second SQ counter pass of the bench command (wait / LDS counters; rocprofv3 --kernel-trace --pmc)")):
    body = [l for l in lines(src) if "amdgpu.ids" not in l]
    if body:
        open(os.path.join(P, dst), "w").write("\n".join([head] + body) + "\n")
for src, dst, head in (("api_wall.txt", ROUND + "_api_wall.txt", "# tools/api_wall.py: NumPy time series -> NumPy results through the public API at the cfg3 shape (third call), and the torch-free NumPy host"),
                       ("numpy_host.txt", ROUND + "_numpy_host.txt", "# tools/numpy_host_time.py: the torch-free host (ctypes + NumPy over sc_device_alloc / sc_memcpy_* / sc_stream_*), cfg3 shape"),
                       ("stage_a_wide.txt", ROUND + "_stage_a_wide.txt", "# tools/stage_a_long.py: stage A for long windows at the cfg3 data volume (128 channels, 7 tapers, one window per trial, linear detrend): the round-3 kernels (SC_MTFFT_LONG=0) against the anti-phase kernel of sc_mtfft_long.hip (the default)"),
                       ("stage_a_antiphase_ab.txt", ROUND + "_stage_a_antiphase_ab.txt", "# tools/stage_a_antiphase_ab.py: complex64 stage A at the cfg3 volume, round-3 kernels against the anti-phase kernel with either workgroup size"),
                       ("global_canonical.txt", ROUND + "_global_canonical.txt", "# tools/global_time.py: global coherence (1024 two-sided bins) and canonical coherence with large groups at the cfg5 shape"),
                       ("measure_table.txt", ROUND + "_measure_table.txt", "# tools/measure_table.py: every measure of the public interface at the cfg3 shape"),
                       ("fused2_fold_ab.txt", ROUND + "_fused2_fold_ab.txt", "# (profile round's box)"),
                       ("issue_rates_f64.txt", ROUND + "_issue_rates_f64.txt", "# fp64 issue rates (profile round's box)"),
                       ("sharded_one_rank.txt", ROUND + "_sharded_one_rank.txt", "# tools/sharded_one_rank.sh: the N > 1 code path of bench.py on ONE rank (nccl backend, world size 1, every collective called, nothing crosses a link) at the trial counts a rank holds at 1 / 2 / 4 / 8 GPUs, for 4 / 2 / 1 pipelined frequency groups; 'plain' = the N = 1 path"),
                       ("kt4.txt", ROUND + "_bench_kernel_stats_cfg4.txt", "# rocprofv3 --kernel-trace --stats -- python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline (pairwise Granger, 2016 pairs x 4096 bins)"),
                       ("mvar_size_time.txt", ROUND + "_mvar_size_time.txt", "# tools/mvar_size_time.py: full Wilson factorisation + DTF across system sizes, one window x 256 bins, float64 records"),
                       ("stage_a_ab.txt", ROUND + "_stage_a_ab_final.txt", "# tools/stage_a_ab.py 0 16 2 4 6 (and again with SC_AB_LONG=1): the anti-phase stage A under SC_MTFFT_DEBUG inside one process, wall time per call incl. launch (16 = non-temporal stores, 2 = no passes, 4 = no split / store loop, 6 = neither: tile load + detrend + the empty slots' barriers)")):
    body = [l for l in lines(src) if "amdgpu.ids" not in l]
    if body:
        open(os.path.join(P, dst), "w").write("\n".join([head] + body) + "\n")
for src, dst, head in (("stage_a_mixed.txt", ROUND + "_stage_a_mixed.txt", "# tools/stage_a_mixed.py time: stage A for the window lengths that are not powers of two, the round-2 kernels (SC_MTFFT_MIXED=0) against sc_mtfft_mixed.hip (geometry 0 / 1 of every length), both outputs (profile round's box)"),
                       ("e2e_lengths.txt", ROUND + "_e2e_lengths.txt", "# tools/e2e_lengths.py: stage A -> stage B -> epilogue (coherence + wPLI) at the cfg3 data volume per window length, the engine's own choice of kernels and device format"),
                       ("api_timeline.txt", ROUND + "_api_timeline.txt", "# tools/api_timeline.py: the public-API pass (series in HBM, cfg3) phase by phase, and the engine chain after n ms of idle GPU (profile round's box)")):
    body = [l for l in lines(src) if "amdgpu.ids" not in l]
    if body:
        open(os.path.join(P, dst), "w").write("\n".join([head] + body) + "\n")
for cfg in ("cfg2", "cfg4", "cfg5"):
    body = [l for l in lines(f"bench_{cfg}.json") if l.startswith("{")]
    if body:
        open(os.path.join(P, f"{ROUND}_bench_line_{cfg}.json"), "w").write(body[-1] + "\n")
print("profiles written:", sorted(f for f in os.listdir(P) if f.startswith(ROUND + "_")))

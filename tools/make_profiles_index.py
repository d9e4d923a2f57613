#!/usr/bin/env python3
"""Writes profiles/INDEX.md: the evidence files the CURRENT documents (README.md) cite, one line each,
grouped by what they are evidence of.  Everything else under profiles/ belongs to HISTORY.md (rounds 1-3)."""
import glob, itertools, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def expand(pattern):
    m = re.search(r"\{([^{}]*)\}", pattern)
    if not m:
        return [pattern]
    out = []
    for alt in m.group(1).split(","):
        out += expand(pattern[:m.start()] + alt + pattern[m.end():])
    return out


cited = {}
for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
    text = open(os.path.join(ROOT, doc)).read()
    for ref in re.findall(r"profiles/([A-Za-z0-9_{},.*-]+)", text):
        ref = ref.rstrip(".,")
        for name in expand(ref):
            for path in (glob.glob(os.path.join(P, name)) or [os.path.join(P, name)]):
                cited.setdefault(os.path.basename(path), set()).add(doc)
cited.pop("INDEX.md", None)   # (this file)
missing = [n for n in cited if not os.path.exists(os.path.join(P, n))]
WHAT = [
    (r"r06_bench_cfg2\.json", "the headline `bench.py` line (BASELINE configs[1], N = 1) of round 6's final build, with `roofline`, `cpu_baseline`, `parity_full_stream`, `strong_curve_n1`, `end_to_end`"),
    (r"r06_bench_cfg2_steps20\.json", "the same command as the driver runs it (`--steps 20 --warmup 3`)"),
    (r"r06_bench_(cfg3|cfg4|cfg5|hq48|hq44|dn8|cfg2_s16)\.json", "`bench.py` lines of the other workloads, same box and build"),
    (r"r06_bench_n2_sharedgpu_gloo\.json", "`bench.py --gpus 2` with the ranks SHARING one GPU over gloo (validation of the N > 1 path; not a scaling measurement)"),
    (r"r06_trace_timed_means\.log", "rocprofv3 `--kernel-trace` of the bench command: mean over ALL dispatches and over the TIMED ones bench.py names (`tools/trace_timed_mean.py`)"),
    (r"r06_kernel_stats_.*\.csv", "`rocprofv3 --kernel-trace --stats` summary of the same `bench.py` command (average over every dispatch, clock ramp included)"),
    (r"r06_kernel_trace_head_.*\.csv", "first dispatches of that trace: grid, LDS, register counts"),
    (r"r06_(cfg2|cfg3|cfg4)_pmc_summary\.txt", "per-dispatch PMC means (separate `--pmc` passes), stamped with the library's source id; `bench.py` quotes `traffic` / `roofline_valu` from them"),
    (r"r06_lines\.log", "one line per workload of the final lease: kernel, µs, fraction, parity, traffic"),
    (r"r06_kseg_xcd_ab\.log", "`k_seg`: tiles dealt one by one against runs of 4 ... 128 consecutive tiles per XCD, same box, three rounds"),
    (r"r06_kseg_xcd_pmc\.log", "... and the counters of the two legs: `FETCH_SIZE` 87,343 -> 12,557 KiB per launch"),
    (r"r06_kseg_ab\.log", "`k_seg` on the final build: default against `xcd_run` 0 and against `k_up2`"),
    (r"r06_host_path_ab\.log", "the host-pointer entry points from pageable memory: copies in 1 MiB pieces against whole copies (the runtime's pinned-in-place path)"),
    (r"r06_host_paths\.log", "the host-pointer entry points on the final build"),
    (r"r06_loop_a_summary\.log", "the GPU suite ten times in fresh processes BEFORE the containment: one death"),
    (r"r06_loop_[bcd]_summary\.log", "the GPU suite in loops of fresh processes AFTER it, three leases: 10 + 10 + 12 green"),
    (r"r06_death_loop_a\.txt", "the death of lease a: the runtime's fault line (a heap address), the flight recorder, the Python stack (`tensor.cpu()`)"),
    (r"r06_repro3_death_flight_recorder\.log", "the death under guarded library allocations: same test, same copy, a heap address"),
    (r"r06_repro3_summary\.log", "the guarded-buffer tests and the suite under `CLOWNRESAMPLER_AMD_GUARD_MALLOC=1 / 2`"),
    (r"r06_copy_path\.txt", "the runtime's own log of a 0.9 MB, a 1.2 MB and a 5 MB copy into pageable memory: staging up to 1 MiB, page-locked in place above"),
    (r"r06_minxfer_probe\.txt", "`GPU_PINNED_MIN_XFER_SIZE` moves that threshold (what the test and bench processes set)"),
    (r"r06_hipfree_probe\.log", "`hipFree` / `hipHostFree` wait for work in flight"),
    (r"r06_va_reuse_probe\.log", "device addresses handed out again by `hipMalloc`: kernels and copy engines agree on their contents (82,393 rounds)"),
    (r"r06_vmm_probe.*\.log", "address ranges of the virtual-memory API handed out again: they do NOT (why the guarded allocators keep their ranges)"),
    (r"r06_asan_gpu_suite\.log", "the GPU suite with the library's host code under AddressSanitizer"),
    (r"r06_arming_probe\.log", "a pinned-path copy after every GPU test: three full runs survive"),
    (r"r06_(pin_cache_probe_heap|pin_cache_probe_mmap|pin_overlap_probe|register_reuse_probe|window_heap_trim)\.log", "the stories about the runtime's pinned copies that were tried in isolation and survive"),
    (r"r06_repro1_summary\.log", "the first loop of the round: 25 windows + 6 full suites, green"),
    (r"r05_bench_cfg2\.json", "the headline `bench.py` line (BASELINE configs[1], N = 1) of round 5's final build, with `roofline`, `cpu_baseline`, `parity_full_stream`, `strong_curve_n1`, `end_to_end`, `timed_dispatches`"),
    (r"r05_bench_cfg2_steps20\.json", "the same command as the driver runs it (`--steps 20 --warmup 3`)"),
    (r"r05_bench_(cfg3|cfg4|cfg5|hq48|hq44|dn8|cfg2_s16)\.json", "`bench.py` lines of the other workloads, same box and build"),
    (r"r05_bench_n\d_sharedgpu_gloo\.json", "`bench.py --gpus N` with the ranks SHARING one GPU over gloo (validation of the N > 1 path; not a scaling measurement)"),
    (r"r05_trace_timed_means\.log", "rocprofv3 `--kernel-trace` of the bench command: mean over ALL dispatches and over the TIMED ones bench.py names (`tools/trace_timed_mean.py`)"),
    (r"r05_kernel_stats_.*\.csv", "`rocprofv3 --kernel-trace --stats` summary of the same `bench.py` command (average over every dispatch, clock ramp included)"),
    (r"r05_kernel_trace_head_.*\.csv", "first dispatches of that trace: grid, LDS, register counts"),
    (r"r05_(cfg2|cfg3|cfg4)_pmc_summary\.txt", "per-dispatch PMC means (separate `--pmc` passes), stamped with the library's source id; `bench.py` quotes `traffic` / `roofline_valu` from them"),
    (r"r05_(cfg2_s16|dn1|mono|dn2)_pmc_summary\.txt", "the same for the int16 output form and the mono / mild-downsampling shapes (taken after the evidence run, same library build)"),
    (r"r05_(hq48|hq44|dn8)_pmc_summary\.txt", "the same for the LDS-bound long-window shapes (VERDICT r4 item 3 asked for stamped summaries of the final build)"),
    (r"r05_(cfg3|hq44|dn8)_lds_counters_before\.txt", "LDS-side counters (`SQ_LDS_*`, waits) of round 4's kernels, taken before round 5 touched anything"),
    (r"r05_kup2_lds_ablations\.log", "timing-only builds of `k_up2`: no row reads per frame / conflict-free staging writes (what each costs)"),
    (r"r05_kseg_.*\.log", "`k_seg`: same-box A/B against `k_up2`, tile sizes, timing-only ablations, where a wave's cycles go, forms tried and not kept"),
    (r"r05_seg_ratio_sweep\.log", "`k_seg` against `k_up2` / `k_wave2` by upsampling ratio, 40 M output frames (`tools/seg_ratio_sweep.py`)"),
    (r"r05_valurate\.log", "ns per wave-instruction of the tap's building blocks, incl. `v_pk_fma_f32` with a scalar weight pair"),
    (r"r05_scatterwrite\.log", "what `k_seg`'s store pattern (whole lines, 64 segments 512 KB apart) costs against a contiguous one"),
    (r"r05_pkfma_sgpr\.log", "`v_pk_fma_f32` with an SGPR-pair operand: the four `op_sel` forms give the expected lanes"),
    (r"r05_parity_holes_gpu_tests\.log", "the C99-library and `HighLevel_Adjust` mid-stream tests, first green run"),
    (r"r05_all_workloads\.log", "one line per workload of `bench.py`'s table: kernel, µs, Msamples/s, fraction of the roofline, full-stream parity"),
    (r"r05_channel_table\.log", "every channel count 1-16 × ratio, 3 and 8 lobes (`tools/channel_table.py`)"),
    (r"r05_gpu_tests.*\.log", "`pytest -m gpu` on the evidence box"),
    (r"r05_host_paths_pinned\.log", "host-pointer entry point from pageable / page-locked memory"),
    (r"r05_preflight_dry.*", "shared-GPU dry run of `tools/multi_gpu_preflight.sh` (VALIDATION ONLY)"),
    (r"r05_soak\.log", "`tests/soak_gpu.py`: the two runs that died before the fixes, then every run after them (GPU boxes, both ABIs; the CPU runs against the fake seam under AddressSanitizer)"),
    (r"r05_kwave2_lds_forms\.log", "timing-only forms of `k_wave2` (`hq44`, `dn8`, `hq48`): window / row reads without bank conflicts - the ceiling of the long-window shapes"),
    (r"r05_kwave2_frame_pipeline_ab\.log", "`k_wave2` with the frames of a wave-tile as one pipeline of LDS reads against frame by frame (bit-exact, ±1.5 %: not kept)"),
    (r"r05_kpoly_forms\.log", "timing-only forms of the headline `k_poly<2,5>`: compute floor, floor without bank conflicts, without stores; the int16 output by variant and tile size"),
    (r"r05_ldswin\.log", "`tools/microbench/ldswin.hip`: LDS cycles per window slot by lane spacing for `ds_read_b64`, 8-byte-aligned and 16-byte-aligned `ds_read_b128`"),
    (r"r05_mono_env_sweep\.log", "the host's rules against their environment overrides on the mono and mild-downsampling shapes (rotation, tile size, tickets, dual mono, lane order): no override beats the rule"),
    (r"r05_gpu_tests_last\.log", "`pytest -m gpu` with the long-stream tests added after the evidence run (same library build)"),
    (r"r05_ldsvalu\.log", "`tools/microbench/ldsvalu.hip`: LDS reads and VALU work of a wave add up rather than overlap"),
    (r"r04_bench_cfg2\.json", "the headline `bench.py` line (BASELINE configs[1], N = 1) of the final build, with `roofline`, `cpu_baseline`, `parity_full_stream`, `strong_curve_n1`, `end_to_end`"),
    (r"r04_bench_(cfg3|cfg4|cfg5|hq48|cfg2_s16)\.json", "`bench.py` lines of the other workloads, same box and build"),
    (r"r04_bench_n\d_sharedgpu_gloo\.json", "`bench.py --gpus N` with the ranks SHARING one GPU over gloo (validation of the N > 1 path; not a scaling measurement)"),
    (r"r04_kernel_stats_.*\.csv", "`rocprofv3 --kernel-trace --stats` summary of the same `bench.py` command (average launch duration per kernel)"),
    (r"r04_kernel_trace_head_.*\.csv", "first dispatches of that trace: grid, LDS, register counts"),
    (r"r04_(cfg2|cfg3|cfg4)_pmc_summary\.txt", "per-dispatch PMC means (separate `--pmc` passes), stamped with the library's source id; `bench.py` quotes `traffic` / `roofline_valu` from them"),
    (r"r04_(mono|dn1|hq48m|hq44m|dn8m)_pmc_summary\.txt", "the same for the mono workloads, final build (dual mono)"),
    (r"r04_.*_before_pmc_summary\.txt", "PMC means of the mono workloads BEFORE round 4 touched them (VERDICT r3 item 3 asked for these first)"),
    (r"r04_all_workloads\.log", "one line per workload of `bench.py`'s table: kernel, µs, Msamples/s, fraction of the roofline, full-stream parity"),
    (r"r04_channel_table\.log", "every channel count 1-16 × ratio, 3 and 8 lobes (`tools/channel_table.py`)"),
    (r"r04_gpu_tests\.log", "`pytest -m gpu` of the final build on the evidence box"),
    (r"r04_host_paths_pinned\.log", "host-pointer calls from pageable against page-locked buffers, by `CLOWNRESAMPLER_AMD_HOST_DIRECT` mode"),
    (r"r04_kup2_ab\.log", "cfg 3, same box: round 3's `k_up2`, round 4's with the integer chain, with the FP32 chain, with the shared window conversion"),
    (r"r04_kup2_frame\.s", "annotated disassembly of one `k_up2` frame"),
    (r"r04_kup2_ablations\.log", "`k_up2` timing-only ablations and in-kernel phase stamps"),
    (r"r04_rtzchain\.log", "the FP32 round-toward-zero chain against the integer definition (2.7e8 frames), `tools/microbench/rtzchain.hip`"),
    (r"r04_valurate\.log", "issue cost of the VALU instructions the taps are built from, `tools/microbench/valurate.hip`"),
    (r"r04_dual_mono_ab\.log", "dual mono against the mono kernels, same box"),
    (r"r04_three_group_tiles_ab\.log", "3,072-frame tiles against 2,048 (48 → 44.1 kHz shapes)"),
    (r"r04_mono_rotated_rows_ab\.log", "rotated rows for mono"),
    (r"r04_dn8m_kwave2_ab\.log", "mono 44.1 → 8 kHz: `k_wave2<1,33>` against `k_poly<1,33>`"),
    (r"r04_dmarepack\.log", "LDS-DMA repacking odd frames while copying (groundwork for odd wide frames)"),
    (r"r02_streambench\.log", "what HBM delivers for the path's read:write mix with no arithmetic (the flat-copy ceiling)"),
    (r"r03_channel_table\.log", "round 3's channel table (cells whose code did not change)"),
]
lines = ["# profiles/INDEX.md - the evidence the CURRENT documents cite", "",
         "Everything under `profiles/` that `DESIGN.md`, `README.md` or `INTEGRATION.md` refers to (generated by `tools/make_profiles_index.py`).",
         "The other files in this directory are the evidence of rounds 1-3, cited by `HISTORY.md`; `profiles/README.md` describes round 1's.", "",
         "| file | what it is | cited by |", "|---|---|---|"]
for name in sorted(cited):
    what = next((w for pat, w in WHAT if re.fullmatch(pat, name)), "")
    lines.append("| `%s`%s | %s | %s |" % (name, " (MISSING)" if name in missing else "", what, ", ".join(sorted(cited[name]))))
open(os.path.join(P, "INDEX.md"), "w").write("\n".join(lines) + "\n")
print(len(cited), "files cited;", "missing:", missing)

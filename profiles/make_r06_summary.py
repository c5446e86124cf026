#!/usr/bin/env python3
"""Assembles profiles/r06_round_end.md from the outputs of `bash profiles/collect_r06_final.sh <tag>` (gpurun_out/<tag>) and the counter
summary profiles/r06_pmc.json (profiles/collect_r06_pmc.sh + pmc_summary_r06.py).
usage: make_r06_summary.py [tag] > profiles/r06_round_end.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06fin"
A = os.path.join(ROOT, "gpurun_out", TAG)


def lines(f):
    p = os.path.join(A, f)
    return [l for l in open(p).read().strip().splitlines() if l.startswith("{")] if os.path.exists(p) else []


def last(f):
    ls = lines(f)
    return ls[-1] if ls else None


def block(f):
    return "```json\n" + (last(f) or "(not collected)") + "\n```\n"


def text(f):
    p = os.path.join(A, f)
    return open(p).read() if os.path.exists(p) else "(not collected)"


o = []
o.append("# Round 6 -- end-of-round measurement set (MI355X, 1 GPU)\n")
o.append("Collected through gpurun with `bash profiles/collect_r06_final.sh %s`; counters from `bash profiles/collect_r06_pmc.sh` (`r06_pmc.json`, "
         "`pmc_summary_r06.py`).  Raw outputs live under `gpurun_out/` (scratch); this file is `profiles/make_r06_summary.py %s`.  Beside it: "
         "`r06_kernel_a.md` (eight experiments on the band step), `r06_genome_reduction.md` (the N > 1 line as a genome-level reduction), `r06_batch_binding.md` (batches in pieces, "
         "the stretch cache), `r06_detector.md`, `r06_kernel_b_lds.md`, `r06_soak.md`, `r06_boxes.md` (the same commands on several boxes); issue-cycle calibration: "
         "`r04_valu_calibration.json`.\n" % (TAG, TAG))
ps = [l for l in text("pytest.log").splitlines() if " passed" in l or " failed" in l]
o.append("GPU tests on the same box (`python -m pytest tests -m gpu -q`, the sanitizer builds of the shims included): `%s`\n" % (ps[-1].strip() if ps else "?"))
pr = [l for l in text("probe.log").splitlines() if l.startswith("probe:")]
if pr:
    o.append("Hardware probe at `np_create` (`np_ctx_info`): `%s`\n" % pr[-1])
o.append("## Bench lines\n")
o.append("The driver's command (`python bench.py --gpus 1 --steps 20 --warmup 5`): BASELINE.json configs[1] -- 100 000 reads per step (20 000 distinct x 5), ~8k events each; "
         "`value` resident, `value_streamed` host-fed, `value_ragged` log-normal lengths with its parity check; folded in, each with its parity fields: "
         "`value_from_raw` (the step from int16 raw signal), `value_eventalign` (configs[2]), `value_variants` (configs[3]), `value_binding_512` / "
         "`value_binding_8192` (reads/s through the reference-side batched binding, host memory to ScoredSite maps); `roofline` (HBM, kernel A) with "
         "`roofline_issue` (vector issue) beside it, `roofline_hmm_forward` for kernel B:\n")
o.append(block("bench_default.json"))
d = last("bench_default.json")
if d:
    d = json.loads(d)
    r = d["roofline"]
    hb_ = r.get("hbm") or r                      # round 6: the line's roofline is the vector-issue one, the HBM figures sit in roofline.hbm
    iss = r.get("issue") or {}
    o.append("| | |\n|---|---|")
    o.append("| value | %.0f reads/s, %.1f ms per step (kernel A %.1f, kernel B %.1f, glue + work items %.1f) |" % (
        d["value"], d["ms_per_step"], r["kernel_ms_per_step"]["event_align"], r["kernel_ms_per_step"]["hmm_score"], r["kernel_ms_per_step"]["glue_and_work_items"]))
    o.append("| roofline (kernel A) | %.1f GB algorithmic / %.2f ms = %.0f GB/s = **%.4f** of 8 TB/s; traffic %.1f GB per launch (counter bytes per band x this run's bands); "
             "vector issue: floor %s, priced %s of the SIMDs' issue time (%s instructions per band over %s SIMD-cycles) |" % (
                 r["algo_bytes_per_launch"] / 1e9, r["avg_launch_ms"], hb_["achieved"], hb_["frac"], (hb_["traffic"] or 0) / 1e9,
                 iss.get("valu_issue_floor"), iss.get("valu_issue_priced"), iss.get("valu_per_unit"), iss.get("simd_cycles_per_unit")))
    o.append("| streamed / ragged | %.0f / %.0f reads/s; ragged parity: %s |" % (d.get("value_streamed") or 0, d.get("value_ragged") or 0, json.dumps((d.get("ragged") or {}).get("check"))))
    ea, va = d.get("eventalign") or {}, d.get("variants") or {}
    if "value" in ea:
        er = ea["roofline"]
        o.append("| eventalign leg | %.0f reads/s, %.1f ms per 50 000 reads, chain kernel %.1f ms, roofline frac %.4f (traffic %.1f GB), issue %s / %s, cpu %s |" % (
            ea["value"], ea["ms_per_step"], ea["kernel_ms_per_step"]["eventalign_chain"], er["frac"], (er.get("traffic") or 0) / 1e9,
            (er.get("issue") or {}).get("valu_issue_floor"), (er.get("issue") or {}).get("valu_issue_priced"), json.dumps(ea.get("cpu_baseline"))))
        # the line's `traffic` / `issue` are look-ups into the counter summary committed WHEN THE LINE WAS PRINTED; recomputed here from the committed one
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import pmc_lookup
            segs = ea.get("hmm_align_calls_per_step")
            cyc = ea["kernel_ms_per_step"]["eventalign_chain"] * 1e-3 * 2.4e9 * 1024 / segs
            iss = pmc_lookup.issue("chain", "segment", cyc)
            if iss and abs(iss["valu_per_unit"] - ((er.get("issue") or {}).get("valu_per_unit") or 0)) > 1:
                o.append("| (eventalign leg, counters as committed) | the line above was printed with the counter summary collected BEFORE this round's chain-kernel "
                         "work (%.0f vector instructions per segment); with `r06_pmc.json` as committed (%.0f per segment, counted on the final code): traffic %.1f GB, "
                         "issue %s / %s, roofline_issue %.3f |" % ((er.get("issue") or {}).get("valu_per_unit") or 0, iss["valu_per_unit"],
                                                                   pmc_lookup.traffic("chain", "segment", segs) / 1e9, iss["valu_issue_floor"], iss.get("valu_issue_priced"),
                                                                   pmc_lookup.roofline_issue("chain", "segment", cyc)["frac"]))
        except Exception as e:  # noqa: BLE001
            o.append("| (eventalign leg, counters as committed) | not recomputed: %r |" % (e,))
    if "value" in va:
        vr = va["roofline"]
        o.append("| variants leg | %.0f calls/s, %.1f ms per step, roofline frac %.4f (traffic %.1f GB), issue %s / %s, cpu %s |" % (
            va["value"], va["ms_per_step"], vr["frac"], (vr.get("traffic") or 0) / 1e9, (vr.get("issue") or {}).get("valu_issue_floor"),
            (vr.get("issue") or {}).get("valu_issue_priced"), json.dumps(va.get("cpu_baseline"))))
    fr, bl_ = d.get("from_raw") or {}, d.get("binding") or {}
    if "value" in fr:
        o.append("| from-raw leg | %.0f reads/s, %.1f ms per 100 000 reads (%s), check %s |" % (fr["value"], fr["ms_per_step"], json.dumps(fr["kernel_ms_per_step"]), json.dumps(fr.get("check"))))
    for k in ("records_512", "records_8192"):
        if k in bl_:
            b = bl_[k]
            o.append("| binding leg, %d records per batch | %.0f reads/s over %d batches (%.2f ms per batch), records not ok %d, sites match the reference: %s |" % (
                b["records_per_batch"], b["value"], b["batches"], b["ms_per_batch"], b["records_not_ok"], b.get("sites_match_reference")))
    rb = d.get("roofline_hmm_forward") or {}
    if rb:
        o.append("| kernel B in the same step | HBM frac %s (traffic %.1f GB), roofline_issue frac %s |" % ((rb.get("hbm") or {}).get("frac"), ((rb.get("hbm") or {}).get("traffic") or 0) / 1e9, (rb.get("roofline_issue") or {}).get("frac")))
    o.append("| roofline (the line's: vector issue, kernel A) | %s |" % json.dumps({k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "valu_per_unit", "simd_cycles_per_unit")}))
    o.append("| cpu_baseline | %s |" % json.dumps({k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "t1_value") if k in d["cpu_baseline"]}))
    o.append("")
o.append("configs[4] on the one GPU (`--gpus 1 --genome 1 --pool 50000 --tile 5 --steps 5 --warmup 2`: 250 000 genome-placed reads per step, the per-site table keyed by "
         "genome position -- what every rank of `--gpus 8` runs by default; `profiles/r06_genome_reduction.md`):\n")
o.append(block("bench_250k.json"))
o.append("configs[0] plumbing line (`--workload cpu-t1 --cpu-sample 200`: the reference's code, one host thread, no GPU):\n")
o.append(block("bench_cpu_t1.json"))
o.append("2-rank rehearsal over gloo (`NP_BENCH_BACKEND=gloo python bench.py --gpus 2 --pool 2000 --tile 5 --steps 2`):\n")
o.append(block("bench_2rank_gloo.json"))
o.append("8-rank rehearsal of the N > 1 line on ONE MI355X (`NP_BENCH_BACKEND=gloo python bench.py --gpus 8 --pool 1000 --tile 5 --steps 2`: the eight "
         "ranks share the device and the box's 16 CPUs, the site table is all-reduced over gloo; a rehearsal of the launcher, of rank 0's `cpu_baseline` + "
         "parity sample and of every rank's own parity check (`per_rank[].parity`), NOT a scaling measurement):\n")
o.append(block("bench_8rank_gloo.json"))
o.append("## The reference-side bindings\n")
o.append("Batched (`tests/bench_batch_dropin.py`): BAM records + raw signal in host memory -> `NpBatchPipeline` (packer, device and finisher threads of its own, "
         "three batches in flight) -> the reference's `ScoredSite` maps.  `pipelined`: float samples, `pipelined_adc`: int16 samples, `2ctx`: two contexts "
         "on the one device (the multi-GPU mode of the binding), `sync`: one batch at a time.  16 CPUs (cgroup quota).\n")
bl, bn = lines("batch_dropin.json"), {json.loads(l)["batch_size"]: json.loads(l) for l in lines("batch_dropin_whole.json")}
if bl:
    o.append("| records per batch | int16 samples, batches in pieces of 512 (the default) | batches whole (`NP_BATCH_PIECE=1000000`, round 5's behaviour) | synchronous | host ms per batch: fetch + pack / maps / waiting for the device |")
    o.append("|---|---|---|---|---|")
    for l in bl:
        b = json.loads(l)
        g = lambda k: (b.get(k) or {}).get("value")
        h = (b.get("pipelined_adc") or {}).get("host_ms_per_batch") or {}
        f = lambda v: "%.0f" % v if isinstance(v, (int, float)) else "-"
        o.append("| %d | %s | %s | %s | %.1f / %.1f / %.1f |" % (b["batch_size"], f(g("pipelined_adc")), f(((bn.get(b["batch_size"]) or {}).get("pipelined_adc") or {}).get("value")), f(g("sync")),
                                                            h.get("phase1a_fetch_sizes", 0) + h.get("phase1b_pack", 0), h.get("phase3_maps", 0), h.get("finisher_wait_device", 0)))
    o.append("")
o.append("Per call (`tests/bench_percall_dropin.py`: `profile_hmm_score` through `libnp_ref_dropin.so` from inside an OpenMP loop, beside the reference's own function):\n")
o.append(block("percall.json"))
o.append("## rocprofv3 --kernel-trace --stats\n")
o.append("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0` (4 launches of the step incl. "
         "warm-up; the bench's CPU-parity sample is off, so every `np_event_align_kernel` launch is the 100 000-read step), `profiles/summarize_rocpd.py`:\n")
o.append(text("trace_default.md"))
o.append("\nThe step from raw signal (`python bench.py --from-raw 1 --pool 4000 --tile 25 --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0`):\n")
o.append(text("trace_from_raw.md"))
o.append("\nThe eventalign leg (`python bench.py --workload eventalign --steps 3 --warmup 1 --cpu-sample 0`):\n")
o.append(text("trace_eventalign.md"))
o.append("\nThe variants leg (`python bench.py --workload variants --steps 3 --warmup 1 --cpu-sample 0`):\n")
o.append(text("trace_variants.md"))
pm = os.path.join(ROOT, "profiles", "r06_pmc.json")
if os.path.exists(pm):
    P = json.load(open(pm))
    o.append("\n## Counters of the shipped kernels (`profiles/r06_pmc.json`)\n")
    o.append("`rocprofv3 --kernel-trace --pmc <set> -- python tools/pmc_workload.py ...`, one counter set per pass (`profiles/collect_r06_pmc.sh`: 8 192 reads per "
             "launch; passes over 100 000-read launches do not finish on this pool).  FETCH_SIZE x2 / WRITE_SIZE x1 as the guide's gfx950 corrections prescribe.  "
             "Issue: `floor` = vector instructions x %.2f cycles (the fastest class, `r04_valu_calibration.json`) over the UNPROFILED launch's SIMD-cycles -- a lower "
             "bound of the vector port's busy fraction; `priced` = by the class mix of the kernel's loops (an estimate; see `pmc_summary_r06.py`).\n" % P.get("fast_class_cycles_per_instruction", 0))
    o.append("| kernel | unit | vector instr / unit | scalar / unit | LDS / unit | HBM fetched + written B / unit | algorithmic B / unit | SIMD-cycles / unit | issue floor | issue priced | waiting on a counter / for an issue slot (share of wave cycles) | LDS cycles per LDS instr (conflict share) |")
    o.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, u in (("event_align", "band"), ("hmm_forward", "call"), ("hmm_forward_variants", "call"), ("chain", "segment"), ("ed_peaks", "sample"), ("recalibrate", "read"),
                   ("build_map", "read"), ("resolve", "read"), ("cm_items", "read"), ("cm_groups", "read")):
        e = P.get(key)
        if not e:
            continue
        g = lambda n: e.get("%s_per_%s" % (n, u))
        f = lambda v, fmt="%.1f": (fmt % v) if isinstance(v, (int, float)) else "-"
        o.append("| %s | %s | %s | %s | %s | %s + %s | %s | %s | %s | %s | %s / %s | %s (%s) |" % (
            key, u, f(g("valu")), f(g("salu")), f(g("lds"), "%.3f"), f(g("fetch_bytes")), f(g("write_bytes")), f(g("algo_bytes")), f(g("simd_cycles")),
            f(e.get("valu_issue_floor"), "%.3f"), f(e.get("valu_issue_priced"), "%.3f"), f(e.get("sq_wait_any_over_wave_cycles"), "%.2f"),
            f(e.get("sq_wait_inst_any_over_wave_cycles"), "%.2f"), f(e.get("lds_cycles_per_lds_inst"), "%.2f"), f(e.get("lds_bank_conflict_frac"), "%.2f")))
    o.append("")
print("\n".join(o))

#!/usr/bin/env python3
"""Assembles profiles/r03_round_end.md from the outputs of `bash profiles/collect_r03_final.sh <tag>` (gpurun_out/<tag>) and the counter
summary profiles/r03_pmc.json (profiles/collect_r03_pmc.sh + pmc_summary_r03.py).
usage: make_r03_summary.py [tag] > profiles/r03_round_end.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03fin"
A = os.path.join(ROOT, "gpurun_out", TAG)


def last(f):
    p = os.path.join(A, f)
    if not os.path.exists(p):
        return None
    ls = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
    return ls[-1] if ls else None


def block(f):
    l = last(f)
    return "```json\n" + (l or "(not collected)") + "\n```\n"


o = []
o.append("# Round 3 -- end-of-round measurement set (MI355X, 1 GPU)\n")
o.append("Collected through gpurun with `bash profiles/collect_r03_final.sh %s` (GPU tests, the default bench line with its folded eventalign and "
         "variants legs, the from-raw line, the CPU plumbing line, the 2-rank gloo rehearsal, the reference-side batched binding, kernel traces); "
         "counters from `bash profiles/collect_r03_pmc.sh` (`r03_pmc.json`).  Raw outputs live under `gpurun_out/` (scratch); this file is "
         "`profiles/make_r03_summary.py %s`.  Beside it: `r03_kernel_a_split.md` (kernel A's back-track as its own launch, the pipelined pass, the "
         "14-instruction walk step), `r03_soak.md` (4 500 indel records against the reference itself), `r03_rccl_probe.md`, `r03_pmc.json` "
         "(+ `r03_pmc_mid_round.json`, before the walk's trim and with the one-read chain kernel); the chain-kernel log, the kernel-B ablations and the "
         "dropped experiments are at the end of this file.\n" % (TAG, TAG))
pl = os.path.join(A, "pytest.log")
if os.path.exists(pl):
    ps = [l for l in open(pl).read().splitlines() if " passed" in l or " failed" in l]
    o.append("GPU tests on the same box (`python -m pytest tests -m gpu -q`): `%s`\n" % (ps[-1].strip() if ps else "?"))
pr = os.path.join(A, "probe.log")
if os.path.exists(pr):
    ls = [l for l in open(pr).read().splitlines() if l.startswith("probe:")]
    if ls:
        o.append("Hardware probe at `np_create` (`np_ctx_info`): `%s`\n" % ls[-1])
o.append("## Bench lines\n")
o.append("Default (`python bench.py --steps 5 --warmup 1`): BASELINE.json configs[1] -- 100 000 reads per step (20 000 distinct x 5), ~8k events each; "
         "`value` resident, `value_streamed` host-fed, `value_ragged` log-normal lengths with its parity check; `value_eventalign` (configs[2]) and "
         "`value_variants` (configs[3]) folded in, each with parity fields and a roofline of its own:\n")
o.append(block("bench_default.json"))
d = last("bench_default.json")
if d:
    d = json.loads(d)
    r = d["roofline"]
    o.append("| | |\n|---|---|")
    o.append("| value | %.0f reads/s, %.1f ms per step (kernel A %.1f, kernel B %.1f, glue + work items %.1f) |" % (
        d["value"], d["ms_per_step"], r["kernel_ms_per_step"]["event_align"], r["kernel_ms_per_step"]["hmm_score"], r["kernel_ms_per_step"]["glue_and_work_items"]))
    o.append("| roofline (kernel A) | %.1f GB algorithmic / %.2f ms = %.0f GB/s = **%.4f** of 8 TB/s; traffic %.1f GB per launch (counter bytes per band x this run's bands) |" % (
        r["algo_bytes_per_launch"] / 1e9, r["avg_launch_ms"], r["achieved"], r["frac"], (r["traffic"] or 0) / 1e9))
    o.append("| streamed / ragged | %.0f / %.0f reads/s; ragged parity: %s |" % (d.get("value_streamed") or 0, d.get("value_ragged") or 0, json.dumps((d.get("ragged") or {}).get("check"))))
    ea, va = d.get("eventalign") or {}, d.get("variants") or {}
    if "value" in ea:
        o.append("| eventalign leg | %.0f reads/s, %.1f ms per 50 000 reads, chain kernel %.1f ms, roofline frac %.4f, cpu %s |" % (
            ea["value"], ea["ms_per_step"], ea["kernel_ms_per_step"]["eventalign_chain"], ea["roofline"]["frac"], json.dumps(ea.get("cpu_baseline"))))
    if "value" in va:
        o.append("| variants leg | %.0f calls/s, %.1f ms per step, roofline frac %.4f, cpu %s |" % (va["value"], va["ms_per_step"], va["roofline"]["frac"], json.dumps(va.get("cpu_baseline"))))
    o.append("| cpu_baseline | %s |" % json.dumps({k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "t1_value") if k in d["cpu_baseline"]}))
    o.append("")
o.append("From raw signal (`--from-raw 1 --steps 3 --cpu-sample 256 --legs 0`):\n")
o.append(block("bench_from_raw.json"))
o.append("configs[0] plumbing line (`--workload cpu-t1 --cpu-sample 200`: the reference's code, one host thread, no GPU):\n")
o.append(block("bench_cpu_t1.json"))
o.append("2-rank rehearsal on ONE MI355X (`NP_BENCH_BACKEND=gloo python bench.py --gpus 2 --pool 4000 --tile 5 ...`: both ranks share the device, the site "
         "table is all-reduced over gloo; a rehearsal of the launcher and of the `per_rank` diagnostics, not a scaling measurement):\n")
o.append(block("bench_2rank_gloo.json"))
o.append("## The reference-side batched binding (`tests/bench_batch_dropin.py`)\n")
o.append("BAM records + raw signal in host memory -> `NpBatchPipeline` (submit / collect, two batches in flight) -> the reference's `ScoredSite` maps; "
         "`pipelined_adc`: int16 samples; `sync`: one batch at a time; `value_binding_only` excludes the harness's own per-batch work.  16 CPUs (cgroup quota).\n")
bp = os.path.join(A, "batch_dropin.json")
if os.path.exists(bp):
    o.append("| records per batch | pipelined reads/s | int16 samples | synchronous | host ms per batch inside the binding (float / int16) | waiting for the device |")
    o.append("|---|---|---|---|---|---|")
    for l in open(bp).read().splitlines():
        if not l.startswith("{"):
            continue
        b = json.loads(l)
        o.append("| %d | %.0f | %.0f | %.0f | %.1f / %.1f | %.1f / %.1f |" % (
            b["batch_size"], b["pipelined"]["value"], b["pipelined_adc"]["value"], b["sync"]["value"],
            b["pipelined"]["host_ms_per_batch"]["inside_binding"], b["pipelined_adc"]["host_ms_per_batch"]["inside_binding"],
            b["pipelined"]["host_ms_per_batch"]["wait_device"], b["pipelined_adc"]["host_ms_per_batch"]["wait_device"]))
    o.append("")
o.append("## rocprofv3 --kernel-trace --stats\n")
o.append("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --legs 0` (4 launches of the step incl. "
         "warm-up; the bench's CPU-parity sample is off, so every `np_event_align_kernel` launch is the 100 000-read step), `profiles/summarize_rocpd.py`:\n")
for f in ("trace.md",):
    p = os.path.join(A, f)
    o.append(open(p).read() if os.path.exists(p) else "(not collected)")
o.append("\nThe eventalign leg (`python bench.py --workload eventalign --steps 3 --warmup 1`):\n")
p = os.path.join(A, "trace_eventalign.md")
o.append(open(p).read() if os.path.exists(p) else "(not collected)")
# counters
pm = os.path.join(ROOT, "profiles", "r03_pmc.json")
if os.path.exists(pm):
    P = json.load(open(pm))
    o.append("\n## Counters of the shipped kernels (`profiles/r03_pmc.json`)\n")
    o.append("`rocprofv3 --kernel-trace --pmc <set> -- python tools/pmc_workload.py --reads 8192 --ea-reads 8192 --reps 2`, one counter set per pass "
             "(`profiles/collect_r03_pmc.sh`; passes over 100 000-read launches do not finish on this pool).  FETCH_SIZE x2 / WRITE_SIZE x1 as the guide's "
             "gfx950 corrections prescribe; busy percentages are counter cycles over the UNPROFILED launch time x SIMDs (a counter pass slows these kernels "
             "~1.7x, so compare the per-unit instruction counts first).\n")
    o.append("| kernel | unit | vector instr / unit | scalar / unit | LDS / unit | HBM fetched + written B / unit | algorithmic B / unit | VALU busy % | SALU busy % | LDS cycles per LDS instr (conflict share) |")
    o.append("|---|---|---|---|---|---|---|---|---|---|")
    for key, unit, u in (("event_align", "band", "band"), ("hmm_forward", "call", "call"), ("chain", "segment", "segment")):
        e = P.get(key)
        if not e:
            continue
        g = lambda n: e.get("%s_per_%s" % (n, u))
        f = lambda v, fmt="%.1f": (fmt % v) if isinstance(v, (int, float)) else "-"
        o.append("| %s | %s | %s | %s | %s | %s + %s | %s | %s | %s | %s (%s) |" % (
            key, unit, f(g("valu")), f(g("salu")), f(g("lds"), "%.3f"), f(g("fetch_bytes")), f(g("write_bytes")), f(g("algo_bytes")),
            f(e.get("valu_busy_pct")), f(e.get("salu_busy_pct")), f(e.get("lds_cycles_per_lds_inst"), "%.2f"), f(e.get("lds_bank_conflict_frac"), "%.2f")))
    o.append("")
ex = os.path.join(ROOT, "profiles", "r03_experiments_tail.md")
if os.path.exists(ex):
    o.append(open(ex).read())
print("\n".join(o))

#!/usr/bin/env python3
"""profiles/r06_boxes.md: the same commands on the round's boxes (ranges next to every single-box figure).
usage: make_r06_boxes.py tag [tag ...] > profiles/r06_boxes.md   (tags: gpurun_out/<tag> of profiles/collect_r06_final.sh)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last(tag, f):
    p = os.path.join(ROOT, "gpurun_out", tag, f)
    if not os.path.exists(p):
        return None
    ls = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None


tags = sys.argv[1:]
rows = []
for t in tags:
    d = last(t, "bench_default.json")
    if not d:
        continue
    k = d["roofline"]["kernel_ms_per_step"]
    fr, ea = d.get("from_raw") or {}, d.get("eventalign") or {}
    g = last(t, "bench_250k.json") or {}
    rows.append(dict(tag=t, value=d["value"], ms=d["ms_per_step"], a=k["event_align"], b=k["hmm_score"], glue=k["glue_and_work_items"], frac_issue=d["roofline"]["frac"],
                     frac_hbm=d["roofline"]["hbm"]["frac"], streamed=d.get("value_streamed"), ragged=d.get("value_ragged"), from_raw=d.get("value_from_raw"),
                     detect=(fr.get("kernel_ms_per_step") or {}).get("event_detect"), ea=d.get("value_eventalign"), chain=(ea.get("kernel_ms_per_step") or {}).get("eventalign_chain"),
                     variants=d.get("value_variants"), b512=d.get("value_binding_512"), b8192=d.get("value_binding_8192"), cpu=(d.get("cpu_baseline") or {}).get("value"),
                     genome=g.get("value"), genome_sites=(g.get("site_table") or {}).get("sites")))
o = ["## Round 6: the driver's command and configs[4]'s shape on the round's boxes (`profiles/collect_r06_final.sh`, one gpurun call = one box each)\n",
     "Every row is `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) on the round's final code; `genome 250k` is "
     "`python bench.py --gpus 1 --genome 1 --pool 50000 --tile 5 --steps 5 --warmup 2` in the same call.\n",
     "| call | value (reads/s) | ms per step | kernel A / B / glue (ms) | roofline.frac (vector issue) / hbm.frac | streamed | ragged | from raw (detector ms) | eventalign (chain ms) | variants (calls/s) | binding 512 / 8 192 | CPU reference, 16 cores | genome 250k (keys in the table) |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
f = lambda v, fmt="%.0f": (fmt % v) if isinstance(v, (int, float)) else "-"
for r in rows:
    o.append("| %s | %s | %s | %s / %s / %s | %s / %s | %s | %s | %s (%s) | %s (%s) | %s | %s / %s | %s | %s (%s) |" % (
        r["tag"], f(r["value"]), f(r["ms"], "%.1f"), f(r["a"], "%.1f"), f(r["b"], "%.1f"), f(r["glue"], "%.2f"), f(r["frac_issue"], "%.3f"), f(r["frac_hbm"], "%.4f"),
        f(r["streamed"]), f(r["ragged"]), f(r["from_raw"]), f(r["detect"], "%.1f"), f(r["ea"]), f(r["chain"], "%.1f"), f(r["variants"], "%.3g"), f(r["b512"]), f(r["b8192"]),
        f(r["cpu"], "%.1f"), f(r["genome"]), f(r["genome_sites"], "%d")))
if rows:
    rng = lambda k, fmt="%.0f": "%s - %s" % (f(min(r[k] for r in rows if r[k] is not None), fmt), f(max(r[k] for r in rows if r[k] is not None), fmt))
    o.append("| **range** | **%s** | %s | %s / %s / %s | %s / %s | %s | %s | %s (%s) | %s (%s) | %s | %s / %s | %s | %s |" % (
        rng("value"), rng("ms", "%.1f"), rng("a", "%.1f"), rng("b", "%.1f"), rng("glue", "%.2f"), rng("frac_issue", "%.3f"), rng("frac_hbm", "%.4f"), rng("streamed"), rng("ragged"),
        rng("from_raw"), rng("detect", "%.1f"), rng("ea"), rng("chain", "%.1f"), rng("variants", "%.3g"), rng("b512"), rng("b8192"), rng("cpu", "%.1f"), rng("genome")))
print("\n".join(o))

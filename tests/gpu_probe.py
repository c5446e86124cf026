"""Step-by-step GPU probe with flushed prints (debug aid; not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
t0 = time.time()
def log(*a):
    print("[%.1fs]" % (time.time() - t0), *a, flush=True)
import numpy as np
log("import torch"); import torch; log("torch ok", torch.cuda.is_available(), torch.cuda.get_device_name(0))
from oracle import Oracle, load_models
from nanopolish_amd.api import Context
from cases import *
models = load_models(); orc = Oracle(); log("oracle ok")
ctx = Context(0); log("ctx ok")
ctx.register_model(models["nucleotide"], "nucleotide"); ctx.register_model(models["cpg"], "cpg"); log("models ok")
mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
rd = synth_read(0, models["nucleotide"], L=400)
sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
want = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"]); log("oracle align", len(want))
S = orc.scalings(rd["shift"], rd["scale"], rd["var"])
epb, mj = methylation_jobs(orc, rd, want)
j = mj[0]
ranks = orc.sequence_kmer_ranks("cpg", j["subseq"], j["rc_subseq"], K, j["rc"])
w = orc.hmm_score(mc, S, rd["events"], ranks, j["e1"], j["e2"], j["stride"], epb, 1.0, 3)
log("score job...")
got = ctx.profile_hmm_score([dict(events=rd["events"], ranks=ranks, e_start=j["e1"], e_stop=j["e2"], stride=j["stride"], model=ctx.models["cpg"], scale=rd["scale"], shift=rd["shift"], var=rd["var"], events_per_base=epb, flags=3)])
log("hmm_score", got, w)
log("event align...")
g = ctx.adaptive_banded_simple_event_align([dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0)])[0]
log("event_align", len(g), np.array_equal(g, want))
epb2, segs = eventalign_segments(orc, rd, want)
sg = segs[0]; r2 = orc.sequence_kmer_ranks("nucleotide", sg["seq"], None, K, 0)
wa = orc.hmm_align(mn, S, rd["events"], r2, sg["e1"], sg["e2"], 1, epb2)
log("hmm align...")
ga = ctx.profile_hmm_align([dict(events=rd["events"], ranks=r2, e_start=sg["e1"], e_stop=sg["e2"], stride=1, model=ctx.models["nucleotide"], scale=rd["scale"], shift=rd["shift"], var=rd["var"], events_per_base=epb2, flags=0)])[0]
log("hmm_align", len(ga[0]), all(np.array_equal(a, b) for a, b in zip(ga, wa)))

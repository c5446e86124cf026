"""Debug helper (not a test): event-align parity of a few reads against the oracle, for A/B library builds (NP_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import Oracle, load_models
from nanopolish_amd.api import Context
from nanopolish_amd.synth import synth_read
models = load_models(); orc = Oracle(); mn = orc.model(models["nucleotide"])
ctx = Context(0); ctx.register_model(models["nucleotide"], "nucleotide")
bad = 0
for rid, L in [(0, 700), (1, 700), (12, 2500), (16, 130), (3, 5450), (5, 300)]:
    rd = synth_read(rid, models["nucleotide"], L=L)
    sh, sc = orc.estimate_scalings_mom(mn, rd["ranks"], rd["events"])
    want = orc.event_align(mn, orc.scalings(sh, sc, 1.0), rd["events"], rd["ranks"])
    got = ctx.adaptive_banded_simple_event_align([dict(events=rd["events"], ranks=rd["ranks"], model=ctx.models["nucleotide"], scale=sc, shift=sh, var=1.0)])[0]
    same = got.shape == want.shape and np.array_equal(got, want)
    first = -1
    if not same and len(got) and len(want):
        n = min(len(got), len(want)); d = np.nonzero((got[:n] != want[:n]).any(axis=1))[0]
        first = int(d[0]) if len(d) else n
    print(rid, L, "OK" if same else "MISMATCH got %d want %d first diff at pair %d: %s vs %s" % (len(got), len(want), first, got[first:first+3].tolist() if first >= 0 else None, want[first:first+3].tolist() if first >= 0 else None))
    bad += not same
print("bad", bad)

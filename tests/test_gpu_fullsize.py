"""BASELINE-size reads (~8k events) through the fused device pass: size-independent properties of the results plus a
spot check against the oracle."""
import numpy as np
import pytest

from cases import call_methylation_read

pytestmark = pytest.mark.gpu


def test_full_size_batch_properties(ctx, orc, models):
    from nanopolish_amd.pipeline import build_host_batch, tile_host_batch, CallMethylationBatch
    pool, tile = 96, 4
    hb = build_host_batch(models, list(range(1000, 1000 + pool)), L=5450)
    batch = CallMethylationBatch(ctx, tile_host_batch(hb, tile), "cuda:0")
    batch.step()
    scores = batch.scores()
    n_groups = len(scores) // 2 // tile
    # replication invariance: every independent HBM copy of a read gives bit-identical results
    for t in range(1, tile):
        assert np.array_equal(scores[:2 * n_groups], scores[2 * n_groups * t:2 * n_groups * (t + 1)], equal_nan=True)
    n_pairs = batch.d_n_pairs.cpu().numpy()
    assert (n_pairs > 0).all()
    for r in range(0, pool, 7):
        p = batch.pairs_of(r)
        E, Kk = len(hb["reads"][r]["events"]), len(hb["reads"][r]["ranks"])
        assert np.array_equal(p, batch.pairs_of(r + pool * (tile - 1)))
        assert p[0, 0] == 0 and p[-1, 0] == Kk - 1 and p[-1, 1] < E and p[0, 1] >= 0      # spans k-mer 0 .. K-1
        d = np.diff(p, axis=0)
        assert set(map(tuple, np.unique(d, axis=0))) <= {(1, 1), (0, 1), (1, 0)}          # D, U, L moves only
        assert len(p) <= E + Kk
    llr = scores[1::2].astype(np.float64) - scores[0::2]
    ok = np.isfinite(llr)
    assert ok.mean() > 0.9 and np.abs(llr[ok]).max() < 500
    # oracle spot check on BASELINE-size reads
    mn = orc.model(models["nucleotide"]); mc = orc.model(models["cpg"])
    g0 = 0
    for i in range(3):
        want = call_methylation_read(orc, mn, mc, hb["reads"][i])
        assert np.array_equal(batch.pairs_of(i), want["pairs"])
        firsts = list(hb["meta"][i]["first"])
        for f, u, m in zip(want["first"], want["unmeth"], want["meth"]):
            g = g0 + firsts.index(f)
            assert scores[2 * g] == u and scores[2 * g + 1] == m
        g0 += len(firsts)

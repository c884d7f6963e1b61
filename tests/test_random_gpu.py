"""GPU: the HIP path against the oracle on RANDOM inputs (hypothesis), through the C ABI — the same strategies as
tests/test_oracle_random.py: special floats (zero, denormal, inf, NaN, negative), empty maps, up to 40 tokens
(the u32 mask wraps at 32), thresholds, OMC maps; bit-exact ids, scores and counts."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oramacore_amd as oa
from oracle import oracle as orc
from oramacore_amd import fulltext as ft
from test_oracle_random import contributions, two_maps

pytestmark = pytest.mark.gpu
CTX = None


def context():
    global CTX
    if CTX is None:
        CTX = oa.Context(0)
    return CTX


def bits(x):
    """Bit patterns, with -0.0 folded onto +0.0: K4's order-preserving key canonicalises the sign of zero so that
    +-0 tie exactly like `NotNan<f32>` compares them, and the value is rebuilt from the key — a score of -0.0 comes
    back as +0.0 (declared deviation; the scoring pipeline itself never produces -0.0)."""
    a = np.asarray(x, dtype=np.float32).copy()
    a[a == 0] = 0.0
    return a.view(np.uint32).tolist()


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(contributions(), st.integers(1, 50), st.dictionaries(st.integers(0, 60), st.sampled_from([0.25, 0.5, 2.0, 10.0]),
                                                           max_size=4))
def test_bm25_score_random(c, top_k, omc):
    entries, n_tokens, n_docs, thr = c
    if not entries:
        entries = [(0, np.zeros(0, np.uint64), np.zeros(0, np.float32))]
    ids, sc, count = ft.bm25_score(context(), entries, n_tokens, float(n_docs), top_k, thr, omc=omc or None)
    od, os_ = orc.search_full_text(entries, n_tokens, float(n_docs), 1.2, thr)
    if omc:
        os_ = orc.apply_omc(od, os_, sorted(omc), [omc[d] for d in sorted(omc)])
    td, ts = orc.top_n(od, os_, top_k)
    assert count == len(od)
    assert ids.tolist() == td.tolist()
    assert bits(sc) == bits(ts)


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(two_maps())
def test_hybrid_combine_and_top_n_random(m):
    vec, ftm, n = m
    n = max(n, 1)
    ids, sc, count = ft.hybrid_combine(context(), vec, ftm, n)
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), list(ftm), list(ftm.values()))
    td, ts = orc.top_n(od, os_, n)
    assert count == len(od)
    assert ids.tolist() == td.tolist()
    assert bits(sc) == bits(ts)
    if ftm:
        i2, s2 = ft.top_n(context(), ftm, n)
        d2, v2 = orc.top_n(sorted(ftm), [ftm[d] for d in sorted(ftm)], n)
        assert i2.tolist() == d2.tolist() and bits(s2) == bits(v2)

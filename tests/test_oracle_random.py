"""CPU: the C oracle against the independent numpy restatement (tests/golden/make_golden.py) on RANDOM inputs —
the goldens pin fixed cases, this pins the two restatements to each other bit for bit over the input space
(special values included: zeros, denormals, infinities, NaN, negative scores, empty maps, > 32 tokens)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import util
from oracle import oracle as orc

F = np.float32
mg = util.mg

special = [0.0, -0.0, 1e-45, 1e-39, 1.1754944e-38, 3.4e38, float("inf"), float("nan"), 1.0, 0.5]
ntf_val = st.one_of(st.sampled_from(special), st.floats(min_value=0.0, max_value=50.0, width=32))
score_val = st.one_of(st.sampled_from(special + [-1.0, -0.25]), st.floats(min_value=-2.0, max_value=30.0, width=32))


@st.composite
def contributions(draw):
    n_tokens = draw(st.integers(1, 40))
    n_entries = draw(st.integers(0, 12))
    entries = []
    for _ in range(n_entries):
        tok = draw(st.integers(0, n_tokens - 1))
        docs = sorted(draw(st.sets(st.integers(0, 60), max_size=20)))
        ntf = [draw(ntf_val) for _ in docs]
        entries.append((tok, np.array(docs, dtype=np.uint64), np.array(ntf, dtype=np.float32)))
    n_docs = draw(st.integers(1, 200))
    thr = draw(st.one_of(st.none(), st.integers(0, 5)))
    return entries, n_tokens, n_docs, thr


@pytest.fixture(autouse=True)
def same_libm_idf(monkeypatch):
    """ln_1p is the one libm call on the path.  numpy's float32 log1p and glibc's log1pf differ by 1 ulp on some
    arguments (e.g. log1p(1/3)); the oracle uses glibc's (what Rust's f32::ln_1p lowers to on Linux) and is pinned
    by the reference's known-answer tests, so the structural cross-check below borrows that one function."""
    monkeypatch.setattr(mg, "idf", lambda n_docs, df: orc.bm25_idf(float(n_docs), int(df)))


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32).tolist()


@settings(max_examples=300, deadline=None)
@given(contributions())
def test_search_full_text_restatements_agree(c):
    entries, n_tokens, n_docs, thr = c
    od, os_ = orc.search_full_text(entries, n_tokens, float(n_docs), 1.2, thr)
    exp = mg.search_full_text([(t, d.tolist(), v.tolist()) for t, d, v in entries], n_tokens, n_docs, 1.2, thr)
    assert od.tolist() == sorted(exp)
    assert bits(os_) == bits([exp[d] for d in sorted(exp)])


@st.composite
def two_maps(draw):
    vd = sorted(draw(st.sets(st.integers(0, 40), max_size=10)))
    fd = sorted(draw(st.sets(st.integers(0, 40), max_size=25)))
    return ({d: F(draw(score_val)) for d in vd}, {d: F(draw(score_val)) for d in fd}, draw(st.integers(0, 30)))


@settings(max_examples=300, deadline=None)
@given(two_maps())
def test_combine_and_top_n_restatements_agree(m):
    vec, ftm, n = m
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), list(ftm), list(ftm.values()))
    exp = mg.normalize_and_combine(vec, ftm)
    assert od.tolist() == sorted(exp)
    assert bits(os_) == bits([exp[d] for d in sorted(exp)])
    td, ts = orc.top_n(od, os_, n)
    etop = mg.top_n(exp, n)
    assert td.tolist() == [d for d, _ in etop]
    assert bits(ts) == bits([s for _, s in etop])

"""Term-dictionary expansion on the device (orama_dict_*) against a plain host restatement:
exact / prefix / Levenshtein <= tolerance over a sorted term list."""
import numpy as np
import pytest

import oramacore_amd as oa
from oramacore_amd import fulltext as ft
from util import levenshtein_le

pytestmark = pytest.mark.gpu


def host_expand(terms, token, exact, tolerance):
    out = []
    for i, t in enumerate(terms):
        if exact:
            hit = t == token
        else:
            hit = t.startswith(token) or (tolerance > 0 and abs(len(t) - len(token)) <= tolerance
                                           and levenshtein_le(t, token, tolerance))
        if hit:
            out.append(i)
    return out


def test_expand_matches_host_restatement(ctx):
    rng = np.random.default_rng(3)
    alphabet = "abcdefghij"
    terms = set()
    while len(terms) < 30000:
        n = int(rng.integers(1, 12))
        terms.add("".join(alphabet[int(c)] for c in rng.integers(0, len(alphabet), size=n)))
    terms = sorted(terms)
    d = ft.TermDictionary(ctx, terms)
    tokens = ["a", "abc", "jjj", "abcdefghij", "bad", "cafe", "zzz", "", terms[100], terms[29999], terms[5][:-1] + "x"]
    for token in tokens:
        for exact in (True, False):
            for tol in (0, 1, 2):
                got = d.expand(token, exact=exact, tolerance=tol)
                exp = host_expand(terms, token, exact, tol)
                assert got == exp, (token, exact, tol, len(got), len(exp))
    d.close()


def test_expand_edge_cases(ctx):
    d = ft.TermDictionary(ctx, ["main", "maple", "street", "streets", "strut"])
    assert d.expand("mxin", tolerance=1) == [0]              # src/tests/fulltext_search.rs:956-1018
    assert d.expand("msple", tolerance=1) == [1]
    assert d.expand("mxin") == []
    assert d.expand("str") == [2, 3, 4]                      # prefix (fulltext_search.rs:603-753)
    assert d.expand("street", exact=True) == [2]
    assert d.expand("stret", tolerance=1) == [2, 4]          # "street" (one deletion), "strut" (one substitution)
    assert d.expand("stret", tolerance=2) == [2, 3, 4]
    with pytest.raises(oa.OramaError):
        d.expand("x" * 65)
    empty = ft.TermDictionary(ctx, [])
    assert empty.expand("a", tolerance=2) == []
    with pytest.raises(ValueError):
        ft.TermDictionary(ctx, ["b", "a"])
    d.close()
    empty.close()

"""Shared helpers for the tests: deterministic inputs (same generators as tests/golden/make_golden.py)
and tolerance-aware comparison of ranked lists."""
from __future__ import annotations

import importlib.util
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

_spec = importlib.util.spec_from_file_location("make_golden", GOLDEN / "make_golden.py")
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)

det_matrix = mg.det_matrix
hash_u64 = mg.hash_u64


def load_json(name: str):
    return json.loads((GOLDEN / name).read_text())


def gaussian_rows(n: int, d: int, seed: int, scale_rows: bool = True) -> np.ndarray:
    """SURVEY §8d synthetic embeddings: N(0,1), L2-normalised, re-scaled by u ~ U(0.5, 2)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    if scale_rows:
        x *= rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def assert_ranked_equal(got_ids, got_val, exp_ids, exp_val, tol: float, what: str = ""):
    """Two ranked lists agree up to fp32 summation-order noise: values within `tol` position by
    position; ids identical except inside groups of near-ties (consecutive gaps <= 2*tol), where the
    same ids may appear in a different order.  A near-tie group cut by the end of the list (k) may
    also differ in membership, because its other members lie beyond k.
    (The HIP kernel reduces each dot product as a tree, the oracle sequentially — SURVEY §7.)"""
    got_ids, exp_ids = np.asarray(got_ids), np.asarray(exp_ids)
    got_val, exp_val = np.asarray(got_val, dtype=np.float64), np.asarray(exp_val, dtype=np.float64)
    assert got_ids.shape == exp_ids.shape, f"{what}: length {got_ids.shape} vs {exp_ids.shape}"
    if len(exp_ids) == 0:
        return
    err = np.max(np.abs(got_val - exp_val))
    assert err <= tol, f"{what}: values differ by {err} > {tol}"
    n = len(exp_ids)
    i = 0
    while i < n:
        j = i + 1
        while j < n and abs(exp_val[j] - exp_val[j - 1]) <= 2 * tol:
            j += 1
        if j < n:
            assert sorted(got_ids[i:j].tolist()) == sorted(exp_ids[i:j].tolist()), (
                f"{what}: ids differ in rank range [{i},{j}): {got_ids[i:j]} vs {exp_ids[i:j]}")
        i = j


def assert_topk_sound(got_rows, got_val, all_val, k: int, tol: float, what: str = ""):
    """Soundness of a top-k (smaller value = better) against the oracle's full value array:
    sorted ascending, each reported value within tol of the oracle's value for that row, and no row
    that is better than the reported k-th value by more than 2*tol is missing."""
    got_rows = np.asarray(got_rows, dtype=np.int64)
    got_val = np.asarray(got_val, dtype=np.float64)
    all_val = np.asarray(all_val, dtype=np.float64)
    valid = ~np.isnan(all_val)
    assert len(got_rows) == min(k, int(valid.sum())), f"{what}: count {len(got_rows)}"
    if len(got_rows) == 0:
        return
    assert np.all(np.diff(got_val) >= 0), f"{what}: not sorted"
    assert len(set(got_rows.tolist())) == len(got_rows), f"{what}: duplicate rows"
    err = np.max(np.abs(all_val[got_rows] - got_val))
    assert err <= tol, f"{what}: value error {err} > {tol}"
    kth = got_val[-1]
    must = np.nonzero(valid & (all_val < kth - 2 * tol))[0]
    missing = set(must.tolist()) - set(got_rows.tolist())
    assert not missing, f"{what}: rows clearly inside the top-k are missing: {sorted(missing)[:5]}"


def levenshtein_le(a: str, b: str, k: int) -> bool:
    """Levenshtein(a, b) <= k (plain DP; dictionary terms are short)."""
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        if min(cur) > k:
            return False
        prev = cur
    return prev[-1] <= k

"""Parity helpers shared by the CPU and GPU test suites.

Tie contract (oracle/bm25_oracle.c): the reference leaves the order of exactly equal
fp32 scores to NumPy's introselect/quicksort.  A result list therefore matches when
  * the score sequence is bit-identical,
  * inside every maximal run of equal scores the doc ids match as SETS, and
  * for the last run, if the list is full (count == k) the tie group may straddle rank k: then every
    doc we return in that run must be a doc whose true score equals the run's score (checked against
    the full score vector when one is supplied).
"""
import numpy as np


def runs(scores):
    out, s = [], 0
    for i in range(1, len(scores) + 1):
        if i == len(scores) or scores[i] != scores[s]:
            out.append((s, i))
            s = i
    return out


def assert_ranked_equal(got_docs, got_scores, exp_docs, exp_scores, k=None, full_scores=None, label=""):
    got_docs = np.asarray(got_docs)
    exp_docs = np.asarray(exp_docs)
    gs = np.asarray(got_scores, dtype=np.float32)
    es = np.asarray(exp_scores, dtype=np.float32)
    assert len(gs) == len(es), f"{label}: count {len(gs)} != {len(es)}"
    assert np.array_equal(gs.view(np.uint32), es.view(np.uint32)), f"{label}: scores differ\n{gs}\n{es}"
    assert np.all(gs[:-1] >= gs[1:]), f"{label}: not descending"
    rr = runs(es)
    for n, (a, b) in enumerate(rr):
        last_full = (n == len(rr) - 1) and (k is not None) and (len(es) == k)
        if not last_full:
            assert set(got_docs[a:b].tolist()) == set(exp_docs[a:b].tolist()), f"{label}: docs differ in run {a}:{b}"
        else:
            assert len(set(got_docs[a:b].tolist())) == b - a, f"{label}: duplicate docs in boundary run"
            if full_scores is not None:
                fs = np.asarray(full_scores, dtype=np.float32)
                for d in got_docs[a:b]:
                    assert fs[int(d)] == es[a], f"{label}: doc {d} in boundary tie run has score {fs[int(d)]} != {es[a]}"
    assert len(set(got_docs.tolist())) == len(got_docs), f"{label}: duplicate docs"


def assert_canonical_order(docs, scores, label=""):
    """Our own contract: (score desc, doc asc)."""
    d = np.asarray(docs, dtype=np.int64)
    s = np.asarray(scores, dtype=np.float32)
    for i in range(1, len(s)):
        assert s[i - 1] > s[i] or (s[i - 1] == s[i] and d[i - 1] < d[i]), f"{label}: order violated at {i}"

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


def np_build_blocks(indptr, indices, data, n_docs, vocab, tile_log2, unit_tiles, val_dtype=np.float32, block_pad=256):
    """NumPy restatement of the blocked posting layout (include/sparse_rx.h, srx_index_desc) from a doc-major CSR whose
    `data` are already the values to store.  Returns (term_ptr i64[V+1], post i32[(n_blocks+pad)*words],
    tile_skip i32[V*(n_tiles+1)], n_blocks).  Slow, small cases only: the checker of srx_build_blocks."""
    from scipy.sparse import csr_matrix
    m = csr_matrix((np.asarray(data, np.float32), np.asarray(indices), np.asarray(indptr)), shape=(n_docs, vocab)).tocsc()
    m.sort_indices()
    G = 1 << tile_log2
    n_tiles = (n_docs + G - 1) >> tile_log2
    U = unit_tiles * G
    n_units = (n_tiles + unit_tiles - 1) // unit_tiles
    words = 8 if val_dtype == np.float32 else 6
    term_ptr = np.zeros(vocab + 1, np.int64)
    skip = np.zeros((vocab, n_tiles + 1), np.int32)
    docs_out, vals_out = [], []
    pos = 0
    for t in range(vocab):
        term_ptr[t] = pos
        docs = m.indices[m.indptr[t]:m.indptr[t + 1]].astype(np.int64)
        vals = m.data[m.indptr[t]:m.indptr[t + 1]]
        start = pos
        unit_start = {}
        for u in range(n_units):
            sel = (docs >= u * U) & (docs < (u + 1) * U)
            unit_start[u] = pos - start
            d, v = docs[sel], vals[sel]
            padn = (-len(d)) % 4
            docs_out.append(np.concatenate([d, np.full(padn, -1 - 32 * (t % 64), np.int64)]))
            vals_out.append(np.concatenate([v, np.zeros(padn, np.float32)]))
            pos += len(d) + padn
        unit_start[n_units] = pos - start
        for j in range(n_tiles + 1):
            if j == n_tiles:
                skip[t, j] = pos - start
            else:
                u = j // unit_tiles
                skip[t, j] = unit_start[u] + int(((docs >= u * U) & (docs < j * G)).sum())
    term_ptr[vocab] = pos
    n_blocks = pos // 4
    dd = np.concatenate(docs_out + [np.repeat(-1 - 32 * (np.arange(block_pad, dtype=np.int64) % 64), 4)]).astype(np.int32).reshape(-1, 4)
    vv = np.concatenate(vals_out + [np.zeros(4 * block_pad, np.float32)]).astype(val_dtype).reshape(-1, 4)
    post = np.zeros((n_blocks + block_pad, words), np.int32)
    post[:, :4] = dd
    post[:, 4:] = vv.view(np.int32).reshape(n_blocks + block_pad, -1)
    return term_ptr, post.reshape(-1), skip.reshape(-1), n_blocks


def np_compact_blocks(post, unit_docs, val_dtype=np.float32):
    """NumPy restatement of the compact copy the tier-1 kernel streams (include/sparse_rx.h, srx_build_compact): per block
    four 16-bit unit-local doc ids (doc mod unit_docs; a sentinel -1 - 32 x becomes 49152 + 32 (x mod 64)) packed in two
    words, then the value words unchanged."""
    words = 8 if val_dtype == np.float32 else 6
    b = np.asarray(post, np.int32).reshape(-1, words)
    d = b[:, :4].astype(np.int64)
    loc = np.where(d >= 0, d % unit_docs, 49152 + 32 * (((-1 - d) >> 5) & 63)).astype(np.uint32)
    out = np.zeros((b.shape[0], words - 2), np.int32)
    out[:, 0] = (loc[:, 0] | (loc[:, 1] << 16)).astype(np.uint32).view(np.int32)
    out[:, 1] = (loc[:, 2] | (loc[:, 3] << 16)).astype(np.uint32).view(np.int32)
    out[:, 2:] = b[:, 4:]
    return out.reshape(-1)

"""Dev probe: a unit too large for the compact copy (tier 2 serves everything) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import sparse_rx as rx
from sparse_rx import synth
big = synth.uniform_corpus_np(70_000, 500, 6, seed=5)
q = synth.queries_np(16, big.vocab, 4, seed=6)
idf = np.ones(big.vocab, np.float32)
ed, es, ec = oracle.search_batch(big.indptr, big.indices, big.data, None, idf, q[0], q[1], q[2], 10, mode=oracle.MODE_TFIDF_F32)
for ut, dbg in ((4, 0), (3, 0), (3, 8), (1, 8), (4, 8)):
    ix = rx.DeviceIndex.from_csr(big.indptr, big.indices, big.data, idf, mode="dot", tile_log2=14, unit_tiles=ut)
    ix.set_opts(debug=dbg)
    gd, gs, gc = ix.search(*q, 10)
    ok = np.array_equal(gc, ec) and np.array_equal(gd, ed) and np.array_equal(gs.view(np.uint32), es.view(np.uint32))
    print("unit_tiles", ut, "dbg", dbg, "post16", ix.post16 is not None, "OK" if ok else "MISMATCH")
    if not ok:
        bad = [i for i in range(len(gc)) if gc[i] != ec[i] or not np.array_equal(gd[i], ed[i]) or not np.array_equal(gs[i], es[i])]
        i = bad[0]
        print(" first bad query", i, "of", len(bad), "counts", gc[i], ec[i])
        print("  gpu", gd[i], gs[i])
        print("  exp", ed[i], es[i])
    ix.close()

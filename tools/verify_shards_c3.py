"""Full-scale check of the N = 8 result path on ONE GPU: the C3 corpus as 8 doc-range shards (global idf / avgdl,
corpus-wide score bounds), every shard searched with srx_search_packed, srx_merge_topk_packed_out over the 8 row sets
-- against the single 10 M-doc index, bit for bit.  (What the 8-rank run does, minus RCCL.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_rx
from sparse_rx import synth
from sparse_rx.index import combine_term_bounds, merge_topk_packed_out_device

dev = torch.device("cuda:0")
n_docs, V, nnz, nq, k, seed, W = 10_000_000, 100_000, 100, 10_000, 100, 20253, 8
chunk = synth.CHUNK_DOCS
n_chunks = n_docs // chunk
parts = [synth.uniform_chunk_torch(c, chunk, V, nnz, seed, dev) for c in range(n_chunks)]
df = sum(torch.bincount(p[1], minlength=V) for p in parts)
dl_all = torch.cat([p[3] for p in parts])
avgdl = float(np.mean(dl_all.cpu().numpy()))
idf = torch.as_tensor(np.log((n_docs - df.cpu().numpy() + 0.5) / (df.cpu().numpy() + 0.5)).astype(np.float32), device=dev)
q_ptr, q_term, q_w = synth.queries_np(nq, V, 8, seed=seed + 1)
qd = [torch.as_tensor(x, device=dev) for x in (q_ptr, q_term, q_w)]

def build(cs, base):
    rows = torch.cat([parts[c][0] + i * chunk for i, c in enumerate(cs)])
    cols = torch.cat([parts[c][1] for c in cs]); tf = torch.cat([parts[c][2] for c in cs]); dl = torch.cat([parts[c][3] for c in cs])
    return sparse_rx.DeviceIndex.from_coo(rows, cols, tf, idf, len(cs) * chunk, doc_lengths=dl, avgdl=avgdl, device=dev, doc_base=base)

per = n_chunks // W
shards = [build(range(r * per, (r + 1) * per), r * per * chunk) for r in range(W)]
table = combine_term_bounds([s.fine_bound for s in shards], W)
rows_local = torch.stack([s.search_packed_device(*qd, k).clone() for s in shards])
for s in shards:
    s.set_term_bound(table)
t0 = time.perf_counter()
rows_global = torch.stack([s.search_packed_device(*qd, k).clone() for s in shards])
torch.cuda.synchronize()
n_local, n_global = int(rows_local[:, :, 2 * k].sum()), int(rows_global[:, :, 2 * k].sum())
m_local = merge_topk_packed_out_device(rows_local.contiguous(), k)
m_global = merge_topk_packed_out_device(rows_global.contiguous(), k)
for s in shards:
    s.close()
del shards, rows_local, rows_global
torch.cuda.empty_cache()
whole = build(range(n_chunks), 0)
exp = whole.search_packed_device(*qd, k)
torch.cuda.synchronize()
ok_l, ok_g = bool(torch.equal(m_local, exp)), bool(torch.equal(m_global, exp))
print(f"8 shards x {per * chunk} docs, {nq} queries, k={k}: rows returned by the shards {n_local} (shard bounds) -> {n_global} (corpus-wide bounds)")
print(f"merged == single 10 M-doc index, bit for bit: shard bounds {ok_l}, corpus-wide bounds {ok_g}")
assert ok_l and ok_g

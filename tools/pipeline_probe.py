"""Dev probe: host time of every call HostBatchPipeline.submit makes (C2-sized batch), steady state."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import sparse_rx
from sparse_rx import synth, _capi
L = _capi.lib()
dev = torch.device("cuda:0")
c = synth.uniform_corpus_np(200_000, 20_000, 40, seed=1)
_, idf, avgdl = synth.corpus_stats(c)
ix = sparse_rx.DeviceIndex.from_csr(c.indptr, c.indices, c.data, idf, doc_lengths=c.doc_lengths, avgdl=avgdl, tile_log2=12)
nq, k = 1000, 100
q = synth.queries_np(nq, c.vocab, 8, seed=2)
nt = int(q[0][-1]); row = 2 * k + 1
hq = torch.empty(nq + 1 + 2 * nt, dtype=torch.int32).pin_memory()
hq.numpy()[: nq + 1] = q[0]; hq.numpy()[nq + 1: nq + 1 + nt] = q[1]; hq.numpy()[nq + 1 + nt:].view(np.float32)[:] = q[2]
dq = hq.to(dev)
d_out = [torch.empty((nq, row), dtype=torch.int32, device=dev) for _ in range(3)]
h_out = [torch.empty((nq, row), dtype=torch.int32).pin_memory() for _ in range(3)]
s_copy = torch.cuda.Stream(device=dev)
evd = [torch.cuda.Event() for _ in range(3)]; evo = [torch.cuda.Event() for _ in range(3)]
main = torch.cuda.current_stream(dev)
def run(src, n=30, copy_stream=s_copy, label=""):
    T = np.zeros(6)
    torch.cuda.synchronize()
    for i in range(n):
        j = i % 3
        t0 = time.perf_counter()
        ix.search_packed_device(src[: nq + 1], src[nq + 1: nq + 1 + nt], src[nq + 1 + nt:].view(torch.float32), k, out=d_out[j])
        t1 = time.perf_counter()
        evd[j].record(main); t2 = time.perf_counter()
        copy_stream.wait_event(evd[j]); t3 = time.perf_counter()
        L.srx_memcpy_async(h_out[j].data_ptr(), d_out[j].data_ptr(), 4 * nq * row, copy_stream.cuda_stream); t4 = time.perf_counter()
        evo[j].record(copy_stream); t5 = time.perf_counter()
        if i >= 2: evo[(i - 2) % 3].synchronize()
        t6 = time.perf_counter()
        if i >= 5: T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5]
    torch.cuda.synchronize()
    print(label, "search %.3f  rec %.3f  wait_event %.3f  memcpy %.3f  rec %.3f  sync %.3f ms" % tuple(1e3 * T / (n - 5)))
run(dq, label="device queries, copy stream :")
run(hq, label="zero-copy queries, copy stream:")
run(dq, copy_stream=main, label="device queries, same stream :")
run(hq, copy_stream=main, label="zero-copy queries, same stream:")
ix.set_opts(profile=True)
run(hq, label="zero-copy, copy stream, PROFILE on:")
ix.set_opts(profile=False)
pipe = sparse_rx.HostBatchPipeline(ix, nq, nt, k, depth=3)
for rep in range(2):
    ts = tr = 0.0; tk = []
    for i in range(30):
        a = time.perf_counter(); tk.append(pipe.submit(*q)); b = time.perf_counter(); ts += b - a
        if len(tk) == 3: pipe.result(tk.pop(0)); tr += time.perf_counter() - b
    while tk: pipe.result(tk.pop(0))
    print("HostBatchPipeline: submit %.3f ms, result %.3f ms" % (1e3 * ts / 30, 1e3 * tr / 30))

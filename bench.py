#!/usr/bin/env python3
"""bench.py -- batched BM25 top-k throughput of the HIP engine on synthetic corpora (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (srx_search: scoring kernel + merge kernel; for N > 1 also the RCCL
all-gather of per-shard top-k and the final merge) over one resident query batch.  Default workload "c3" is
the corpus BASELINE.json's targets are quoted on: 10 M docs x 100 k vocab, 100 nnz/doc (10^9 postings, 8 GB of
postings -- fits one MI355X), 10 k queries x 8 terms, k = 100.  With N ranks the SAME corpus and batch are
doc-range sharded N ways (global idf / avgdl), i.e. strong scaling, which is what "queries/s at 1/2/4/8 GPUs,
>= 6x at 8" in the north star measures.  "c2" is BASELINE.json configs[1] (1 M x 50 k, 1 k queries).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- algorithmic posting bytes of one scoring-kernel launch / its average duration, measured with
                  hipEvents recorded on the search stream around the kernel inside the timed region
  cpu_baseline -- the oracle (C/OpenMP restatement of the reference's full-CSR-scan scorer + top-k) timed on this
                  box's host cores on a bounded sample of the same batch; the sample doubles as a parity check.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CPU_THREADS = int(os.environ.get("SRX_CPU_THREADS", "16"))  # the 1-GPU box's CPU share
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: n_docs, vocab, nnz/doc, n_queries, terms/query, k, seed
    "c3": dict(n_docs=10_000_000, vocab=100_000, nnz_per_doc=100, n_queries=10_000, terms=8, k=100, seed=20253),
    "c2": dict(n_docs=1_000_000, vocab=50_000, nnz_per_doc=50, n_queries=1_000, terms=8, k=100, seed=20252),
    # secondary workloads (BASELINE.json configs[3], configs[4]); single-GPU numbers are reported in DESIGN.md only
    "c4": dict(n_docs=5_000_000, vocab=30_000, nnz_per_doc=150, n_queries=1_000, terms=50, k=1000, seed=20254,
               kind="splade"),
    # dev variant of c4 without hot terms (uniform term ids): every tile holds ~4 k postings of ~50 terms
    "c4u": dict(n_docs=5_000_000, vocab=30_000, nnz_per_doc=150, n_queries=1_000, terms=50, k=1000, seed=20256,
                kind="splade", zipf_s=0.0),
    "c5": dict(n_docs=10_000_000, vocab=100_000, nnz_per_doc=100, n_queries=256, terms=8, k=100, seed=20255, kind="zipf"),
}
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--docs", type=int, default=0, help="override n_docs (scaled-down checks; marks the workload custom)")
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--tile-log2", type=int, default=14)
    ap.add_argument("--supertile-log2", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--unit-tiles", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug", type=int, default=0, help="kernel ablation flags (timing experiments only; results are wrong)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget")
    ap.add_argument("--exchange", default="a2a", choices=["a2a", "allgather"], help="N > 1: how per-shard top-k lists meet")
    ap.add_argument("--local-bounds", action="store_true", help="N > 1: keep per-shard score bounds (no corpus-wide bound exchange)")
    ap.add_argument("--emulate-world", type=int, default=0, help="dev: on one GPU, use the score bounds a shard would get among this many identical shards")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: do not overlap the exchange of a batch with the scoring of the next one")
    ap.add_argument("--chunks", type=int, default=0, help="N > 1: sub-batches whose exchange overlaps the next one's scoring (0 = auto)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 code path (RCCL all-gather + packed merge) with world size 1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import sparse_rx
    from sparse_rx import synth
    sparse_rx._capi.lib()  # the HIP engine is mandatory

    w = dict(WORKLOADS[args.workload])
    custom = False
    for key, val in (("n_docs", args.docs), ("n_queries", args.queries), ("k", args.k)):
        if val:
            w[key] = val
            custom = True
    n_docs, V, k, nq = w["n_docs"], w["vocab"], w["k"], w["n_queries"]
    chunk_docs = min(synth.CHUNK_DOCS, n_docs)
    n_chunks = (n_docs + chunk_docs - 1) // chunk_docs
    assert n_docs % chunk_docs == 0 and n_chunks % world == 0, "corpus must split into equal chunks per rank"
    my_chunks = range(rank * n_chunks // world, (rank + 1) * n_chunks // world)
    shard_docs = len(my_chunks) * chunk_docs
    doc_base = my_chunks[0] * chunk_docs

    # ---- query batch: generated on the host FIRST, so that nothing but kernel launches separates the index build
    #      (GPU busy, clocks up) from the warm-up and timed steps ----
    kind = w.get("kind", "uniform")
    if kind == "uniform":
        q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1)
    elif kind == "zipf":
        q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1, dist="zipf", s=1.0)
    else:
        q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1, dist="zipf", s=w.get("zipf_s", 0.7), weights="learned")
    # ---- corpus shard on the device (doc-major COO in CSR order), global statistics ---------------------------
    t_build = time.perf_counter()
    rows_l, cols_l, tf_l, dl_l = [], [], [], []
    for ci, c in enumerate(my_chunks):
        kind = w.get("kind", "uniform")
        gen = {"uniform": synth.uniform_chunk_torch, "zipf": synth.zipf_chunk_torch, "splade": synth.splade_chunk_torch}[kind]
        r, cc, tf, dl = (gen(c, chunk_docs, V, w["nnz_per_doc"], w["seed"], dev, s=w["zipf_s"]) if "zipf_s" in w
                         else gen(c, chunk_docs, V, w["nnz_per_doc"], w["seed"], dev))
        rows_l.append(r + ci * chunk_docs)
        cols_l.append(cc)
        tf_l.append(tf)
        dl_l.append(dl)
    rows, cols, tf, dl = torch.cat(rows_l), torch.cat(cols_l), torch.cat(tf_l), torch.cat(dl_l)
    del rows_l, cols_l, tf_l, dl_l
    df = torch.bincount(cols, minlength=V)
    df_local = df.clone()
    if dist is not None:
        dist.all_reduce(df)
        dl_all = torch.empty(n_docs, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(dl_all, dl)
    else:
        dl_all = dl
    kind = w.get("kind", "uniform")
    avgdl = float(np.mean(dl_all.cpu().numpy()))  # retrieval.py:190 over the WHOLE corpus
    if kind == "splade":  # learned-sparse dot product: no idf, no length normalisation (simd_tfidf_score with idf == 1)
        idf_np = np.ones(V, dtype=np.float32)
        avgdl = 1.0
    else:
        idf_np = np.log((n_docs - df.cpu().numpy() + 0.5) / (df.cpu().numpy() + 0.5)).astype(np.float32)  # retrieval.py:189
    idf = torch.as_tensor(idf_np, device=dev)
    nnz_local = int(cols.numel())

    host_csr = None
    want_cpu = (not args.no_cpu_baseline) and world == 1  # also in --force-dist rehearsals: verifies the exchange path
    if want_cpu:
        indptr = torch.zeros(shard_docs + 1, dtype=torch.int64, device=dev)
        indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=shard_docs), 0)
        host_csr = (indptr.cpu().numpy(), cols.cpu().numpy(), tf.cpu().numpy(), dl.cpu().numpy())
        del indptr

    ix = sparse_rx.DeviceIndex.from_coo(rows, cols, tf, idf, shard_docs, doc_lengths=dl, k1=1.2, b=0.75, avgdl=avgdl,
                                        device=dev, doc_base=doc_base, tile_log2=args.tile_log2,
                                        mode="dot" if kind == "splade" else "bm25",
                                        val_dtype="f16" if kind == "splade" else "f32")
    del rows, cols, tf  # (no empty_cache(): 288 GB of HBM, and freeing would idle the GPU before the timed region)
    ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=True, debug=args.debug,
                unit_tiles=args.unit_tiles)
    if world > 1 and not args.local_bounds:
        # corpus-wide score bounds (one all-gather of a few MB at start-up): every shard starts from the single-GPU
        # threshold instead of its own shard's; exact
        sparse_rx.global_term_bounds(ix)
    elif args.emulate_world > 1 and ix.fine_bound is not None:
        # dev rehearsal on one GPU: the bounds this shard would get among `emulate_world` statistically identical shards
        # (its rows are then NOT its own full top-k: use with --no-cpu-baseline)
        from sparse_rx.index import combine_term_bounds
        ix.set_term_bound(combine_term_bounds(ix.fine_bound.unsqueeze(0).expand(args.emulate_world, -1, -1), args.emulate_world))
    build_s = time.perf_counter() - t_build

    # ---- query batch (generated on the host before the corpus was built), resident in HBM before the timed region ----
    qp, qt, qw = (torch.as_tensor(x, device=dev) for x in (q_ptr, q_term, q_w))
    out = (torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
           torch.empty((nq,), dtype=torch.int32, device=dev))
    ix_search = ix.search_device
    if dist is None:
        ix.search_device = lambda a, b, c, kk: ix_search(a, b, c, kk, out=out)  # reuse the output tensors every step

    searcher = sparse_rx.ShardedSearcher.for_device_index(ix)  # N > 1: + RCCL all-gather of the packed per-shard top-k + merge
    searcher.force_exchange = args.force_dist
    searcher.mode = args.exchange
    searcher.overlap = not args.no_overlap

    def step():
        return searcher.search(qp, qt, qw, k, chunks=args.chunks, q_ptr_host=q_ptr)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize(dev)
    ix.profile_read()  # drop warm-up samples
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    t_submit = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"[bench] host submit time {1e3 * t_submit / args.steps:.3f} ms/step of {1e3 * elapsed / args.steps:.3f} ms/step")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ix.profile_read()

    pcie_qps = None
    if dist is None:
        t = time.perf_counter()
        for _ in range(3):
            ix.search(q_ptr, q_term, q_w, k)  # H2D of the query batch + search + D2H of nq*k results
        pcie_qps = 3 * nq / (time.perf_counter() - t)
        log(f"[bench] PCIe-inclusive (host query batch in, host results out): {pcie_qps:,.0f} queries/s")
        ix.profile_read()

    # ---- roofline of the dominant kernel (this rank's scoring kernel) --------------------------------------------
    post_bytes = 8 if ix.post_val.dtype == torch.float32 else 6
    alg_bytes = int(df_local[qt.long()].sum().item()) * post_bytes + nq * k * 8  # sum df_t*(4+4) + k*8
    # Dominant kernel = the tier-1 wave kernel; the tier-2 block kernel only sees flagged units (none on the uniform
    # corpora), but its time is kept in the denominator so that no posting byte is counted without its time.
    score_s = (prof["wave_ms"] + prof["block_ms"]) * 1e-3
    achieved = alg_bytes / score_s / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    wl_name = args.workload + ("-custom" if custom else "")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(f"{wl_name}@{world}", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "queries/sec + achieved HBM GB/s, batch BM25 top-k=%d" % k,
        "value": nq * args.steps / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f16" if kind == "splade" else "f32",
        "data": "synthetic",
        "config": {"workload": f"{wl_name}: {n_docs} docs x {V} vocab, {w['nnz_per_doc']} nnz/doc, "
                               f"{nq}-query batch x {w['terms']} terms, k={k}",
                   "n_docs": n_docs, "vocab": V, "nnz": nnz_local * world if world > 1 else nnz_local, "n_queries": nq, "k": k,
                   "sharding": f"doc-range x{world}" + ((" + RCCL all-to-all of packed per-shard top-k, merge of the own query block, all-gather of merged rows" if args.exchange == "a2a" else " + one RCCL all-gather of packed per-shard top-k") if (world > 1 or args.force_dist) else ""),
                   "index_build_s": round(build_s, 2), "pcie_inclusive_qps": pcie_qps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": "srx_wave_kernel<float>" if kind != "splade" else "srx_score_kernel<__half> (tier 2: k > 128)",
                     "kernel_ms": prof["wave_ms"], "tier2_kernel_ms": prof["block_ms"], "merge_kernel_ms": prof["merge_ms"],
                     "launches_timed": prof["calls"],
                     "algorithmic_bytes_per_launch": alg_bytes},
    }

    # ---- CPU baseline (rank 0, N = 1): the oracle on a bounded sample; doubles as a parity check ------------------
    if want_cpu and rank == 0:
        import oracle
        indptr_h, cols_h, tf_h, dl_h = host_csr
        gd, gs, gc = (x.cpu().numpy() for x in res)
        # size the sample: one query first
        def run(nsamp):
            qs = (q_ptr[: nsamp + 1] - q_ptr[0], q_term[: q_ptr[nsamp]], q_w[: q_ptr[nsamp]])
            t = time.perf_counter()
            r = oracle.search_batch(indptr_h, cols_h, tf_h, dl_h, idf_np, qs[0], qs[1], qs[2], k, 1.2, 0.75, avgdl, native=True,
                                    mode=oracle.MODE_TFIDF_F32 if kind == "splade" else oracle.MODE_BM25_F32)
            return time.perf_counter() - t, r
        t1, _ = run(1)   # also warms the page cache / thread pool
        t1, _ = run(1)
        nsamp = int(max(2, min(nq, 512, args.cpu_seconds / max(t1, 1e-4))))
        tc, (ed, es, ec) = run(nsamp)
        ok = (np.array_equal(gc[:nsamp], ec) and np.array_equal(gd[:nsamp], ed)
              and np.array_equal(gs[:nsamp].view(np.uint32), es.view(np.uint32)))
        if not ok:
            raise SystemExit("PARITY FAILURE: GPU results differ from the CPU oracle on the baseline sample")
        result["cpu_baseline"] = {"value": nsamp / tc, "unit": "queries/s", "cores": oracle.num_threads(native=True),
                                  "kind": "port",
                                  "sample": f"first {nsamp} queries of the same batch, full-CSR-scan scorer + top-k "
                                            f"(oracle/bm25_oracle.c, -O3 -march=native, OpenMP), {tc:.1f} s; "
                                            "GPU output for the sample verified bit-exact against it",
                                  "cpu": _cpu_model()}
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    ix.close()
    if dist is not None:
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()

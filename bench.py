#!/usr/bin/env python3
"""bench.py -- batched BM25 top-k throughput of the HIP engine on synthetic corpora (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2|c1|c4|c5] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (srx_search: scoring kernels + merge kernel; for N > 1 also the RCCL exchange of
the per-shard top-k and the final merge) over one query batch resident in HBM.  Default workload "c3" is the corpus
BASELINE.json's targets are quoted on: 10 M docs x 100 k vocab, 100 nnz/doc (10^9 postings, 8 GB of postings -- fits
one MI355X), 10 k queries x 8 terms, k = 100.  With N ranks the SAME corpus and batch are doc-range sharded N ways
(global idf / avgdl), i.e. strong scaling, which is what "queries/s at 1/2/4/8 GPUs, >= 6x at 8" in the north star
measures.  "c2" is BASELINE.json configs[1] (1 M x 50 k, 1 k queries), "c1" configs[0] (FiQA-shaped text through the
full build_bm25_index path, k = 10), "c4" / "c5" configs[3] / configs[4].

Rank 0 prints ONE JSON line (contract in the task statement).  Fields beyond the contract:
  value / ms_per_step          device-resident: the query batch is in HBM when the timed region starts, results stay there
  config.pcie_inclusive_qps    the same steps fed from HOST batches (pinned, multi-buffered H2D of the query CSR and
                               D2H of the nq x k result rows on copy streams, overlapped with the search of the next
                               batch) -- SURVEY.md 8(d)'s "batch wall time incl. H2D and D2H"
  roofline.achieved / frac     algorithmic bytes of one step / the scoring kernels' time per step (hipEvents recorded on
                               the search stream around the kernels inside the timed region)  -- the kernel's roofline.
                               Bytes per posting = what the dominant kernel streams (roofline.bytes_per_posting: 6 for the
                               tier-1 kernel on fp32 values = 16-bit local doc id + value; SURVEY.md 8(d) "smaller actual
                               per-posting size"); roofline.frac_canonical = the same time against 4-byte doc ids
  roofline.batch_achieved / batch_frac       the same bytes / the step's wall time / (8 TB/s x n_gpus)  
  roofline.pcie_inclusive_frac               the same bytes / the PCIe-inclusive step time
  roofline.traffic             HBM bytes per launch from the rocprofv3 PMC pass recorded in profiles/traffic.json -- only
                               when that pass profiled THIS kernel source (sha256 of csrc/*.hip + srx_common.h), else null
  cpu_baseline                 the oracle (C/OpenMP restatement of the reference's full-CSR-scan scorer + top-k) timed on
                               this box's host cores on a bounded sample of the same batch
Every run checks its GPU results against the oracle on a sample of the timed batch and exits non-zero on a mismatch:
at N = 1 the sample of the cpu_baseline leg; at N > 1 every rank scores a 32-query sample on its own shard's host CSR,
rank 0 merges the per-shard lists on the host by (score desc, doc asc) and compares them with the exchanged rows.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_share() -> int:
    """Cores this process may actually use: the scheduler affinity, cut down to the cgroup CPU quota when there is one
    (a GPU box hands a 1-GPU job a share of the host's cores; more threads than that only oversubscribe it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


# CPU-baseline threads: every core this process may run on (OMP_NUM_THREADS = nproc); with N ranks on one
# node the cores are divided between them.  SRX_CPU_THREADS overrides.
_WORLD = int(os.environ.get("WORLD_SIZE", "1"))
CPU_THREADS = int(os.environ.get("SRX_CPU_THREADS", "0")) or max(1, _cpu_share() // max(1, _WORLD))
_OMP_PRESET = "OMP_NUM_THREADS" in os.environ
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: n_docs, vocab, nnz/doc, n_queries, terms/query, k, seed
    "c3": dict(n_docs=10_000_000, vocab=100_000, nnz_per_doc=100, n_queries=10_000, terms=8, k=100, seed=20253),
    "c2": dict(n_docs=1_000_000, vocab=50_000, nnz_per_doc=50, n_queries=1_000, terms=8, k=100, seed=20252),
    # BASELINE.json configs[0]: FiQA-shaped synthetic TEXT through the full text path (sizes are the generator's)
    "c1": dict(n_docs=57_638, vocab=80_000, nnz_per_doc=130, n_queries=100, terms=10, k=10, seed=20251, kind="text"),
    # secondary workloads (BASELINE.json configs[3], configs[4]); single-GPU numbers are reported in DESIGN.md only
    "c4": dict(n_docs=5_000_000, vocab=30_000, nnz_per_doc=150, n_queries=1_000, terms=50, k=1000, seed=20254,
               kind="splade", tile_log2=12, unit_tiles=1),  # 4096-doc tiles, runs padded per tile: tier 2's wave-level
                                                            # dense path in its unmasked form (four tiles per workgroup)
    # dev variant of c4 without hot terms (uniform term ids): every tile holds ~4 k postings of ~50 terms
    "c4u": dict(n_docs=5_000_000, vocab=30_000, nnz_per_doc=150, n_queries=1_000, terms=50, k=1000, seed=20256,
                kind="splade", zipf_s=0.0),
    "c5": dict(n_docs=10_000_000, vocab=100_000, nnz_per_doc=100, n_queries=256, terms=8, k=100, seed=20255, kind="zipf"),
}
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
N_CHECK = 32            # N > 1: queries of the timed batch verified against the per-shard oracle


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha() -> str:
    import sparse_rx
    return sparse_rx._capi.kernel_sources_sha256()


def merge_shard_lists(parts, k):
    """Host merge of per-shard oracle results [(doc i32[n,k] GLOBAL ids, score f32[n,k], count i32[n])] ->
    (doc, score, count) ranked (score desc, doc asc), padded with -1 / 0: the contract of srx_merge_topk."""
    n = parts[0][0].shape[0]
    out_d = np.full((n, k), -1, np.int32)
    out_s = np.zeros((n, k), np.float32)
    out_c = np.zeros(n, np.int32)
    for q in range(n):
        d = np.concatenate([p[0][q, : p[2][q]] for p in parts])
        s = np.concatenate([p[1][q, : p[2][q]] for p in parts])
        order = np.lexsort((d, -s.astype(np.float64)))[:k]
        out_d[q, : len(order)] = d[order]
        out_s[q, : len(order)] = s[order]
        out_c[q] = len(order)
    return out_d, out_s, out_c


def sharded_sample_check(dist_mod, rank, world, host_csr, doc_base, idf_np, avgdl, q_host, k, mode, gpu_rows, n_check=N_CHECK):
    """N > 1 self-check.  Every rank runs the oracle on ITS shard's host CSR (corpus-wide idf / avgdl) for the first
    n_check queries of the batch; the per-shard lists are gathered on rank 0, merged on the host and compared with the
    rows the GPU path returned, bit for bit.  Returns (ok, n_checked) on rank 0, (True, n) elsewhere."""
    import oracle
    q_ptr, q_term, q_w = q_host
    n = int(min(n_check, len(q_ptr) - 1))
    qs = (q_ptr[: n + 1] - q_ptr[0], q_term[: q_ptr[n]], q_w[: q_ptr[n]])
    indptr_h, cols_h, tf_h, dl_h = host_csr
    d, s, c = oracle.search_batch(indptr_h, cols_h, tf_h, dl_h, idf_np, qs[0], qs[1], qs[2], k, 1.2, 0.75, avgdl, native=True, mode=mode)
    d = np.where(d >= 0, d + doc_base, -1).astype(np.int32)
    gathered = [None] * world if rank == 0 else None
    dist_mod.gather_object((d, s, c), gathered, dst=0)
    if rank != 0:
        return True, n
    ed, es, ec = merge_shard_lists(gathered, k)
    gd, gs, gc = gpu_rows
    ok = (np.array_equal(gc[:n], ec) and np.array_equal(gd[:n], ed) and np.array_equal(gs[:n].view(np.uint32), es.view(np.uint32)))
    return bool(ok), n


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...  bench.py <same argv>`)
    and return their exit code.  This parent never touches the GPU (no HIP call, no torch.cuda.is_available(): only the
    device count, which does not initialise the runtime) and never re-execs; rank 0's JSON line goes to the inherited
    stdout."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n_gpus:
        log(f"[bench] --gpus {n_gpus} but only {have} device(s) visible")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    if not _OMP_PRESET:
        env.pop("OMP_NUM_THREADS", None)  # each rank sizes its own CPU share from WORLD_SIZE
    log("[bench] self-launch: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def emulated_bound_check(ix, host_csr, idf_np, avgdl, q_host, k, mode, gpu_rows, n_check=N_CHECK):
    """--emulate-world rehearsals: the shard searches with the score bounds it would have among W identical shards, so a
    query's rows stop where its initial threshold tau0 = max_t (bound[t] * idf[t]) * qw[t] cuts them (the rest of the top-k
    would come from the other shards).  Checked against the oracle's full ranked list of the shard, for the first
    n_check queries: the GPU rows must be a prefix of it (same docs, same score bits, same order) and must hold every oracle
    row whose score is >= tau0.  Returns (ok, n_checked, message)."""
    import oracle
    q_ptr, q_term, q_w = q_host
    n = int(min(n_check, len(q_ptr) - 1))
    qs = (q_ptr[: n + 1] - q_ptr[0], q_term[: q_ptr[n]], q_w[: q_ptr[n]])
    ed, es, ec = oracle.search_batch(host_csr[0], host_csr[1], host_csr[2], host_csr[3], idf_np, qs[0], qs[1], qs[2], k, 1.2, 0.75, avgdl,
                                     native=True, mode=mode)
    ed = np.where(ed >= 0, ed + ix.doc_base, -1).astype(np.int32)
    gd, gs, gc = gpu_rows
    tb = ix.term_bound.cpu().numpy() if ix.term_bound is not None else None
    col = 0 if k <= 1 else 1 if k <= 10 else 2 if k <= 100 else 3 if k <= 1000 else -1
    for q in range(n):
        t, w = qs[1][qs[0][q]: qs[0][q + 1]], qs[2][qs[0][q]: qs[0][q + 1]]
        tau0 = np.float32(0.0)
        if tb is not None and col >= 0 and len(t) and np.all(idf_np[t] >= 0) and np.all(w >= 0):
            ok_t = (idf_np[t] > 0) & (w > 0)
            if ok_t.any():
                tau0 = np.max((tb[t[ok_t], col].astype(np.float32) * idf_np[t[ok_t]]) * w[ok_t].astype(np.float32))  # the kernels' fp32 expression
        c = int(gc[q])
        if c > int(ec[q]) or not (np.array_equal(gd[q, :c], ed[q, :c]) and np.array_equal(gs[q, :c].view(np.uint32), es[q, :c].view(np.uint32))):
            return False, n, f"query {q}: the GPU rows are not a prefix of the oracle's ranked list"
        need = int(np.sum(es[q, : ec[q]] >= tau0))
        if c < need:
            return False, n, f"query {q}: {need} oracle rows score >= tau0 = {tau0}, the GPU returned {c}"
    return True, n, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--docs", type=int, default=0, help="override n_docs (scaled-down checks; marks the workload custom)")
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--tile-log2", type=int, default=0, help="docs per skip-table tile = 2^n (0 = the workload's default, 14 unless it says otherwise)")
    ap.add_argument("--supertile-log2", type=int, default=0)
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--unit-tiles", type=int, default=0)
    ap.add_argument("--build-unit-tiles", type=int, default=-1,
                    help="tiles per padded run of the posting layout (index build); -1 = the workload's, 0 = automatic")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline AND every oracle check (profiling runs)")
    ap.add_argument("--debug", type=int, default=0, help="kernel ablation flags (timing experiments only; results are wrong)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget")
    ap.add_argument("--exchange", default="a2a", choices=["a2a", "allgather"], help="N > 1: how per-shard top-k lists meet")
    ap.add_argument("--local-bounds", action="store_true", help="N > 1: keep per-shard score bounds (no corpus-wide bound exchange)")
    ap.add_argument("--emulate-world", type=int, default=0, help="dev: on one GPU, use the score bounds a shard would get among this many identical shards")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: do not overlap the exchange of a batch with the scoring of the next one")
    ap.add_argument("--keep-canonical", action="store_true",
                    help="keep the canonical blocks (32-bit doc ids) resident next to the compact copy (default: one copy -- the compact one, read by both tiers)")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: submit every step eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--chunks", type=int, default=0, help="N > 1: sub-batches whose exchange overlaps the next one's scoring (0 = auto)")
    ap.add_argument("--same-query", action="store_true", help="dev: every query of the batch is query 0 (postings stay in cache: the compute-bound time of the kernels)")
    ap.add_argument("--pipe-copy", action="store_true", help="host-batch pipeline: explicit H2D copy of the query block instead of zero-copy reads")
    ap.add_argument("--pipe-depth", type=int, default=3, help="host-batch pipeline slots (PCIe-inclusive leg)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 code path (RCCL exchange + packed merge + sharded self-check) with world size 1")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="bracket the kernels of every N-th search with hipEvents (roofline leg); 0 = never (dev: roofline fields are then meaningless)")
    ap.add_argument("--streams", type=int, default=0,
                    help="batches kept in flight on separate HIP streams in the steady-state leg (0 = 4 for batches of <= 2048 queries, none otherwise)")
    ap.add_argument("--self-launch", action="store_true",
                    help="dev: take the launcher path (child ranks under torch.distributed.run) even at --gpus 1")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_launch):
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    # Everything below is submitted from a stream of our own, not from the legacy default stream: the default stream
    # synchronises implicitly with blocking streams (RCCL's among them), which serialised the scoring of a batch behind the
    # exchange of the one before it (profiles/r03_exchange_timeline_default_stream.txt)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    import sparse_rx
    from sparse_rx import synth
    sparse_rx._capi.lib()  # the HIP engine is mandatory

    w = dict(WORKLOADS[args.workload])
    custom = False
    for key, val in (("n_docs", args.docs), ("n_queries", args.queries), ("k", args.k)):
        if val:
            w[key] = val
            custom = True
    kind = w.get("kind", "uniform")
    if args.tile_log2 == 0:
        args.tile_log2 = w.get("tile_log2", 14)
    if args.build_unit_tiles < 0:
        args.build_unit_tiles = w.get("unit_tiles", 0)
    n_docs, V, k, nq = w["n_docs"], w["vocab"], w["k"], w["n_queries"]
    want_check = not args.no_cpu_baseline          # oracle checks (and, at N = 1, the timed CPU baseline)
    # --emulate-world: the emulated corpus-wide bounds cut this shard's rows below its own top-k (the other shards' docs
    # would fill them), so the rows are checked as what they must be -- a PREFIX of the oracle's ranked list for the shard
    # that holds every row at or above the query's initial threshold (emulated_bound_check) -- and no CPU baseline is timed
    emu_check = args.emulate_world > 1 and want_check
    want_cpu = want_check and world == 1 and not args.force_dist and not emu_check

    t_build = time.perf_counter()
    host_csr = None
    if kind == "text":
        # ---- C1: FiQA-shaped text through the reference's host path (tokenise, vocabulary, CSR, idf, avgdl); every
        #      rank builds the same host index and uploads its own doc range with the corpus-wide statistics ----
        corpus, queries = synth.fiqa_shaped_text(n_docs=n_docs, vocab=V, mean_doc_len=w["nnz_per_doc"], n_queries=nq,
                                                 mean_query_len=w["terms"], seed=w["seed"])
        hi = sparse_rx.build_host_index(corpus)
        del corpus
        V = hi.vocab_size
        q_ptr, q_term, q_w = sparse_rx.encode_queries(list(queries.values()), hi.vocabulary)
        a, b = sparse_rx.shard_range(n_docs, world, rank)
        lo, hi_ = int(hi.indptr[a]), int(hi.indptr[b])
        sub = ((hi.indptr[a: b + 1] - lo).astype(np.int64), hi.indices[lo:hi_], hi.data[lo:hi_], hi.doc_lengths[a:b])
        shard_docs, doc_base, idf_np, avgdl = b - a, a, hi.idf, hi.avgdl
        df_local = torch.as_tensor(np.bincount(sub[1], minlength=V), device=dev)
        nnz_local, nnz_total = len(sub[1]), hi.nnz
        if want_check:
            host_csr = sub
        ix = sparse_rx.DeviceIndex.from_csr(sub[0], sub[1], sub[2], idf_np, doc_lengths=sub[3], k1=1.2, b=0.75, avgdl=avgdl,
                                            device=dev, doc_base=doc_base, tile_log2=args.tile_log2,
                                            keep_canonical=args.keep_canonical or bool(args.supertile_log2 or args.unit_tiles))
    else:
        chunk_docs = min(synth.CHUNK_DOCS, n_docs)
        n_chunks = (n_docs + chunk_docs - 1) // chunk_docs
        assert n_docs % chunk_docs == 0 and n_chunks % world == 0, "corpus must split into equal chunks per rank"
        my_chunks = range(rank * n_chunks // world, (rank + 1) * n_chunks // world)
        shard_docs = len(my_chunks) * chunk_docs
        doc_base = my_chunks[0] * chunk_docs
        # ---- query batch: generated on the host FIRST, so that nothing but kernel launches separates the index build
        #      (GPU busy, clocks up) from the warm-up and timed steps ----
        if kind == "uniform":
            q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1)
        elif kind == "zipf":
            q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1, dist="zipf", s=1.0)
        else:
            q_ptr, q_term, q_w = synth.queries_np(nq, V, w["terms"], seed=w["seed"] + 1, dist="zipf", s=w.get("zipf_s", 0.7), weights="learned")
        if args.same_query:
            n0 = int(q_ptr[1])
            q_term, q_w = np.tile(q_term[:n0], nq), np.tile(q_w[:n0], nq)
            q_ptr = (np.arange(nq + 1) * n0).astype(np.int32)
        # ---- corpus shard on the device (doc-major COO in CSR order), global statistics ---------------------------
        rows_l, cols_l, tf_l, dl_l = [], [], [], []
        gen = {"uniform": synth.uniform_chunk_torch, "zipf": synth.zipf_chunk_torch, "splade": synth.splade_chunk_torch}[kind]
        for ci, c in enumerate(my_chunks):
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
        avgdl = float(np.mean(dl_all.cpu().numpy()))  # retrieval.py:190 over the WHOLE corpus
        if kind == "splade":  # learned-sparse dot product: no idf, no length normalisation (simd_tfidf_score with idf == 1)
            idf_np = np.ones(V, dtype=np.float32)
            avgdl = 1.0
        else:
            idf_np = np.log((n_docs - df.cpu().numpy() + 0.5) / (df.cpu().numpy() + 0.5)).astype(np.float32)  # retrieval.py:189
        idf = torch.as_tensor(idf_np, device=dev)
        nnz_local = int(cols.numel())
        nnz_total = int(df.sum().item())
        if want_check:  # this shard's doc-major CSR on the host: the oracle's input
            indptr = torch.zeros(shard_docs + 1, dtype=torch.int64, device=dev)
            indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=shard_docs), 0)
            host_csr = (indptr.cpu().numpy(), cols.cpu().numpy(), tf.cpu().numpy(), dl.cpu().numpy())
            del indptr
        ix = sparse_rx.DeviceIndex.from_coo(rows, cols, tf, idf, shard_docs, doc_lengths=dl, k1=1.2, b=0.75, avgdl=avgdl,
                                            device=dev, doc_base=doc_base, tile_log2=args.tile_log2,
                                            unit_tiles=args.build_unit_tiles, mode="dot" if kind == "splade" else "bm25",
                                            val_dtype="f16" if kind == "splade" else "f32",
                                            keep_canonical=args.keep_canonical or bool(args.supertile_log2 or args.unit_tiles))
        del rows, cols, tf  # (no empty_cache(): 288 GB of HBM, and freeing would idle the GPU before the timed region)
    ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=args.profile_every, debug=args.debug,
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

    # ---- query batch resident in HBM before the timed region ----
    qp, qt, qw = (torch.as_tensor(x, device=dev) for x in (q_ptr, q_term, q_w))
    out = (torch.empty((nq, k), dtype=torch.int32, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
           torch.empty((nq,), dtype=torch.int32, device=dev))
    calls = {"n": 0}  # local searches (kernel-launch groups) issued: with --chunks a step is several of them
    ix_search, ix_search_packed = ix.search_device, ix.search_packed_device

    def counted_search(a, b, c, kk, out_=None):
        calls["n"] += 1
        return ix_search(a, b, c, kk, out=out if (dist is None and out_ is None) else out_)  # N = 1: reuse the output tensors

    def counted_search_packed(a, b, c, kk, out_=None, **kw):
        calls["n"] += 1
        return ix_search_packed(a, b, c, kk, out=out_, **kw)

    ix.search_device, ix.search_packed_device = counted_search, counted_search_packed
    searcher = sparse_rx.ShardedSearcher.for_device_index(ix)  # N > 1: + RCCL exchange of the packed per-shard top-k + merge
    searcher.force_exchange = args.force_dist
    searcher.mode = args.exchange
    searcher.overlap = not args.no_overlap
    # N > 1: the step (search + exchange + merge) is captured into HIP graphs on two alternating lanes (ShardedSearcher._search_graph)
    searcher.graph = dist is not None and not args.no_graph and not args.no_overlap
    want_graph = bool(searcher.graph)
    if want_graph:  # no event records inside the captured steps: the kernel timings come from eager steps after the timed region
        ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=0, debug=args.debug, unit_tiles=args.unit_tiles)

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
    calls["n"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    t_submit = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"[bench] rank {rank}: host submit time {1e3 * t_submit / args.steps:.3f} ms/step of {1e3 * elapsed / args.steps:.3f} ms/step")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ix.profile_read()
    calls_per_step = calls["n"] / max(1, args.steps)
    kernel_timing = "hipEvents on the search stream around the kernels of every %d-th timed step" % max(1, args.profile_every)
    if want_graph:
        # the timed steps were graph replays without event records: time the same kernels on a few eager steps now
        searcher.wait()
        torch.cuda.synchronize(dev)
        graphed = bool(searcher.graph)  # False if the capture was refused and the eager path ran
        searcher.graph = False
        ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=1, debug=args.debug, unit_tiles=args.unit_tiles)
        calls["n"] = 0
        for _ in range(5):
            step()
        searcher.wait()
        torch.cuda.synchronize(dev)
        prof = ix.profile_read()
        calls_per_step = calls["n"] / 5
        kernel_timing = ("5 eager steps after the timed region, hipEvents around every kernel (the timed steps were HIP-graph replays)" if graphed
                         else "5 eager steps after the timed region (graph capture was refused: the timed steps were eager too)")
        result_graph = graphed
    else:
        result_graph = False
    ix.search_device, ix.search_packed_device = ix_search, ix_search_packed

    # ---- PCIe-inclusive: the same steps fed from host batches through the pinned double-buffered pipeline ----------
    pcie_qps = pcie_ms = None
    if dist is None:
        pipe = sparse_rx.HostBatchPipeline(ix, nq, len(q_term), k, depth=args.pipe_depth, zero_copy_queries=not args.pipe_copy,
                                           zero_copy_results=not args.pipe_copy)
        n_p = max(20, min(2 * args.steps, 60))
        tickets = []
        for i in range(12):  # warm-up: the pinned staging / result buffers are first touched here (slow the first few times)
            pipe.result(pipe.submit(q_ptr, q_term, q_w))
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        t_sub = t_res = 0.0
        for i in range(n_p):
            ta = time.perf_counter()
            tickets.append(pipe.submit(q_ptr, q_term, q_w))
            tb = time.perf_counter()
            t_sub += tb - ta
            if len(tickets) == args.pipe_depth:
                pd_, ps_, pc_ = pipe.result(tickets.pop(0))  # host arrays (views of the pinned result slot)
                t_res += time.perf_counter() - tb
        while tickets:
            pd_, ps_, pc_ = pipe.result(tickets.pop(0))
        pcie_s = (time.perf_counter() - t) / n_p
        pcie_qps, pcie_ms = nq / pcie_s, pcie_s * 1e3
        rd, rs, rc_ = (x.cpu().numpy() for x in res)
        if not args.debug and not (np.array_equal(pd_, rd) and np.array_equal(ps_.view(np.uint32), rs.view(np.uint32)) and np.array_equal(pc_, rc_)):
            raise SystemExit("PARITY FAILURE: the host-batch pipeline returned rows that differ from the device-resident search")
        log(f"[bench] host pipeline: submit {1e3 * t_sub / n_p:.3f} ms/batch, result wait {1e3 * t_res / n_p:.3f} ms/batch")
        ht = [1e3 * x / (n_p + 12) for x in pipe.host_times]
        log(f"[bench] host pipeline: inside submit (incl. warm-up batches): validate {ht[0]:.3f}, staging {ht[1]:.3f}, search call {ht[2]:.3f}, events + D2H call {ht[3]:.3f} ms/batch")
        log(f"[bench] PCIe-inclusive (host query batch in, host results out, pinned, {args.pipe_depth} slots): {pcie_qps:,.0f} queries/s "
            f"({pcie_ms:.3f} ms/step vs {1e3 * elapsed / args.steps:.3f} device-resident)")
        pp = ix.profile_read()
        log(f"[bench] host pipeline: kernels in that loop: wave {pp['wave_ms']:.3f} ms, block {pp['block_ms']:.3f} ms, merge {pp['merge_ms']:.3f} ms per batch")
        pipe.close()

    # ---- steady state with several batches in flight (small batches are launch / latency bound one at a
    #      time -- C2 moves 48 MB per batch, 6 us at peak -- and the reference's pipeline searches batches of <= 100 queries,
    #      evaluate_rag_pipeline.py:741,779): `streams` slots, each with its own HIP stream, workspace and result rows,
    #      (a) device-resident batches, (b) host batches through the multi-stream HostBatchPipeline ----
    steady = None
    n_streams = args.streams if args.streams > 0 else (4 if (nq <= 2048 and elapsed / args.steps < 0.5e-3) else 0)  # small, short batches only
    if dist is None and n_streams > 1:
        ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=False, debug=args.debug, unit_tiles=args.unit_tiles)
        n_b = max(200, 20 * args.steps)
        slots = [(torch.cuda.Stream(device=dev), torch.empty(max(ix.workspace_bytes(nq, k), 1 << 16), dtype=torch.uint8, device=dev),
                  torch.empty((nq, 2 * k + 1), dtype=torch.int32, device=dev)) for _ in range(n_streams)]
        for rep_ in range(2):  # first pass = warm-up
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for i in range(n_b):
                st_, ws_, out_ = slots[i % n_streams]
                ix.search_packed_device(qp, qt, qw, k, out=out_, stream=st_, workspace=ws_)
            torch.cuda.synchronize(dev)
            dev_s = (time.perf_counter() - t) / n_b
        rows = slots[(n_b - 1) % n_streams][2].cpu().numpy()
        rd, rs, rc_ = (x.cpu().numpy() for x in res)
        if not args.debug and not (np.array_equal(rows[:, :k], rd) and np.array_equal(rows[:, k:2 * k], rs.view(np.int32)) and np.array_equal(rows[:, 2 * k], rc_)):
            raise SystemExit("PARITY FAILURE: a multi-stream search returned rows that differ from the single-stream search")
        pipe = sparse_rx.HostBatchPipeline(ix, nq, len(q_term), k, depth=n_streams, multi_stream=True)
        for i in range(12):
            pipe.result(pipe.submit(q_ptr, q_term, q_w))
        torch.cuda.synchronize(dev)
        tickets = []
        t = time.perf_counter()
        for i in range(n_b):
            tickets.append(pipe.submit(q_ptr, q_term, q_w))
            if len(tickets) == n_streams:
                pd_, ps_, pc_ = pipe.result(tickets.pop(0))
        while tickets:
            pd_, ps_, pc_ = pipe.result(tickets.pop(0))
        host_s = (time.perf_counter() - t) / n_b
        if not args.debug and not (np.array_equal(pd_, rd) and np.array_equal(ps_.view(np.uint32), rs.view(np.uint32)) and np.array_equal(pc_, rc_)):
            raise SystemExit("PARITY FAILURE: the multi-stream host pipeline returned rows that differ from the single-stream search")
        pipe.close()
        steady = {"streams": n_streams, "batches": n_b, "device_resident_qps": nq / dev_s, "device_resident_ms_per_batch": dev_s * 1e3,
                  "pcie_inclusive_qps": nq / host_s, "pcie_inclusive_ms_per_batch": host_s * 1e3}
        log(f"[bench] steady state, {n_streams} batches in flight: {nq / dev_s:,.0f} queries/s device-resident ({dev_s * 1e3:.4f} ms / batch), "
            f"{nq / host_s:,.0f} queries/s host batches in / host rows out ({host_s * 1e3:.4f} ms / batch); single stream: {1e3 * elapsed / args.steps:.4f} ms / batch")
        ix.set_opts(supertile_log2=args.supertile_log2, target_blocks=args.target_blocks, profile=args.profile_every, debug=args.debug, unit_tiles=args.unit_tiles)

    # ---- roofline of the dominant kernel (this rank's scoring kernels) --------------------------------------------
    # sum df_t * (doc id bytes + value bytes) + k * 8, with "the smaller actual per-posting size" when the
    # stored layout is smaller than canonical: the tier-1 kernel streams the compact copy (16-bit unit-local doc ids), the
    # tier-2 kernel the canonical blocks (32-bit doc ids).  The dominant kernel's posting size is the one used; the
    # canonical figure (4-byte doc ids) is reported next to it.
    n_post = int(df_local[qt.long()].sum().item())
    tier1_dominant = prof["wave_ms"] >= prof["block_ms"]
    doc_bytes = 2 if (ix.post16 is not None and (tier1_dominant or ix.post is None)) else 4  # the copy the dominant kernel streams
    post_bytes = doc_bytes + ix.value_bytes
    alg_bytes = n_post * post_bytes + nq * k * 8
    alg_bytes_canonical = n_post * (4 + ix.value_bytes) + nq * k * 8
    # Dominant kernel = the tier-1 wave kernel; the tier-2 block kernel's time is kept in the denominator so that no
    # posting byte is counted without its time.  Per STEP: the per-call averages times the calls one step makes.
    score_s = max((prof["wave_ms"] + prof["block_ms"]) * 1e-3 * calls_per_step, 1e-12)
    achieved = alg_bytes / score_s / 1e9
    alg_total = alg_bytes
    if dist is not None:
        t = torch.tensor([alg_bytes], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        alg_total = int(t.item())
    step_s = elapsed / args.steps
    batch_achieved = alg_total / step_s / 1e9
    traffic, traffic_note = None, "no PMC pass recorded for this workload"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    wl_name = args.workload + ("-custom" if custom else "")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(f"{wl_name}@{world}")
            if ent is not None:
                if ent.get("kernel_src_sha256") == kernel_source_sha():
                    traffic, traffic_note = ent.get("hbm_bytes_per_launch"), ent.get("note")
                else:
                    traffic_note = "the recorded PMC pass profiled a different kernel source (sha256 of csrc/*.hip + srx_common.h differs); re-run tools/gpu_profile.sh"
        except Exception as e:  # pragma: no cover
            traffic_note = f"profiles/traffic.json unreadable: {e}"
    dominant = {"splade": "srx_score_kernel<__half> (tier 2: k > 128)", "zipf": "srx_score_kernel<float> (tier 2: dense tiles)"}.get(
        kind, "srx_wave_kernel<float>")
    if prof["wave_ms"] >= prof["block_ms"]:
        dominant = "srx_wave_kernel<__half>" if kind == "splade" else "srx_wave_kernel<float>"

    result = {
        "metric": "queries/sec + achieved HBM GB/s, batch BM25 top-k=%d" % k,
        "value": nq * args.steps / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f16" if kind == "splade" else "f32",
        "data": "synthetic",
        "config": {"workload": f"{wl_name}: {n_docs} docs x {V} vocab, {w['nnz_per_doc']} nnz/doc, "
                               f"{nq}-query batch x {w['terms']} terms, k={k}; value = device-resident batches "
                               f"(pcie_inclusive_qps = host batches in / host rows out)",
                   "n_docs": n_docs, "vocab": V, "nnz": nnz_total, "n_queries": nq, "k": k,
                   "sharding": f"doc-range x{world}" + ((" + RCCL all-to-all of packed per-shard top-k, merge of the own query block, all-gather of merged rows" if args.exchange == "a2a" else " + one RCCL all-gather of packed per-shard top-k") if (world > 1 or args.force_dist) else ""),
                   "step_submission": "HIP-graph replay, two lanes" if result_graph else "eager launches",
                   "device_index_mb": round(ix.device_bytes() / 2 ** 20, 1), "posting_copies": "compact only" if ix.post is None else "canonical + compact",
                   "index_build_s": round(build_s, 2), "pcie_inclusive_qps": pcie_qps, "pcie_inclusive_ms_per_step": pcie_ms,
                   "steady_state": steady},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note, "kernel": dominant,
                     "kernel_ms": prof["wave_ms"] * calls_per_step, "tier2_kernel_ms": prof["block_ms"] * calls_per_step,
                     "merge_kernel_ms": prof["merge_ms"] * calls_per_step,
                     "launches_timed": prof["calls"], "launches_per_step": calls_per_step, "kernel_timing": kernel_timing,
                     "bytes_per_posting": post_bytes,
                     "achieved_canonical": alg_bytes_canonical / score_s / 1e9,   # the same time against 4-byte doc ids
                     "frac_canonical": alg_bytes_canonical / score_s / 1e9 / HBM_PEAK_GBPS,
                     "algorithmic_bytes_per_launch": alg_bytes / max(calls_per_step, 1e-9),
                     "algorithmic_bytes_per_step_all_gpus": alg_total,
                     "batch_achieved": batch_achieved, "batch_frac": batch_achieved / (HBM_PEAK_GBPS * world),
                     "pcie_inclusive_frac": (alg_total / (pcie_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if pcie_ms else None},
    }

    oracle_mode = None
    if want_check:
        import oracle
        oracle_mode = oracle.MODE_TFIDF_F32 if kind == "splade" else oracle.MODE_BM25_F32
    # ---- --emulate-world rehearsals: prefix + completeness against the shard's oracle list (emulated_bound_check) ----
    if emu_check:
        if rank == 0:
            gpu_rows = tuple(x.contiguous().cpu().numpy() for x in res)
            ok, n_chk, msg = emulated_bound_check(ix, host_csr, idf_np, avgdl, (q_ptr, q_term, q_w), k, oracle_mode, gpu_rows)
            if not ok:
                log("PARITY FAILURE (emulated corpus-wide bounds): " + msg)
                raise SystemExit(3)
            result["parity_check"] = {"queries": n_chk, "shards": 1, "how": f"--emulate-world {args.emulate_world}: GPU rows are a prefix of the "
                                      "shard's oracle list (bit-exact) and hold every oracle row at or above the query's initial threshold"}
            result["cpu_baseline"] = None
    # ---- N > 1 (and --force-dist rehearsals): per-shard oracle sample, host merge, compare with the exchanged rows --
    elif want_check and dist is not None:
        gpu_rows = tuple(x.cpu().numpy() for x in res)
        ok, n_chk = sharded_sample_check(dist, rank, world, host_csr, doc_base, idf_np, avgdl, (q_ptr, q_term, q_w), k, oracle_mode, gpu_rows)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.broadcast(flag, src=0)
        if int(flag.item()) != 1:
            if rank == 0:
                log("PARITY FAILURE: exchanged GPU rows differ from the host merge of the per-shard oracle lists")
            sys.stderr.flush()
            os._exit(3)  # not a collective tear-down with graphs alive: the verdict is what matters here
        result["parity_check"] = {"queries": n_chk, "shards": world, "how": "per-shard oracle (full-CSR scan) on each rank's host CSR, "
                                  "host merge by (score desc, doc asc) on rank 0, bit-exact comparison with the exchanged GPU rows"}
        result["cpu_baseline"] = None
    # ---- CPU baseline (rank 0, N = 1): the oracle on a bounded sample; doubles as the parity check ------------------
    elif want_cpu and rank == 0:
        indptr_h, cols_h, tf_h, dl_h = host_csr
        gd, gs, gc = (x.cpu().numpy() for x in res)

        def run(nsamp):
            qs = (q_ptr[: nsamp + 1] - q_ptr[0], q_term[: q_ptr[nsamp]], q_w[: q_ptr[nsamp]])
            t = time.perf_counter()
            r = oracle.search_batch(indptr_h, cols_h, tf_h, dl_h, idf_np, qs[0], qs[1], qs[2], k, 1.2, 0.75, avgdl, native=True,
                                    mode=oracle_mode)
            return time.perf_counter() - t, r
        t1, _ = run(1)   # also warms the page cache / thread pool
        t1, _ = run(1)
        nsamp = int(max(2, min(nq, 512, args.cpu_seconds / max(t1, 1e-4))))
        tc, (ed, es, ec) = run(nsamp)
        ok = (np.array_equal(gc[:nsamp], ec) and np.array_equal(gd[:nsamp], ed)
              and np.array_equal(gs[:nsamp].view(np.uint32), es.view(np.uint32)))
        if not ok and not args.debug:
            raise SystemExit("PARITY FAILURE: GPU results differ from the CPU oracle on the baseline sample")
        result["cpu_baseline"] = {"value": nsamp / tc, "unit": "queries/s", "cores": oracle.num_threads(native=True),
                                  "kind": "port", "nproc": os.cpu_count(), "cpu_share": _cpu_share(),
                                  "sample": f"first {nsamp} queries of the same batch, full-CSR-scan scorer + top-k "
                                            f"(oracle/bm25_oracle.c, -O3 -march=native, OpenMP, {oracle.num_threads(native=True)} threads "
                                            f"= the cores this process may use; the host has {os.cpu_count()}), {tc:.1f} s; "
                                            "GPU output for the sample verified bit-exact against it",
                                  "cpu": _cpu_model()}
        result["parity_check"] = {"queries": nsamp, "shards": 1, "how": "cpu_baseline sample, bit-exact"}
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    sys.stdout.flush()
    if dist is not None:
        # Tear-down in dependency order: lanes / captured graphs (they hold RCCL work) -> index -> process group.  The result
        # line is out; should the collective tear-down of this stack ever wedge (it did once on a one-rank rehearsal with
        # graphs alive), a watchdog ends the rank with the status it has earned instead of hanging the launcher.
        import threading
        threading.Timer(45.0, lambda: os._exit(0)).start()
        dist.barrier()
        searcher.close()
        ix.close()
        dist.destroy_process_group()
        os._exit(0)
    ix.close()


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

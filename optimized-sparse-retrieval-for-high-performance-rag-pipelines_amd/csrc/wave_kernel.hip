// wave_kernel.hip -- tier 1 of the sparse search: ONE WAVEFRONT per (query, split), no barriers, no MFMA, no float atomics.
//
// Replaces simd_bm25_score + fast_topk_selection (rag_system/core/retrieval.py:41-92) / simd_tfidf_score
// (rag_system/pipeline/evaluate_rag_pipeline.py:95-121) for queries of <= 64 terms at k <= 128.
//
// Layout it relies on (srx_common.h, IndexView): a term's postings are padded runs of 32-byte blocks
// [4 docs | 4 values], one run per unit of <= 65536 docs; padding postings are sentinels (negative doc, value 0) and idle
// loads are redirected to an all-sentinel block of the lane's own (their bitmap words differ per lane / per term: LDS
// atomics of several lanes on one address serialise).  A posting with value 0 is a no-op by construction of every step
// below, so the unit loop needs NO per-posting validity predicate:
//   * query term t owns a group of 64 / 2^ceil(log2 nt) lanes; per step a lane loads one block (two adjacent
//     dwordx4: the group reads one contiguous piece of its run); always W_R / 4 steps, the next unit's loads are in
//     flight while this unit is scored from registers (counted vmcnt waits);
//   * pass 1: every posting ORs  bit = min(value bits, 1) << (doc & 31)  into word (doc >> 5) & 2047 of a wave-private
//     doc bitmap (ds_or_rtn_b32; a unit's docs have distinct low 16 bits, so no rebasing is needed); a sentinel ORs 0.
//     `old & bit` != 0 means an earlier posting matched the same doc: accumulated into one register, tested once per
//     unit.  5 VALU + 1 LDS instruction per posting slot;
//   * docs matched by several terms (about 1.4 per unit on the C3 workload) are resolved in registers: the doc is
//     broadcast with v_readlane, every lane picks up and blanks its posting of it, the contributions are read with
//     v_readlane and added in ascending lane order = the order the query lists its terms = the reference's
//     accumulation order, bit for bit;
//   * the bitmap words are cleared again (ds_write_b32 of the kept addresses);
//   * everything still non-blank is a single-term doc whose score is 0 + c: one v_max3 tree per lane against a
//     conservative per-lane threshold screens them; survivors get the exact fp32 test and go to a lazy LDS list that an
//     exact wave-level radix select shrinks when it fills (srx_common.h).
// Units that do not fit (a run longer than W_R / 4 blocks per lane, more than W_DUPCAP multi-term docs) are flagged
// for tier 2, as are queries with > 64 terms and k > 128.

#include "srx_common.h"

namespace {

struct WaveShared2 {
    unsigned bm[W_BM_WORDS];  // doc bitmap of the current unit; FIRST member: its byte offsets are the DS addresses
    unsigned lbits[W_LCAP];   // lazy top-k list (score bits, doc), unordered
    int ldoc[W_LCAP];
    unsigned hist[256];       // radix histogram of the list selection
};

template <typename VT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W_WAVES_PER_EU))) void srx_wave_kernel(const srx_wave_launch a) {
    __shared__ WaveShared2 S;
    constexpr int BW = BlockWords<VT>::value;
    const IndexView &ix = a.ix;
    const int lane = threadIdx.x;
    const int k = a.k;
    int q, split, nsq;
    decode_item((int)blockIdx.x, a.n_whole, a.n_splits, q, split, nsq);
    if (q >= a.nq) return;
    const int64_t list = (int64_t)q * a.lists_per_q + split;
    const int t0 = a.q_ptr[q];
    const int nt = a.q_ptr[q + 1] - t0;
    const int tpu = ix.unit_tiles;
    if (nt == 0 || nt > W_MAXT || k > W_KMAX || (tpu << ix.tile_log2) > (1 << W_UNIT_LOG2) || (a.dbg & 8)) {  // tier 2 serves it
        if (lane == 0) {
            a.cand_count[list] = 0;
            if (nt > 0) a.work[1 + atomicAdd(&a.work[0], 1)] = (int)blockIdx.x;
        }
        return;
    }
    const int su_lo = (int)(((int64_t)a.n_super * split) / nsq);
    const int su_hi = (int)(((int64_t)a.n_super * (split + 1)) / nsq);
    const int row = ix.n_tiles + 1;

    for (int i = lane; i < W_BM_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(S.bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    wsync();
    WaveTopk tk = {0u, 0u};  // wave-uniform lazy top-k list state
    int sink = 0;            // debug only
    bool flagged = false;    // wave-uniform: some unit of this block was handed to tier 2
    int lg = 0;
    while ((1 << lg) < nt) ++lg;

    // Query term t owns a group of LPT = 64 / 2^ceil(log2 nt) lanes; lane jl of the group handles blocks jl, jl + LPT,
    // jl + 2 LPT, ... of the term's run inside the unit.  Term data stays in registers.  LPT is a compile-time constant
    // of the body (7 instantiations): loads use immediate offsets, no per-step address math.
    auto run = [&](auto lconst) __attribute__((always_inline)) {
        constexpr int LPT_LOG2 = decltype(lconst)::value;
        constexpr int LPT = 1 << LPT_LOG2;
        const int tslot = lane >> LPT_LOG2;  // my term slot (the query's term order = the accumulation order)
        const int jl = lane & (LPT - 1);
        const bool has_term = tslot < nt;
        int64_t tblk = 0;
        const int32_t *skip_row = ix.tile_skip;
        float my_idf = 0.f, my_qw = 0.f;
        if (has_term) {
            const int term = a.q_term[t0 + tslot];
            tblk = ix.term_ptr[term] >> 2;
            skip_row = ix.tile_skip + (int64_t)term * row;
            my_idf = ix.idf[term];
            my_qw = a.q_weight[t0 + tslot];
        }
        // Initial threshold: with all query idf >= 0 a doc's score is at least any single contribution, so the K-th
        // largest contribution of any one term (K >= k, from the index's term_bound table) is an exact lower bound
        // on this shard's k-th best score.  Candidates below it can be dropped from the very first unit.
        {
            const int col = bound_column(k);
            float bnd = 0.0f;
            if (ix.term_bound != nullptr && col >= 0 && has_term && my_idf > 0.0f && my_qw > 0.0f)
                bnd = 0.0f + (ix.term_bound[(int64_t)a.q_term[t0 + tslot] * 4 + col] * my_idf) * my_qw;
            const bool neg = has_term && (my_idf < 0.0f || my_qw < 0.0f);
            const unsigned t0bits = wave_max(__float_as_uint(bnd > 0.0f ? bnd : 0.0f));
            tk.tau = (__ballot(neg) != 0ull) ? 0u : uniu(t0bits);
        }
        unsigned tau_seen = 0xFFFFFFFFu;  // uniform: tau the screening threshold vthr was derived from
        float vthr = 0.0f;
        const int32_t *const zblk = ix.post + (ix.zero_block + lane) * BW;  // my lane's all-sentinel block (doc -1 - 32 lane)
        const int32_t *const tpost = ix.post + (tblk + jl) * BW;       // my lane's first block of the term

        // unit boundary j of my term in BLOCKS from the term's start (#padded postings with doc < j * tpu * G, / 4)
        auto bound = [&](int j) __attribute__((always_inline)) -> int {
            return has_term ? (skip_row[min(j * tpu, ix.n_tiles)] >> 2) : 0;
        };

        // Issue the loads of my term's run [lo, lo + len) (in blocks) of the unit: register r = 4 s + i holds posting i
        // of block s * LPT + jl.  Always exactly 2 * W_R / 4 loads, no branches (idle steps read the sentinel block
        // through a pre-biased pointer, same immediate offset), so that the compiler can wait for THIS unit's data with
        // a counted s_waitcnt vmcnt(N) while the NEXT unit's loads stay in flight.
        auto issue = [&](int lo, int len, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) {
            const int32_t *p = tpost + (int64_t)lo * BW;
            const int rem = len - jl;  // step s is mine iff s * LPT < rem
#pragma unroll
            for (int s4 = 0; s4 < W_R / 4; ++s4) {
                const int off = (s4 << LPT_LOG2) * BW;
                const bool ok = (s4 << LPT_LOG2) < rem;
                int dd[4];
                float vv[4];
                load_block((ok ? p : zblk - off) + off, VT(), dd, vv);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d[4 * s4 + c] = dd[c];
                    v[4 * s4 + c] = vv[c];
                }
            }
        };

        // Score one unit from registers (the first NR of them hold postings).  false -> the unit goes to tier 2
        // (nothing emitted).
        auto process = [&](auto nrc, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) -> bool {
            constexpr int NR = decltype(nrc)::value;
            if (tk.count > (unsigned)(W_LCAP - 64 - W_DUPCAP)) {  // uniform, rare: room for this unit's multi-term docs
                tk.tau = uniu(wave_list_select(S, tk.count, k));
                tk.count = (unsigned)k;
            }
            // ---- pass 1: doc bits ----
            unsigned adr[NR], old[NR], bit[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                adr[r] = ((unsigned)d[r] >> 3) & (unsigned)((W_BM_WORDS - 1) << 2);             // byte offset of word (doc >> 5) & 2047
                unsigned one;  // min(value bits, 1): 0 for a value of exactly +0 (v_min_u32; the compiler's own form is cmp + cndmask)
                asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(__float_as_uint(v[r])));
                bit[r] = one << ((unsigned)d[r] & 31u);
                old[r] = atomicOr(reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]), bit[r]);
            }
            unsigned acc = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) acc |= old[r] & bit[r];
            bool dense = false;
            if (__ballot(acc != 0u) != 0ull) {  // uniform: some doc of this unit is matched by several terms (~2 units in 3 on C3)
                const unsigned count0 = tk.count;
                unsigned n_res = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    // postings that found their doc's bit set, still unresolved (v != 0)
                    unsigned long long m = __ballot((old[r] & bit[r]) != 0u && v[r] != 0.0f);
                    while (m != 0ull && !dense) {  // uniform loop, about one doc per unit on sparse queries
                        const int src = __ffsll((long long)m) - 1;
                        const int dd = __builtin_amdgcn_readlane(d[r], src);  // the doc, wave-uniform
                        // A doc occurs at most once per term, hence at most once per lane (sentinels carry doc -1): pick up
                        // my posting of it (if any) and blank it, so that the single-term screening below never sees it.
                        float myv = 0.0f;
#pragma unroll
                        for (int r2 = 0; r2 < NR; ++r2) {
                            const bool hit = d[r2] == dd;
                            myv = hit ? v[r2] : myv;
                            v[r2] = hit ? 0.0f : v[r2];
                        }
                        const float myc = 0.0f + (myv * my_idf) * my_qw;
                        // exact score: contributions in the query's term order = ascending lane (term slots own lane groups)
                        unsigned long long mm = __ballot(myv != 0.0f);
                        float sum = 0.0f;
                        while (mm != 0ull) {
                            const int l2 = __ffsll((long long)mm) - 1;
                            sum = sum + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(myc), l2));
                            mm &= mm - 1ull;
                        }
                        const unsigned b = __float_as_uint(sum);
                        if (sum > 0.0f && b >= tk.tau) {  // uniform; room for W_DUPCAP entries was made above
                            if (lane == 0) {
                                S.lbits[tk.count] = b;
                                S.ldoc[tk.count] = dd;
                            }
                            ++tk.count;
                        }
                        if (++n_res > (unsigned)W_DUPCAP) dense = true;  // too many for this path: tier 2 takes the unit
                        m = __ballot((old[r] & bit[r]) != 0u && v[r] != 0.0f);
                    }
                }
                if (dense) tk.count = count0;  // nothing of this unit stays in the list
            }
            // ---- clear the bitmap words again ----
#pragma unroll
            for (int r = 0; r < NR; ++r) *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]) = 0u;
            if (dense) return false;
            // ---- single-term docs.  Almost no posting can beat tau once the list has warmed up, so a conservative
            //      per-lane threshold on the stored value (vthr <= the smallest v whose contribution could reach tau, and
            //      > 0 so that blanked registers and sentinels never pass) screens them with one compare; the exact fp32
            //      test runs only for survivors ----
            if (tk.tau != tau_seen) {  // uniform, rare
                tau_seen = tk.tau;
                const float tau_f = __uint_as_float(max(tau_seen, 1u));
                vthr = (my_idf > 0.0f && my_qw > 0.0f) ? fmaxf(((tau_f / my_qw) / my_idf) * 0.99999f, __uint_as_float(1u))
                                                     : __builtin_inff();
            }
            float vmax = v[0];
#pragma unroll
            for (int r = 1; r + 1 < NR; r += 2) vmax = fmaxf(fmaxf(vmax, v[r]), v[r + 1]);
            if constexpr (NR % 2 == 0) vmax = fmaxf(vmax, v[NR - 1]);
            if (__ballot(vmax >= vthr) != 0ull) {  // uniform, rare after warm-up
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const bool pass = v[r] >= vthr;
                    if (__ballot(pass) != 0ull) {
                        const float c = 0.0f + (v[r] * my_idf) * my_qw;
                        const unsigned b = __float_as_uint(c);
                        wave_append(S, tk, k, pass && c > 0.0f && b >= tk.tau, b, d[r]);
                    }
                }
            }
            return true;
        };

        auto flag_tier2 = [&](int su) __attribute__((always_inline)) {
            if (lane == 0) atomicOr(&a.ovf[(int64_t)q * a.ovf_words + (su >> 5)], 1u << (su & 31));
            flagged = true;
        };

        // ---- software pipeline over units, unrolled by two (register sets A / B alternate): issue the loads of
        //      unit u+1, then score unit u from registers ----
        int dA[W_R], dB[W_R];
        float vA[W_R], vB[W_R];
        int b0 = bound(su_lo), b1 = bound(su_lo + 1), b2 = bound(su_lo + 2);  // b_j = boundary j; unit u = [b_u, b_{u+1})
        int lenA = (su_lo < su_hi) ? b1 - b0 : 0, lenB = 0;
        issue(b0, (__ballot(lenA > (W_R / 4) * LPT) == 0ull) ? lenA : 0, dA, vA);
        // one stage: unit su is in (lenc, d, v); unit su+1 goes to (lenn, dn, vn)
        auto stage = [&](int su, int lenc, int (&d)[W_R], float (&v)[W_R], int &lenn, int (&dn)[W_R],
                         float (&vn)[W_R]) __attribute__((always_inline)) {
            const int b3 = bound(su + 3);  // boundary needed two units from now (clamped to the row end)
            lenn = (su + 1 < su_hi) ? b2 - b1 : 0;
            const bool fitn = __ballot(lenn > (W_R / 4) * LPT) == 0ull;  // uniform: every term's run fits the steps
            issue(b1, fitn ? lenn : 0, dn, vn);
            if (a.dbg & 4) {  // timing experiment: loads only (results are wrong)
#pragma unroll
                for (int r = 0; r < W_R; ++r) sink += d[r] ^ (int)__float_as_uint(v[r]);
            } else if (__ballot(lenc > (W_R / 4) * LPT) != 0ull) {
                flag_tier2(su);
            } else if (__ballot(lenc > 0) != 0ull) {
                bool fine;
                bool done = false;
                if constexpr (W_R > 12) {
                    if (__ballot(lenc - jl > 3 * LPT) != 0ull) {  // uniform: the fourth load step holds postings
                        fine = process(IntC<16>{}, d, v);
                        done = true;
                    }
                }
                if constexpr (W_R > 8) {
                    if (!done && __ballot(lenc - jl > 2 * LPT) != 0ull) {  // uniform: the third load step holds postings
                        fine = process(IntC<12>{}, d, v);
                        done = true;
                    }
                }
                if (!done) {
                    if (__ballot(lenc - jl > LPT) != 0ull)
                        fine = process(IntC<8>{}, d, v);
                    else
                        fine = process(IntC<4>{}, d, v);
                }
                if (!fine) flag_tier2(su);
            }
            b1 = b2;
            b2 = b3;
        };
        for (int su = su_lo; su < su_hi; su += 2) {
            stage(su, lenA, dA, vA, lenB, dB, vB);
            if (su + 1 < su_hi) stage(su + 1, lenB, dB, vB, lenA, dA, vA);
        }
    };
    switch (6 - lg) {
        case 0: run(IntC<0>{}); break;
        case 1: run(IntC<1>{}); break;
        case 2: run(IntC<2>{}); break;
        case 3: run(IntC<3>{}); break;
        case 4: run(IntC<4>{}); break;
        case 5: run(IntC<5>{}); break;
        default: run(IntC<6>{}); break;
    }
    if ((a.dbg & 4) && sink == 0x7F123457) a.cand_count[list] = sink;  // keeps the loads of the timing experiment alive
    unsigned count = tk.count;
    if (a.dbg & 32) count = 0;  // timing experiment: no final selection / ranking
    if (count > (unsigned)k) {
        wave_list_select(S, count, k);
        count = (unsigned)k;
    }
    if (nsq == 1 && !flagged && a.out_doc != nullptr) {
        // This wave holds the query's complete top-k (one split, nothing handed to tier 2): rank it here and write
        // the final row, so the merge kernel can skip the query.
        wave_rank_emit(S, reinterpret_cast<unsigned long long *>(S.bm), count, k, a.doc_base, a.out_doc + (int64_t)q * a.out_row_stride,
                       a.out_score + (int64_t)q * a.out_row_stride);
        if (lane == 0) {
            a.out_count[(int64_t)q * a.out_cnt_stride] = (int)count;
            a.cand_count[list] = -1;  // tells the merge kernel this query is final
        }
        return;
    }
    const int64_t o = list * k;
    for (unsigned i = lane; i < count; i += 64) {
        a.cand_doc[o + i] = S.ldoc[i];
        a.cand_score[o + i] = __uint_as_float(S.lbits[i]);
    }
    if (lane == 0) {
        a.cand_count[list] = (int)count;
        if (flagged) a.work[1 + atomicAdd(&a.work[0], 1)] = (int)blockIdx.x;
    }
}

}  // namespace

int srx_launch_wave_kernel(const srx_wave_launch &a, int val_type, int64_t blocks, hipStream_t stream) {
    if (blocks <= 0) return SRX_OK;
    if (val_type == SRX_VAL_F32)
        hipLaunchKernelGGL(srx_wave_kernel<float>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(srx_wave_kernel<__half>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

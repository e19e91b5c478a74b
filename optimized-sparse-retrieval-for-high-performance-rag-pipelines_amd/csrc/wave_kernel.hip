// wave_kernel.hip -- tier 1 of the sparse search: ONE WAVEFRONT per (query, split), no barriers, no MFMA, no float atomics.
//
// Replaces simd_bm25_score + fast_topk_selection (rag_system/core/retrieval.py:41-92) / simd_tfidf_score
// (rag_system/pipeline/evaluate_rag_pipeline.py:95-121) for queries of <= 64 terms at k <= 112 (W1_KMAX).
//
// Layout it relies on (srx_common.h, IndexView): a term's postings are padded runs of blocks [4 docs | 4 values], one run
// per unit of <= 63488 docs; tier 1 streams the COMPACT copy of the blocks (post16: 16-bit unit-local doc ids, 24 bytes per
// block with fp32 values, 16 with fp16 -- 6 / 4 bytes per posting instead of 8 / 6: the kernel runs at the HBM ceiling, so
// bytes are time); padding postings are sentinels (local id 0xFFFF - 32 x, above every real local id, value 0) and idle
// loads are redirected to an all-sentinel block of the lane's own (their bitmap words differ per lane / per term: LDS
// atomics of several lanes on one address serialise).  A posting with value 0 is a no-op by construction of every step
// below, so the unit loop needs NO per-posting validity predicate:
//   * query term t owns a group of 64 / 2^ceil(log2 nt) lanes; per step a lane loads one block (two adjacent
//     dwordx4: the group reads one contiguous piece of its run); always W_R / 4 steps, the next unit's loads are in
//     flight while this unit is scored from registers (counted vmcnt waits);
//   * pass 1: every posting ORs  bit = min(value bits, 1) << (doc & 31)  into word (doc >> 5) & 2047 of a wave-private
//     doc bitmap (ds_or_rtn_b32; doc = the unit-local id); a sentinel ORs 0.
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
// for tier 2, as are queries with > 64 terms and k > W1_KMAX.

#include "srx_common.h"

#ifdef SRX_STAMP
#define CNT(i) (++st_cnt[i])
#else
#define CNT(i) \
    do {       \
    } while (0)
#endif

namespace {

#ifndef SRX_W_WPE
#define SRX_W_WPE W_WAVES_PER_EU
#endif
#ifndef SRX_W_DEPTH
#define SRX_W_DEPTH 2  // register sets: units in flight + the one being scored
#endif
#ifndef SRX_W_PREFETCH
#define SRX_W_PREFETCH 0  // 1: touch the lines of the unit after the one being loaded (see fetch).  Measured and left off: with the
                          // touch's waits counted correctly the C3 batch takes 2.12 ms instead of 1.26 -- every line is requested
                          // twice (touch, then the real load after the L1 has dropped it) and the L2 / fabric request rate, not
                          // HBM latency, is what the kernel is up against (loads-only already streams at the HBM ceiling)
#endif
struct WaveShared2 {
    static constexpr int LCAP = W1_LCAP;
    static constexpr bool HIST_ALIASES_ZEROED_LDS = true;
    union {
        unsigned bm[W_BM_WORDS];  // doc bitmap of the current unit; FIRST member: its byte offsets are the DS addresses
        unsigned hist[256];       // radix histogram of the list selection: borrows the bitmap's first words (a selection only
                                  // runs between units, when the bitmap is all zero) and zeroes them again when it is done
    };
    unsigned lbits[LCAP];         // lazy top-k list (score bits, doc), unordered
    int ldoc[LCAP];
};

// d[rs] of lane src for a wave-uniform register index rs: a jump over v_readlane instructions
template <int NR>
__device__ __forceinline__ int lane_reg(const int (&d)[W_R], int rs, int src) {
    switch (rs) {
#define SRX_CASE(i) \
    case i:         \
        return __builtin_amdgcn_readlane(d[(i) < NR ? (i) : 0], src);
        SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7) SRX_CASE(8) SRX_CASE(9) SRX_CASE(10)
        SRX_CASE(11) SRX_CASE(12) SRX_CASE(13) SRX_CASE(14) SRX_CASE(15)
#undef SRX_CASE
        default:
            return __builtin_amdgcn_readlane(d[0], src);
    }
}

template <typename VT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SRX_W_WPE))) void srx_wave_kernel(const srx_wave_launch a) {
    __shared__ WaveShared2 S;
    constexpr int BW = CompactWords<VT>::value;  // tier 1 streams the compact copy (16-bit unit-local docs)
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
    if (nt == 0 || tier1_cannot_serve(ix, nt, k, tpu, a.dbg)) {  // tier 2 serves it
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
#ifdef SRX_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
    unsigned st_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
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
        const int32_t *const zblk = ix.post16 + (ix.zero_block + lane) * BW;  // my lane's all-sentinel block (local id 0xFFFF - 32 lane)
        const int32_t *const tpost = ix.post16 + (tblk + jl) * BW;       // my lane's first block of the term

        // unit boundary j of my term in padded POSTINGS from the term's start (#padded postings with doc < j * tpu * G; a
        // multiple of 4).  Loaded by every lane without a branch (lanes without a term read term 0's row: valid memory,
        // masked where the run length is formed) and consumed SRX_W_DEPTH + 1 fetches later: a load inside a divergent
        // branch made the compiler finish it on the spot with s_waitcnt vmcnt(0), which also drained the posting loads of
        // the next unit issued just before it -- the wave then had nothing in flight while it scored.
        auto bound = [&](int j) __attribute__((always_inline)) -> int { return gload_i32(skip_row + min(j * tpu, ix.n_tiles)); };

        // Issue the loads of my term's run [lo, lo + len) (in blocks) of the unit: register r = 4 s + i holds posting i
        // of block s * LPT + jl.  Always exactly 2 * W_R / 4 loads, no branches (idle steps read a sentinel block
        // through the other pointer, same immediate offset), so that the compiler can wait for THIS unit's data with
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
                load_block16((ok ? p : zblk) + off, VT(), dd, vv);  // idle: sentinel block lane + s4 * LPT (SRX_BLOCK_PAD covers it)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d[4 * s4 + c] = dd[c];
                    v[4 * s4 + c] = vv[c];
                }
            }
        };

        // Score one unit from registers (the first NR of them hold postings).  false -> the unit goes to tier 2
        // (nothing emitted).
        auto process = [&](auto nrc, int ubase, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) -> bool {  // d = unit-local ids, ubase = the unit's first doc
            constexpr int NR = decltype(nrc)::value;
            if (tk.count > (unsigned)(WaveShared2::LCAP - 64 - W_DUPCAP)) {  // uniform, rare: room for this unit's multi-term docs
                tk.tau = uniu(wave_list_select(S, tk.count, k));
                tk.count = (unsigned)k;
            }
            // ---- pass 1: doc bits ----
            unsigned adr[NR], t[NR];  // t[r] != 0: posting r found its doc's bit already set (an earlier posting matched the doc)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                adr[r] = ((unsigned)d[r] >> 3) & (unsigned)((W_BM_WORDS - 1) << 2);             // byte offset of word (doc >> 5) & 2047
                unsigned one;  // min(value bits, 1): 0 for a value of exactly +0 (v_min_u32; the compiler's own form is cmp + cndmask)
                asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(__float_as_uint(v[r])));
                const unsigned bit = one << ((unsigned)d[r] & 31u);
                t[r] = atomicOr(reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]), bit) & bit;
            }
            unsigned acc = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) acc |= t[r];
            bool dense = false;
            const bool anydup = __ballot(acc != 0u) != 0ull;
            STAMP(2);  // wait for the unit's postings + pass 1
            CNT(0);
            if (anydup && !(a.dbg & 1)) {  // uniform: some doc of this unit is matched by several terms (~3 units in 4 on C3)
                CNT(1);
                // Lanes with a flagged posting (typically one or two) are visited one after the other: fm = the lane's
                // flagged slots; the doc of its lowest flagged slot is broadcast, every lane picks up and blanks its
                // posting of that doc (a doc occurs at most once per term, hence at most once per lane; sentinels carry
                // local ids no real posting has), and the contributions are added in ascending lane order = the query's term order.
                unsigned fm = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    unsigned one;
                    asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(t[r]));
                    fm |= one << r;
                }
                const unsigned count0 = tk.count;
                unsigned n_res = 0;
                unsigned long long m = __ballot(fm != 0u);
                while (m != 0ull) {  // uniform loop: about two docs per such unit on C3
                    const int src = __ffsll((long long)m) - 1;
                    const unsigned fms = (unsigned)__builtin_amdgcn_readlane((int)fm, src);
                    const int rs = __ffs((int)fms) - 1;                 // uniform: lane src's lowest flagged slot
                    const int dd = lane_reg<NR>(d, rs, src);            // its doc, wave-uniform
                    CNT(2);
                    if (a.dbg & 512) {  // timing experiment: locate the docs only
                        sink += dd;
                        fm = (lane == src) ? (fm & (fm - 1u)) : fm;
                        m = __ballot(fm != 0u);
                        continue;
                    }
                    float myv = 0.0f;
#pragma unroll
                    for (int r2 = 0; r2 < NR; ++r2) {
                        const bool hit = d[r2] == dd;
                        myv = hit ? v[r2] : myv;
                        v[r2] = hit ? 0.0f : v[r2];
                    }
                    const float myc = 0.0f + (myv * my_idf) * my_qw;
                    unsigned long long mm = __ballot(myv != 0.0f);  // empty when an earlier visit resolved this doc (3 or more terms)
                    float sum = 0.0f;
                    while (mm != 0ull) {
                        const int l2 = __ffsll((long long)mm) - 1;
                        sum = sum + __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(myc), l2));
                        mm &= mm - 1ull;
                    }
                    const unsigned b = __float_as_uint(sum);
                    if (sum > 0.0f && b >= tk.tau && !(a.dbg & 1024)) {  // uniform; room for W_DUPCAP entries was made above
                        if (lane == 0) {
                            S.lbits[tk.count] = b;
                            S.ldoc[tk.count] = ubase + dd;
                        }
                        ++tk.count;
                    }
                    if (++n_res > (unsigned)W_DUPCAP) {  // too many for this path: tier 2 takes the unit, nothing of it stays in the list
                        dense = true;
                        tk.count = count0;
                        break;
                    }
                    fm = (lane == src) ? (fm & (fm - 1u)) : fm;  // that slot is done
                    m = __ballot(fm != 0u);
                }
            }
            STAMP(3);  // multi-term docs
            // ---- clear the bitmap words again ----
#pragma unroll
            for (int r = 0; r < NR; ++r) *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]) = 0u;
            STAMP(4);  // restore
            if (dense) return false;
            if (a.dbg & 2) return true;  // timing experiment: no candidate screening (results are wrong)
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
            const bool anycand = __ballot(vmax >= vthr) != 0ull;
            STAMP(5);  // screening
            if (anycand) {  // uniform, rare after warm-up
                CNT(3);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const bool pass = v[r] >= vthr;
                    if (__ballot(pass) != 0ull) {
                        const float c = 0.0f + (v[r] * my_idf) * my_qw;
                        const unsigned b = __float_as_uint(c);
                        wave_append(S, tk, k, pass && c > 0.0f && b >= tk.tau, b, ubase + d[r]);
                    }
                }
            }
            STAMP(6);  // candidates (appends, selections)
            return true;
        };

        auto flag_tier2 = [&](int su) __attribute__((always_inline)) {
            if (lane == 0) atomicOr(&a.ovf[(int64_t)q * a.ovf_words + (su >> 5)], 1u << (su & 31));
            flagged = true;
        };

        // ---- software pipeline over units: SRX_W_DEPTH register sets rotate; while unit u is scored from registers, the
        //      loads of units u+1 .. u+DEPTH-1 are in flight (a wave has no other way to keep memory requests outstanding:
        //      with one unit ahead the data arrives long before the unit before it has been scored, and nothing is in
        //      flight for the rest of that time) ----
        // score unit su held in (d, v) with run length lenc (blocks)
        auto score = [&](int su, int lenc, int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) {
            if (a.dbg & 4) {  // timing experiment: loads only (results are wrong)
#pragma unroll
                for (int r = 0; r < W_R; ++r) sink += d[r] ^ (int)__float_as_uint(v[r]);
            } else if (__ballot(lenc > (W_R / 4) * LPT) != 0ull) {
                flag_tier2(su);
            } else if (__ballot(lenc > 0) != 0ull) {
                bool fine;
                bool done = false;
                const int ubase = (su * tpu) << ix.tile_log2;
                if constexpr (W_R > 12) {
                    if (__ballot(lenc - jl > 3 * LPT) != 0ull) {  // uniform: the fourth load step holds postings
                        fine = process(IntC<16>{}, ubase, d, v);
                        done = true;
                    }
                }
                if constexpr (W_R > 8) {
                    if (!done && __ballot(lenc - jl > 2 * LPT) != 0ull) {  // uniform: the third load step holds postings
                        fine = process(IntC<12>{}, ubase, d, v);
                        done = true;
                    }
                }
                if (!done) {
                    if (__ballot(lenc - jl > LPT) != 0ull)
                        fine = process(IntC<8>{}, ubase, d, v);
                    else
                        fine = process(IntC<4>{}, ubase, d, v);
                }
                if (!fine) flag_tier2(su);
            }
        };
        // issue unit su's loads into (d, v); returns its run length (0 past the end; a run that does not fit loads nothing)
#if SRX_W_PREFETCH
        int pf_val = 0;
#endif
        constexpr int NBQ = SRX_W_DEPTH + 2 + SRX_W_PREFETCH;  // the touch looks one unit further ahead
        int bq[NBQ];  // bq[i] = boundary (next unit to issue) + i, in padded postings
        int su_issue = su_lo;
#pragma unroll
        for (int i = 0; i < NBQ; ++i) bq[i] = bound(su_lo + i);
        auto fetch = [&](int (&d)[W_R], float (&v)[W_R]) __attribute__((always_inline)) -> int {
            const int len = (has_term && su_issue < su_hi) ? (bq[1] - bq[0]) >> 2 : 0;  // blocks
            const bool fit = __ballot(len > (W_R / 4) * LPT) == 0ull;  // uniform: every term's run fits the steps
            STAMP(0);  // loop overhead / previous tail
#if SRX_W_PREFETCH
            {
                // Touch the 128-byte lines of the unit AFTER the one loaded below (one dword per line, lane jl takes line jl of
                // my term's run; PF_STRIDE blocks per line): those lines are on their way from HBM to L2 while this unit's
                // loads are in flight, so the wave keeps two units' worth of requests outstanding with ONE more register --
                // the kernel is limited by bytes in flight (16 waves per CU x one unit each), and a third register set costs a
                // wave per SIMD.  The touch is issued BEFORE this unit's loads and its value is consumed at the next fetch:
                // loads return in order, so a wait for a touch issued AFTER a unit's loads is a wait for that whole unit.
                sink ^= pf_val;  // the touch issued one fetch ago, older than that fetch's posting loads (a counted wait)
                constexpr int PF_STRIDE = 128 / (BW * 4);  // whole blocks per line (4 for fp32 values, 5 for fp16)
                const int nlen = (has_term && su_issue + 1 < su_hi) ? (bq[2] - bq[1]) >> 2 : 0;
                const int pb = min(PF_STRIDE * jl, nlen - 1);  // the lane after the last whole line takes the run's last block
                const bool pok = PF_STRIDE * jl < nlen + PF_STRIDE - 1;
                // lanes with nothing to touch re-read block 0 of their term (valid memory, already cached): no branch, no
                // second pointer
                pf_val = gload_i32(tpost + (int64_t)((pok ? (bq[1] >> 2) + pb : jl) - jl) * BW);
                __builtin_amdgcn_sched_barrier(0);  // the scheduler must not move the touch behind the posting loads
            }
#endif
            issue(bq[0] >> 2, fit ? len : 0, d, v);
            STAMP(1);  // issue
#pragma unroll
            for (int i = 0; i < NBQ - 1; ++i) bq[i] = bq[i + 1];
            ++su_issue;
            bq[NBQ - 1] = bound(su_issue + NBQ - 1);  // needed NBQ - 2 fetches from now (clamped to the row end)
            return len;
        };
#if SRX_W_DEPTH == 2
        int dA[W_R], dB[W_R];
        float vA[W_R], vB[W_R];
        int lenA = fetch(dA, vA), lenB = 0;
        for (int su = su_lo; su < su_hi; su += 2) {
            lenB = fetch(dB, vB);
            score(su, lenA, dA, vA);
            if (su + 1 < su_hi) {
                lenA = fetch(dA, vA);
                score(su + 1, lenB, dB, vB);
            }
        }
#else
        int dA[W_R], dB[W_R], dC[W_R];
        float vA[W_R], vB[W_R], vC[W_R];
        int lenA = fetch(dA, vA), lenB = fetch(dB, vB), lenC = 0;
        for (int su = su_lo; su < su_hi; su += 3) {
            lenC = fetch(dC, vC);
            score(su, lenA, dA, vA);
            if (su + 1 < su_hi) {
                lenA = fetch(dA, vA);
                score(su + 1, lenB, dB, vB);
            }
            if (su + 2 < su_hi) {
                lenB = fetch(dB, vB);
                score(su + 2, lenC, dC, vC);
            }
        }
#endif
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
    if ((a.dbg & (4 | 512)) && sink == 0x7F123457) a.cand_count[list] = sink;  // keeps the loads of the timing experiment alive
    if (a.nq < 0 && sink == 0x7F123457) a.cand_count[list] = sink;               // never true (nq >= 0): keeps the touch loads alive
    unsigned count = tk.count;
    if (a.dbg & 32) count = 0;  // timing experiment: no final selection / ranking
    if (count > (unsigned)k) {
        wave_list_select(S, count, k);
        count = (unsigned)k;
    }
#ifdef SRX_STAMP
    STAMP(7);  // epilogue (final select)
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp[i], st_acc[i]);
        atomicAdd(&g_stamp[8], 1ull);
        for (int i = 0; i < 7; ++i) atomicAdd(&g_stamp[9 + i], (unsigned long long)st_cnt[i]);
    }
#endif
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

#ifdef SRX_STAMP
extern "C" __attribute__((visibility("default"))) int srx_debug_read_stamps(unsigned long long *h_out16) {
    HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)));
    return SRX_OK;
}
#endif

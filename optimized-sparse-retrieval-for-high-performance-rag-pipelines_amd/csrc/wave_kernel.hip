// wave_kernel.hip -- tier 1 of the sparse search: ONE WAVEFRONT per (query, split), no barriers, no MFMA, no float atomics.
//
// Replaces simd_bm25_score + fast_topk_selection (rag_system/core/retrieval.py:41-92) / simd_tfidf_score
// (rag_system/pipeline/evaluate_rag_pipeline.py:95-121) for queries of <= 64 terms at k <= 112 (W1_KMAX).
//
// Layout it relies on (srx_common.h, IndexView): a term's postings are padded runs of blocks [4 docs | 4 values], one run
// per unit of <= 49152 docs; tier 1 streams the COMPACT copy of the blocks (post16: 16-bit unit-local doc ids, 24 bytes per
// block with fp32 values, 16 with fp16 -- 6 / 4 bytes per posting instead of 8 / 6: the kernel runs at the HBM ceiling, so
// bytes are time); padding postings are sentinels (local id 49152 + 32 x, above every real local id, value 0) and idle
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

enum { U_OK = 0, U_DENSE = 1, U_FULL = 2 };  // outcome of scoring one unit (see process)

#ifndef SRX_W_DEPTH
#define SRX_W_DEPTH 2  // register sets: units in flight + the one being scored
#endif
struct WaveShared2 {  // 6400 + 8 LCAP bytes: 7936 at 5 waves per SIMD (20 x 7936 <= 160 KB), 8448 at 4
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
static_assert(sizeof(WaveShared2) * 4 * W_WAVES_PER_EU <= 160 * 1024, "tier-1 LDS per wave does not allow the waves per SIMD the kernel is built for");

// The unit-local doc ids stay PACKED in registers the way the compact copy stores them (two 16-bit ids per word: slot r
// lives in half r & 1 of word r >> 1): a register set is W_R / 2 + W_R VGPRs instead of 2 W_R, which is what pays for the
// third set in flight (SRX_W_DEPTH).  Pass 1 reads the halves with shifts that cost what the unpacked form's did.
constexpr int W_RP = W_R / 2;
__device__ __forceinline__ unsigned slot_id(const unsigned (&dp)[W_RP], int r) {  // r: compile-time constant after unrolling
    return (r & 1) ? dp[r >> 1] >> 16 : dp[r >> 1] & 0xFFFFu;
}
// id of slot rs of lane src for a wave-uniform slot index rs: a jump over v_readlane instructions (one per word)
template <int NR>
__device__ __forceinline__ int lane_reg(const unsigned (&dp)[W_RP], int rs, int src) {
    unsigned w;
    switch (rs >> 1) {
#define SRX_CASE(i) \
    case i:         \
        w = (unsigned)__builtin_amdgcn_readlane((int)dp[(i) < NR / 2 ? (i) : 0], src); \
        break;
        SRX_CASE(1) SRX_CASE(2) SRX_CASE(3) SRX_CASE(4) SRX_CASE(5) SRX_CASE(6) SRX_CASE(7)
#undef SRX_CASE
        default:
            w = (unsigned)__builtin_amdgcn_readlane((int)dp[0], src);
    }
    return (int)((rs & 1) ? w >> 16 : w & 0xFFFFu);
}

// DBG: the ablation build of the same kernel (bench.py --debug: timing experiments with WRONG results).  The shipped instance
// (DBG = false) carries none of those tests in its unit loop.
template <typename VT, bool DBG>
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
    const int n_terms_all = a.q_ptr[a.nq];  // length of q_term / q_weight (the scalar prologue reads whole vectors of them)
    const int tpu = ix.unit_tiles;
    // This item's slots of the workspace are initialised here, not by a memset in front of the kernel (which was two fill
    // launches, ~10 us of a 0.2 ms shard step): its tier-2 list reads as empty and its tier-2 flags as clear unless
    // something is written later.  (Atomic stores: they reach L2 in order with the atomicOr of flag_tier2.)
    unsigned *const my_ovf = a.ovf + (int64_t)blockIdx.x * a.ovf_words;
    for (int w = lane; w < a.ovf_words; w += 64) __hip_atomic_store(&my_ovf[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) a.cand_count[(int64_t)q * a.lists_per_q + a.n_splits + split] = 0;
    if (nt == 0 || tier1_cannot_serve(ix, nt, k, tpu, a.dbg)) {  // tier 2 serves it
        if (lane == 0) {
            a.cand_count[list] = 0;
            if (nt > 0) a.work[1 + atomicAdd(&a.work[0], 1)] = (int)blockIdx.x;
        }
        if (nt == 0 && nsq == 1 && a.out_doc != nullptr) {  // an unsplit query without terms: nobody else writes its (empty) row
            for (int i = lane; i < k; i += 64) {
                a.out_doc[(int64_t)q * a.out_row_stride + i] = -1;
                a.out_score[(int64_t)q * a.out_row_stride + i] = 0.0f;
            }
            if (lane == 0) {
                a.out_count[(int64_t)q * a.out_cnt_stride] = 0;
                a.cand_count[list] = -1;
            }
        }
        return;
    }
    const int su_lo = (int)(((int64_t)a.n_super * split) / nsq);
    const int su_hi = (int)(((int64_t)a.n_super * (split + 1)) / nsq);
    const int row = ix.n_tiles + 1;

    for (int i = lane; i < W_BM_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(S.bm)[i] = make_uint4(0u, 0u, 0u, 0u);
    wsync();
    WaveTopk tk = {0u, 0u};  // wave-uniform lazy top-k list state
    int sink = 0;            // DBG only
    const int dbg = DBG ? a.dbg : 0;
#ifdef SRX_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime();
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
        unsigned skip_boff = 0;  // BYTE offset of my term's skip row (the table is < 4 GiB: tier1_cannot_serve): one VGPR, and the
                                 // loads take the table's base from SGPRs (global_load ... v_off, s[base])
        float my_idf = 0.f, my_qw = 0.f, my_bnd = 0.f;
        constexpr int NBQ = SRX_W_DEPTH + 2;
        int bq[NBQ];             // bq[i] = boundary (next unit to issue) + i of my term, in padded postings
        const int col = bound_column(k);
        const bool use_bound = ix.term_bound != nullptr && col >= 0;
        constexpr bool SCALAR_PROLOGUE = LPT_LOG2 >= 3;  // <= 8 term slots
        if constexpr (SCALAR_PROLOGUE) {
            // Everything a query needs before its first posting load -- its terms, their weights, run starts, idf, score
            // bounds and first unit boundaries -- is WAVE-UNIFORM data (one value per term, <= 8 terms).  It is read with
            // SCALAR loads: they do not queue behind the ~40 KB of posting loads the CU's other waves keep in the vector memory
            // pipeline, which made each of the four dependent round trips of this prologue cost about as much as a whole
            // unit (the stamps: 20 % of a wave's time on a 1.25 M-doc shard).  The values reach the term's lane group
            // through one compare + select per term.
            constexpr int NTS = 64 >> LPT_LOG2;
            // the NTS term ids and weights: one wide scalar load each (slots >= nt read the next queries' terms -- valid ids,
            // never used: has_term); the batch's last queries, where that would run past the arrays, repeat their last term
            int s_term[NTS];
            float s_qwt[NTS];
            typedef int srx_ivec __attribute__((ext_vector_type(NTS < 2 ? 2 : NTS), aligned(4)));
            typedef float srx_fvec __attribute__((ext_vector_type(NTS < 2 ? 2 : NTS), aligned(4)));
            if (t0 + (NTS < 2 ? 2 : NTS) <= n_terms_all) {  // uniform
                const srx_ivec tv = *(const SRX_CONSTANT srx_ivec *)(a.q_term + t0);
                const srx_fvec wv = *(const SRX_CONSTANT srx_fvec *)(a.q_weight + t0);
#pragma unroll
                for (int i = 0; i < NTS; ++i) {
                    s_term[i] = tv[i];
                    s_qwt[i] = wv[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NTS; ++i) {
                    s_term[i] = cload_i32(a.q_term + t0 + min(i, nt - 1));
                    s_qwt[i] = cload_f32(a.q_weight + t0 + min(i, nt - 1));
                }
            }
            // (value & m) | (old & ~m) with m = all ones in the term's lanes: one v_bfi per value.  Written with selects the
            // compiler sinks every scalar load into a branch on `mine` and waits for it there: eight round trips in a row.
            auto pick = [](unsigned m, unsigned x, unsigned old) __attribute__((always_inline)) { return (x & m) | (old & ~m); };
            unsigned v_tlo = 0, v_thi = 0, v_idf = 0, v_qw = 0, v_bnd = 0, v_bq[2] = {0u, 0u};
            // Batches of <= 4 terms: all scalar loads of a batch are issued before the first one is waited for (scalar loads
            // return out of order, so a wait is always for all of them; ~7 SGPRs per term -- with all 8 terms in one batch the
            // allocator spills SGPRs, and a spill waits for its load).  Only the first TWO unit boundaries come this way; the
            // later ones are loaded by the lanes themselves and arrive behind the first unit's postings.
            constexpr int TB = NTS < 4 ? NTS : 4;
            auto batches = [&](auto with_bound) __attribute__((always_inline)) {  // two instances: no branch sits between the loads of a batch
                constexpr bool WB = decltype(with_bound)::value != 0;
#pragma unroll
                for (int h = 0; h < NTS; h += TB) {
                    int64_t s_tp[TB];
                    float s_idf[TB], s_b[TB];
                    int s_q0[TB], s_q1[TB];
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        const int ti = s_term[h + j];
                        s_tp[j] = cload_i64(ix.term_ptr + ti);
                        s_idf[j] = cload_f32(ix.idf + ti);
                        s_b[j] = WB ? cload_f32(ix.term_bound + (int64_t)ti * 4 + col) : 0.0f;
                        s_q0[j] = cload_i32(ix.tile_skip + (int64_t)ti * row + min(su_lo * tpu, ix.n_tiles));
                        s_q1[j] = cload_i32(ix.tile_skip + (int64_t)ti * row + min((su_lo + 1) * tpu, ix.n_tiles));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        const int ti = s_term[h + j];
                        const unsigned m = tslot == h + j ? 0xFFFFFFFFu : 0u;
                        const int64_t tb = s_tp[j] >> 2;
                        v_tlo = pick(m, (unsigned)(tb & 0xFFFFFFFFll), v_tlo);
                        v_thi = pick(m, (unsigned)(tb >> 32), v_thi);
                        skip_boff = pick(m, ((unsigned)ti * (unsigned)row) << 2, skip_boff);
                        v_idf = pick(m, __float_as_uint(s_idf[j]), v_idf);
                        v_qw = pick(m, __float_as_uint(s_qwt[h + j]), v_qw);
                        v_bnd = pick(m, __float_as_uint(s_b[j]), v_bnd);
                        v_bq[0] = pick(m, (unsigned)s_q0[j], v_bq[0]);
                        v_bq[1] = pick(m, (unsigned)s_q1[j], v_bq[1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (use_bound)
                batches(IntC<1>{});
            else
                batches(IntC<0>{});
            bq[0] = (int)v_bq[0];
            bq[1] = (int)v_bq[1];
            tblk = (int64_t)(((unsigned long long)v_thi << 32) | v_tlo);
            my_idf = __uint_as_float(v_idf);
            my_qw = __uint_as_float(v_qw);
            my_bnd = __uint_as_float(v_bnd);
        } else if (has_term) {
            const int term = a.q_term[t0 + tslot];
            tblk = ix.term_ptr[term] >> 2;
            skip_boff = ((unsigned)term * (unsigned)row) << 2;
            my_idf = ix.idf[term];
            my_qw = a.q_weight[t0 + tslot];
            if (use_bound) my_bnd = ix.term_bound[(int64_t)term * 4 + col];
        }
        // Initial threshold: with all query idf >= 0 a doc's score is at least any single contribution, so the K-th
        // largest contribution of any one term (K >= k, from the index's term_bound table) is an exact lower bound
        // on this shard's k-th best score.  Candidates below it can be dropped from the very first unit.
        {
            float bnd = 0.0f;
            if (use_bound && has_term && my_idf > 0.0f && my_qw > 0.0f) bnd = 0.0f + (my_bnd * my_idf) * my_qw;
            const bool neg = has_term && (my_idf < 0.0f || my_qw < 0.0f);
            const unsigned t0bits = wave_max(__float_as_uint(bnd > 0.0f ? bnd : 0.0f));
            tk.tau = (__ballot(neg) != 0ull) ? 0u : uniu(t0bits);
        }
        unsigned tau_seen = 0xFFFFFFFFu;  // uniform: tau the screening threshold vthr was derived from
        float vthr = 0.0f;
        const int32_t *const zblk = ix.post16 + (ix.zero_block + lane) * BW;  // my lane's all-sentinel block (local id W_SENT_BASE + 32 lane)
        const int32_t *const tpost = ix.post16 + (tblk + jl) * BW;       // my lane's first block of the term

        // unit boundary j of my term in padded POSTINGS from the term's start (#padded postings with doc < j * tpu * G; a
        // multiple of 4).  Loaded by every lane without a branch (lanes without a term read term 0's row: valid memory,
        // masked where the run length is formed) and consumed SRX_W_DEPTH + 1 fetches later: a load inside a divergent
        // branch made the compiler finish it on the spot with s_waitcnt vmcnt(0), which also drained the posting loads of
        // the next unit issued just before it -- the wave then had nothing in flight while it scored.
        auto bound = [&](int j) __attribute__((always_inline)) -> int {
            return gload_i32(reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(ix.tile_skip) +
                                                               (size_t)(skip_boff + ((unsigned)min(j * tpu, ix.n_tiles) << 2))));
        };

        // Issue the loads of my term's run [lo, lo + len) (in blocks) of the unit: register r = 4 s + i holds posting i
        // of block s * LPT + jl.  Always exactly 2 * W_R / 4 loads, no branches (idle steps read a sentinel block
        // through the other pointer, same immediate offset), so that the compiler can wait for THIS unit's data with
        // a counted s_waitcnt vmcnt(N) while the NEXT unit's loads stay in flight.
        auto issue = [&](int lo, int len, unsigned (&d)[W_RP], float (&v)[W_R]) __attribute__((always_inline)) {
            const int32_t *p = tpost + (int64_t)lo * BW;
            const int rem = len - jl;  // step s is mine iff s * LPT < rem
#pragma unroll
            for (int s4 = 0; s4 < W_R / 4; ++s4) {
                const int off = (s4 << LPT_LOG2) * BW;
                const bool ok = (s4 << LPT_LOG2) < rem;
                unsigned dd[2];
                float vv[4];
                load_block16p((ok ? p : zblk) + off, VT(), dd, vv);  // idle: sentinel block lane + s4 * LPT (SRX_BLOCK_PAD covers it)
                d[2 * s4] = dd[0];
                d[2 * s4 + 1] = dd[1];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 * s4 + c] = vv[c];
            }
        };

        // Score one unit from registers (the first NR of them hold postings).  U_OK: done.  U_DENSE: the unit goes to tier 2
        // (nothing emitted).  U_FULL: the lazy list cannot take this unit's candidates -- nothing of the unit stays in the
        // list, the caller shrinks the list OUTSIDE the pipelined loop and comes back to this unit.  (The selection is a
        // function call: inside the loop it would have to keep every register set in flight alive across the call, which
        // cost the loop ~35 VGPRs and with them the third set.)
        auto process = [&](auto nrc, int ubase, unsigned (&d)[W_RP], float (&v)[W_R]) __attribute__((always_inline)) -> int {  // d = packed unit-local ids, ubase = the unit's first doc
            constexpr int NR = decltype(nrc)::value;
            const unsigned count0 = tk.count;
            if (count0 + (unsigned)W_DUPCAP + 1u > (unsigned)WaveShared2::LCAP) return U_FULL;  // uniform, rare: room for this unit's multi-term docs
            // ---- pass 1: doc bits.  Every slot ORs its bit into the bitmap word of its id; `old & bit` != 0 means an earlier
            //      posting (another term's) matched the same doc.  5 VALU + 1 LDS instruction per slot: the byte address and
            //      the bit stay in registers across the LDS round trip (the loop has them to spare now that the list
            //      selection left it) -- recomputing them afterwards cost ~3.5 VALU per slot and 4 % of the batch ----
            unsigned adr[NR], t[NR];  // t[r] != 0: slot r found its doc's bit already set
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const unsigned w = d[r >> 1];  // id = half r & 1 of the word
                adr[r] = (w >> ((r & 1) ? 19 : 3)) & W_BM_ADR_MASK;  // byte offset of word id >> 5
                unsigned one;  // min(value bits, 1): 0 for a value of exactly +0 (v_min_u32; the compiler's own form is cmp + cndmask)
                asm("v_min_u32 %0, 1, %1" : "=v"(one) : "v"(__float_as_uint(v[r])));
                const unsigned bit = one << (((r & 1) ? w >> 16 : w) & 31u);
                t[r] = atomicOr(reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]), bit) & bit;
            }
            unsigned acc = 0;
#pragma unroll
            for (int r = 0; r < NR; ++r) acc |= t[r];
            bool dense = false;
            const bool anydup = __ballot(acc != 0u) != 0ull;
            STAMP(2);  // wait for the unit's postings + pass 1
            CNT(0);
            // ---- the bitmap words go back to zero (measured: clearing the whole 8 KB bitmap with 8 wide stores per lane
            //      instead, which needs no addresses, costs the batch 14 % -- LDS write bandwidth) ----
#pragma unroll
            for (int r = 0; r < NR; ++r) *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(S.bm) + adr[r]) = 0u;
            STAMP(4);  // restore
            if (anydup && !(dbg & 1)) {  // uniform: some doc of this unit is matched by several terms (~3 units in 4 on C3)
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
                unsigned n_res = 0;
                unsigned long long m = __ballot(fm != 0u);
                while (m != 0ull) {  // uniform loop: about two docs per such unit on C3
                    const int src = __ffsll((long long)m) - 1;
                    const unsigned fms = (unsigned)__builtin_amdgcn_readlane((int)fm, src);
                    const int rs = __ffs((int)fms) - 1;                 // uniform: lane src's lowest flagged slot
                    const int dd = lane_reg<NR>(d, rs, src);            // its doc, wave-uniform
                    CNT(2);
                    if (dbg & 512) {  // timing experiment: locate the docs only
                        sink += dd;
                        fm = (lane == src) ? (fm & (fm - 1u)) : fm;
                        m = __ballot(fm != 0u);
                        continue;
                    }
                    float myv = 0.0f;
#pragma unroll
                    for (int r2 = 0; r2 < NR; ++r2) {
                        const bool hit = (int)slot_id(d, r2) == dd;
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
                    if (sum > 0.0f && b >= tk.tau && !(dbg & 1024)) {  // uniform; room for W_DUPCAP entries was made above
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
            if (dense) return U_DENSE;
            if (dbg & 2) return U_OK;  // timing experiment: no candidate screening (results are wrong)
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
                        const bool cand = pass && c > 0.0f && b >= tk.tau;
                        const unsigned long long m2 = __ballot(cand);
                        const unsigned n2 = (unsigned)__popcll(m2);
                        if (tk.count + n2 > (unsigned)WaveShared2::LCAP) {  // uniform: no room -- the unit is redone after a selection
                            tk.count = count0;
                            return U_FULL;
                        }
                        if (cand) {
                            const unsigned pz = tk.count + lane_rank(m2);
                            S.lbits[pz] = b;
                            S.ldoc[pz] = ubase + (int)slot_id(d, r);
                        }
                        tk.count += n2;
#ifdef SRX_STAMP
                        st_cnt[5] += n2;
#endif
                    }
                }
            }
            STAMP(6);  // candidates (appends)
            return U_OK;
        };

        auto flag_tier2 = [&](int su) __attribute__((always_inline)) {
            if (lane == 0) atomicOr(&my_ovf[su >> 5], 1u << (su & 31));
            flagged = true;
        };

        // ---- software pipeline over units: SRX_W_DEPTH register sets rotate; while unit u is scored from registers, the
        //      loads of units u+1 .. u+DEPTH-1 are in flight (a wave has no other way to keep memory requests outstanding:
        //      with one unit ahead the data arrives long before the unit before it has been scored, and nothing is in
        //      flight for the rest of that time) ----
        // score unit su held in (d, v); steps = the load steps that hold postings (0: nothing to do, W_R / 4 + 1: some term's
        // run does not fit the registers -> tier 2).  Wave-uniform, decided when the loads were issued: no per-lane run
        // length has to stay in a register while the unit is in flight.
        // returns true when the list was full: the unit has NOT been scored
        auto score = [&](int su, int steps, unsigned (&d)[W_RP], float (&v)[W_R]) __attribute__((always_inline)) -> bool {
            if (dbg & 4) {  // timing experiment: loads only (results are wrong)
#pragma unroll
                for (int r = 0; r < W_R; ++r) sink += (int)(d[r >> 1] ^ __float_as_uint(v[r]));
            } else if (steps > W_R / 4) {
                flag_tier2(su);
            } else if (steps > 0) {
                int rc;
                const int ubase = (su * tpu) << ix.tile_log2;
                if (W_R > 12 && steps == 4)
                    rc = process(IntC<(W_R > 12 ? 16 : 4)>{}, ubase, d, v);
                else if (W_R > 8 && steps == 3)
                    rc = process(IntC<(W_R > 8 ? 12 : 4)>{}, ubase, d, v);
                else if (steps == 2)
                    rc = process(IntC<8>{}, ubase, d, v);
                else
                    rc = process(IntC<4>{}, ubase, d, v);
                if (rc == U_DENSE) flag_tier2(su);
                return rc == U_FULL;
            }
            return false;
        };
        // issue unit su's loads into (d, v); returns its step count (see score; a run that does not fit loads nothing)
        int su_issue = su_lo;
        auto fetch = [&](unsigned (&d)[W_RP], float (&v)[W_R]) __attribute__((always_inline)) -> int {
            const int len = (has_term && su_issue < su_hi) ? (bq[1] - bq[0]) >> 2 : 0;  // blocks
            int steps = 0;  // uniform
#pragma unroll
            for (int s4 = 0; s4 <= W_R / 4; ++s4) steps += (__ballot(len > (s4 << LPT_LOG2)) != 0ull) ? 1 : 0;
            STAMP(0);  // loop overhead / previous tail
            issue(bq[0] >> 2, steps > W_R / 4 ? 0 : len, d, v);
            STAMP(1);  // issue
#pragma unroll
            for (int i = 0; i < NBQ - 1; ++i) bq[i] = bq[i + 1];
            ++su_issue;
            bq[NBQ - 1] = bound(su_issue + NBQ - 1);  // needed NBQ - 2 fetches from now (clamped to the row end)
            return steps;
        };
        // DEPTH register sets rotate: set j holds unit su + j when the round starts; before a unit is scored the loads of the
        // unit DEPTH - 1 ahead of it are issued into the set that was scored last (a fetch past the end issues sentinel loads only).
        // The pipeline is (re)started at unit su_start: once per item, and again after every selection of the lazy list --
        // the units in flight at that moment are fetched a second time (L2 hits), a few times per query while the threshold
        // warms up and hardly ever after.
        constexpr int DEPTH = SRX_W_DEPTH;
        int su_start = su_lo;
        STAMP(8);  // prologue: item decode, bitmap clear, term metadata, initial threshold
        for (;;) {
            su_issue = su_start;
            const bool first_start = SCALAR_PROLOGUE && su_start == su_lo;  // uniform: bq[0], bq[1] came with the scalar prologue
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int bi = bound(su_start + i);
                bq[i] = (i < 2 && first_start) ? bq[i] : bi;
            }
            unsigned dS[DEPTH][W_RP];
            float vS[DEPTH][W_R];
            int st[DEPTH];  // uniform: load steps of the unit in set j (see score)
#pragma unroll
            for (int j = 0; j < DEPTH - 1; ++j) st[j] = fetch(dS[j], vS[j]);
            st[DEPTH - 1] = 0;
            int su_full = -1;  // uniform: the unit that found the list full
            for (int su = su_start; su < su_hi && su_full < 0; su += DEPTH) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    if ((j == 0 || su + j < su_hi) && su_full < 0) {  // uniform
                        st[(j + DEPTH - 1) % DEPTH] = fetch(dS[(j + DEPTH - 1) % DEPTH], vS[(j + DEPTH - 1) % DEPTH]);
                        if (score(su + j, st[j], dS[j], vS[j])) su_full = su + j;
                    }
                }
            }
            if (su_full < 0) break;
            STAMP(0);
            CNT(4);
            // ---- no register set is live here: shrink the list to its k best (tau rises) and redo unit su_full.  A unit that
            //      does not fit next to a list of k entries either goes to tier 2 ----
            if (tk.count > (unsigned)k) {
                tk.tau = uniu(wave_list_select(S, tk.count, k));
                tk.count = (unsigned)k;
                su_start = su_full;
            } else {
                flag_tier2(su_full);
                su_start = su_full + 1;
            }
            STAMP(9);  // list selection between restarts
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
    if ((dbg & (4 | 512)) && sink == 0x7F123457) a.cand_count[list] = sink;  // keeps the loads of the timing experiment alive
    unsigned count = tk.count;
    if (dbg & 32) count = 0;  // timing experiment: no final selection / ranking
    if (count > (unsigned)k) {
        wave_list_select(S, count, k);
        count = (unsigned)k;
    }
    STAMP(7);  // epilogue (final select)
#ifdef SRX_STAMP
#define STAMP_FLUSH()                                                                              \
    do {                                                                                           \
        STAMP(10); /* rank + row / list write */                                                   \
        if (lane == 0) {                                                                           \
            for (int i = 0; i < 12; ++i) atomicAdd(&g_stamp[i], st_acc[i]);                        \
            atomicAdd(&g_stamp[12], 1ull);                                                         \
            for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp[13 + i], (unsigned long long)st_cnt[i]); \
        }                                                                                          \
    } while (0)
#else
#define STAMP_FLUSH() \
    do {              \
    } while (0)
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
        STAMP_FLUSH();
        return;
    }
    // A split's list is read by another wave of this grid (below), possibly on another XCD with an L2 of its own: its
    // entries are written and read with agent-scope (memory-coherent) accesses.  NO agent-scope fence: on this chip that is
    // a write-back / invalidate of the XCD's whole L2 -- 2 352 of them per C3 batch made the kernel 70 us slower.
    const int64_t o = list * k;
    for (unsigned i = lane; i < count; i += 64) {
        __hip_atomic_store(&a.cand_doc[o + i], S.ldoc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned *>(&a.cand_score[o + i]), S.lbits[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
        __hip_atomic_store(&a.cand_count[list], (int)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flagged) a.work[1 + atomicAdd(&a.work[0], 1)] = (int)blockIdx.x;
    }
    if (nsq > 1 && a.done != nullptr && a.out_doc != nullptr) {
        // A split query: the split that finishes LAST merges the query's lists here -- while the other waves of the grid
        // are still scoring -- unless some split left units to tier 2 (then the merge kernel does it after tier 2).
        // done[q] counts arrivals in its low half and flagged splits in its high half.  Before the counter moves, this
        // wave's stores (write-through, straight to memory) must be acknowledged: an explicit s_waitcnt -- a workgroup-scope
        // release fence compiles to NOTHING here (a workgroup of one wave is its own scope), which let the last split read
        // lists that were still in flight (caught by bench.py's multi-stream parity check on C2).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        unsigned old = 0;
        if (lane == 0) old = atomicAdd(&a.done[q - a.n_whole], 1u + (flagged ? 0x10000u : 0u));
        old = uniu(old);
        if ((old & 0xFFFFu) + 1u == (unsigned)nsq && (old >> 16) == 0u && !flagged) {
            const int64_t l0 = (int64_t)q * a.lists_per_q;  // the query's tier-1 lists: l0 .. l0 + nsq - 1
            for (int s2 = 0; s2 < nsq; ++s2) {              // uniform loop; my own list is already in LDS
                if (s2 == split) continue;
                const unsigned c2 = (unsigned)min(max(__hip_atomic_load(&a.cand_count[l0 + s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0), k);
                if (count + c2 > (unsigned)WaveShared2::LCAP) {  // k + k <= LCAP (k <= 112): after a selection there is room
                    wave_list_select(S, count, k);
                    count = (unsigned)k;
                }
                for (unsigned i = lane; i < c2; i += 64) {
                    S.ldoc[count + i] = __hip_atomic_load(&a.cand_doc[(l0 + s2) * k + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S.lbits[count + i] = __hip_atomic_load(reinterpret_cast<const unsigned *>(&a.cand_score[(l0 + s2) * k + i]), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                count += c2;
                wsync();
            }
            if (count > (unsigned)k) {
                wave_list_select(S, count, k);
                count = (unsigned)k;
            }
            wave_rank_emit(S, reinterpret_cast<unsigned long long *>(S.bm), count, k, a.doc_base, a.out_doc + (int64_t)q * a.out_row_stride,
                           a.out_score + (int64_t)q * a.out_row_stride);
            if (lane == 0) {
                a.out_count[(int64_t)q * a.out_cnt_stride] = (int)count;
                a.cand_count[l0] = -1;  // final: the merge kernel skips the query
            }
        }
    }
    STAMP_FLUSH();
}

}  // namespace

int srx_launch_wave_kernel(const srx_wave_launch &a, int val_type, int64_t blocks, hipStream_t stream) {
    if (blocks <= 0) return SRX_OK;
    const bool ablate = (a.dbg & (1 | 2 | 4 | 32 | 512 | 1024)) != 0;  // timing experiments: the DBG instance
    if (val_type == SRX_VAL_F32) {
        if (ablate)
            hipLaunchKernelGGL((srx_wave_kernel<float, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL((srx_wave_kernel<float, false>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
    } else {
        if (ablate)
            hipLaunchKernelGGL((srx_wave_kernel<__half, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL((srx_wave_kernel<__half, false>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return SRX_OK;
}

#ifdef SRX_STAMP
extern "C" __attribute__((visibility("default"))) int srx_debug_read_stamps(unsigned long long *h_out16) {
    HIP_TRY(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 32));
    unsigned long long z[32] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)));
    return SRX_OK;
}
#endif

// Covisitation candidate lookup. C-ABI and reference citations: include/otto_cand.h.
//
// One 256-thread workgroup per session:
//   A  load the session; per event: is it the last occurrence of its aid (-> U, most recent first), the first
//      click/cart (-> CC) or cart/order (-> CO) occurrence; ranks of the sorted lists by counting smaller aids
//   B  flattened (term, source aid) list: list lengths from the matrices, block exclusive scan = position of every
//      list in the concatenation the reference builds
//   C  hash the list entries into an LDS table: one 64-bit word  aid << 32 | count  (ds_add) and a 32-bit first
//      position (ds_min). Sessions whose concatenation exceeds the table are processed in hash partitions whose
//      results are merged exactly (partitions hold disjoint aids)
//   D  Counter.most_common(n_common): composite key count << 50 | ~first_pos << 26 | aid, two 64-wide block
//      selections (topk.h), then drop the session's own aids and compact
#include "common.h"
#include "topk.h"
#include "scan.h"
#include "../../include/otto_cand.h"

#include <string.h>

namespace otto {

constexpr uint64_t CD_EMPTY = ~0ull;
constexpr int CD_SMALL_MAXL = 32;
constexpr int CD_STACK = 96;                          // partitions waiting (first level <= 64) + refinements                     // sessions up to this many events run in the small-footprint variant

struct CandArgs {
    otto_cand_params p;
    const uint32_t* aid;
    const uint8_t* type;
    const int64_t* sess_off;
    int64_t n_sess;
    int32_t* cand;
    int32_t* count;
    int32_t* n_out;
    uint32_t* err;
    const uint32_t* list;          // this launch's sessions (k_cand_classify: the short ones / the long ones)
    const uint32_t* list_n;        // their number
    uint32_t* work;                // dequeue counter: sessions cost from a few to a few hundred microseconds, a static split leaves a tail
    int32_t* self_count;           // nullable: [n_events] Counter count of every event's aid; the session's aids then leave the selection
#ifdef OTTO_PHASE_PROF
    unsigned long long* prof;      // [8] shader-clock ticks of thread 0 per phase (diagnostic build)
    int debug;                     // diagnostic build, env OTTO_CAND_DEBUG (wrong results): 1 no inserts, 2 no list loads, 4 no selection
#endif
};
#ifdef OTTO_PHASE_PROF
#define CD_PH(i) do { if (threadIdx.x == 0) { const unsigned long long _t = clock64(); ph[i] += _t - ph_t; ph_t = _t; } } while (0)
#else
#define CD_PH(i) do {} while (0)
#endif

__device__ __forceinline__ uint64_t cand_key(uint64_t count, uint32_t fp, uint32_t y) {
    return (count << 50) | ((uint64_t)(0xFFFFFFu - fp) << 26) | (uint64_t)y;
}

// Work-list dequeue of the per-session kernels (thread 0 of a workgroup): sessions are reserved CD_DQ at a time -- same-address
// atomics retire at ~90 M/s, one per session would bound 1.6 M short sessions at 17 ms -- and the next chunk is requested when the
// current one is opened, so its round trip is never waited for.
constexpr uint32_t CD_DQ = 8;
struct WorkPool {
    uint32_t pool, pool_end, pool_next;
    __device__ __forceinline__ void init(uint32_t* counter) {
        pool = atomicAdd(counter, 2 * CD_DQ);
        pool_end = pool + CD_DQ;
        pool_next = pool + CD_DQ;
    }
    __device__ __forceinline__ uint32_t take(uint32_t* counter) {
        const uint32_t r = pool++;
        if (pool == pool_end) {
            pool = pool_next;
            pool_end = pool + CD_DQ;
            pool_next = atomicAdd(counter, CD_DQ);
        }
        return r;
    }
};

// Sessions by length class: hdr[0] / hdr[1] = number of short (<= short_max events) / long sessions, their ids in lists[0] / lists[1]
// (n_sess entries each). One returning atomic per wave and class.
__global__ __launch_bounds__(256) void k_cand_classify(const int64_t* sess_off, int64_t n_sess, int short_max, uint32_t* hdr, uint32_t* list_short, uint32_t* list_long) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned lane = lane_id();
    const bool in = s < n_sess;
    const int64_t n = in ? sess_off[s + 1] - sess_off[s] : 0;
    const bool is_long = in && n > short_max;
    const uint64_t ml = __ballot(is_long), ms = __ballot(in && !is_long);
    uint32_t bl = 0, bs = 0;
    if (lane == 0) {
        if (ml) bl = atomicAdd(&hdr[1], (uint32_t)__popcll(ml));
        if (ms) bs = atomicAdd(&hdr[0], (uint32_t)__popcll(ms));
    }
    bl = (uint32_t)__builtin_amdgcn_readfirstlane((int)bl);
    bs = (uint32_t)__builtin_amdgcn_readfirstlane((int)bs);
    const uint64_t below = (1ull << lane) - 1ull;
    if (is_long) list_long[bl + (uint32_t)__popcll(ml & below)] = (uint32_t)s;
    else if (in) list_short[bs + (uint32_t)__popcll(ms & below)] = (uint32_t)s;
}

// Two instantiations share the code: <32, 10, 128> for short sessions (16 KB of LDS: ten workgroups per CU instead of
// two -- most sessions are short and the per-session phases are barrier / latency bound) and <500, 12, 256> for the rest.
template <int CD_MAXL, int CD_LOG2T, int CD_THREADS>
// (launch bounds: 5 waves per SIMD for the short variant = ten 2-wave workgroups per CU; 4 for the long one = TWO 8-wave workgroups
// per CU, which its 78 KB of LDS allow -- without the bound the compiler took 168 registers and one workgroup fit)
__global__ __launch_bounds__(CD_THREADS, CD_THREADS == 128 ? 5 : 4) void k_cand(CandArgs a) {
    constexpr int CD_NW = CD_THREADS / 64;
    constexpr int CD_T = 1 << CD_LOG2T;
    constexpr int CD_CAP = CD_T / 4 * 3;                 // list entries per hash partition (load <= 3/4)
    constexpr int CD_MAXQ = OTTO_CAND_MAX_TERMS * CD_MAXL;
    static_assert(CD_THREADS >= OTTO_CAND_MAX_COMMON && CD_NW >= 2, "one carried entry per thread, two compaction waves");
    __shared__ __attribute__((aligned(16))) uint32_t s_aid[CD_MAXL];
    __shared__ __attribute__((aligned(16))) uint8_t s_ty[CD_MAXL];
    __shared__ __attribute__((aligned(16))) uint8_t s_flag[CD_MAXL];   // bit0 last occurrence, bit1 first click/cart, bit2 first cart/order, bit3 first click
    __shared__ uint16_t s_src[4][CD_MAXL];            // U, CC, CO, C -- as EVENT positions; source LAST is the last event itself
                                                      // (the aid is s_aid[position]; 16-bit entries keep two of the long-session workgroups on a CU)
    __shared__ uint32_t s_nsrc[5];
    __shared__ uint32_t s_base[CD_MAXQ];               // position of list q in the concatenation | len << 24
    __shared__ unsigned long long s_tab[CD_T];        // aid << 32 | count
    __shared__ uint32_t s_fp[CD_T];                    // first position
    __shared__ uint64_t s_sel[OTTO_CAND_MAX_COMMON];   // running most_common list (sorted)
    // one pool for three arrays that are never live together: the gather's segment table (CD_NW x 256 bytes), the radix-select
    // histogram (256 or 1024 bins) and the compacted selection (OTTO_CAND_MAX_COMMON keys). Short-session variant: 16.3 instead of
    // 17.8 KB of LDS = ten workgroups per CU instead of nine (the kernel scales with its resident waves).
    constexpr int CD_POOL = (CD_THREADS * 8 >= 4096 ? 4096 : 1024) > CD_NW * 256 ? (CD_THREADS * 8 >= 4096 ? 4096 : 1024) : CD_NW * 256;
    __shared__ __attribute__((aligned(16))) uint8_t s_pool[CD_POOL > OTTO_CAND_MAX_COMMON * 8 ? CD_POOL : OTTO_CAND_MAX_COMMON * 8];
    uint64_t* s_ex = reinterpret_cast<uint64_t*>(s_pool);
    __shared__ uint32_t s_nex, s_more, s_ovf, s_nfresh, s_sp, s_scan[CD_NW + 1], s_keep[2], s_maxlen;
    __shared__ uint32_t s_stack[CD_STACK];            // hash partitions still to do: id | level << 24
    uint8_t* s_seg = s_pool;                          // gather: segment -> list lane << 2 | segment of the list, per wave

    const int tid = threadIdx.x, wid = tid >> 6;
    const unsigned lane = lane_id();
    const int K = a.p.k, NC = a.p.n_common;
#ifdef OTTO_PHASE_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = clock64();
#endif

    __shared__ uint32_t s_next[2];
    const uint32_t n_list = *a.list_n;
    WorkPool wp;
    if (tid == 0) { wp.init(a.work); s_next[0] = wp.take(a.work); }
    __syncthreads();
    uint32_t item = s_next[0];
    for (uint32_t it = 0; item < n_list; ++it) {
        if (tid == 0) s_next[(it + 1u) & 1u] = wp.take(a.work);         // read at the end of this session
        const int64_t s = (int64_t)a.list[item];
        const int64_t lo = a.sess_off[s], hi = a.sess_off[s + 1];
        const int n = (int)(hi - lo);
        CD_PH(0);
        if (n > CD_MAXL) {
            if (tid == 0) atomicAdd(a.err, 1u);
            __syncthreads();
            item = s_next[(it + 1u) & 1u];
            continue;
        }
        // ---- A ----------------------------------------------------------------------------------------
        for (int i = tid; i < n; i += CD_THREADS) { s_aid[i] = a.aid[lo + i]; s_ty[i] = a.type[lo + i]; }
        if (tid < 3 || tid == 4) s_nsrc[tid] = 0;
        if (tid == 3) { s_nsrc[3] = n > 0 ? 1u : 0u; s_maxlen = 0; }
        if (a.self_count)
            for (int i = tid; i < n; i += CD_THREADS) a.self_count[lo + i] = 0;
        for (int i = tid; i < NC; i += CD_THREADS) s_sel[i] = 0;
        __syncthreads();
        // Flags and ranks compare every event with every other event of the session. The session is read FOUR events per LDS
        // operation (one 16-byte read of aids, one 4-byte read of types / flags) and without branches: with two workgroups per
        // CU and most waves of a workgroup waiting at the barrier, a loop of one dependent LDS read per event ran at one LDS round
        // trip per event (a quarter of the long-session kernel).
        // SUB adjacent lanes share an event and split the session between them (n = 82 on 512 threads: 4 lanes per event, a
        // quarter of the loop each; partial results are combined with DPP exchanges): the loop length, not the thread count, is
        // what these two passes wait for.
        int lgsub = 0;
        while (lgsub < 3 && (n << (lgsub + 1)) <= CD_THREADS) ++lgsub;
        const int SUB = 1 << lgsub;
        const int ev_i = tid >> lgsub, sub = tid & (SUB - 1);
        const int jlen = (((n + SUB - 1) >> lgsub) + 3) & ~3;      // events per lane of a group, a multiple of 4
        const int jb = sub * jlen, je = (jb + jlen < n) ? jb + jlen : n;
        auto group_or = [&](uint32_t v) {
            if (lgsub >= 1) v |= xor_lane32(v, 1);
            if (lgsub >= 2) v |= xor_lane32(v, 2);
            if (lgsub >= 3) v |= xor_lane32(v, 4);
            return v;
        };
        auto group_add = [&](uint32_t v) {
            if (lgsub >= 1) v += xor_lane32(v, 1);
            if (lgsub >= 2) v += xor_lane32(v, 2);
            if (lgsub >= 3) v += xor_lane32(v, 4);
            return v;
        };
        for (int i0 = 0; i0 < n; i0 += CD_THREADS >> lgsub) {      // one pass unless n > CD_THREADS / 2 (then SUB = 1)
            const int i = i0 + ev_i;
            const bool act = i < n;
            const uint32_t ai = act ? s_aid[i] : 0u;
            const uint32_t ti = act ? s_ty[i] : 0u;
            uint32_t acc = 0;                                      // bit 0 same aid later, bits 1 / 2 / 3 same aid earlier with type <= 1 / >= 1 / == 0
            if (act) {
#pragma unroll 2
                for (int j0 = jb; j0 < je; j0 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[j0]);
                    const uint32_t t4 = *reinterpret_cast<const uint32_t*>(&s_ty[j0]);
                    const uint32_t aj[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = j0 + e;
                        const uint32_t tj = (t4 >> (8 * e)) & 0xFFu;
                        const uint32_t eq = (aj[e] == ai && j < n) ? 1u : 0u;
                        const uint32_t before = eq & (j < i ? 1u : 0u);
                        acc |= (eq & (j > i ? 1u : 0u)) | ((before & (tj <= 1u ? 1u : 0u)) << 1) | ((before & (tj >= 1u ? 1u : 0u)) << 2) |
                               ((before & (tj == 0u ? 1u : 0u)) << 3);
                    }
                }
            }
            acc = group_or(acc);
            if (act && sub == 0) {
                const bool last = (acc & 1u) == 0, fcc = ti <= 1 && (acc & 2u) == 0, fco = ti >= 1 && (acc & 4u) == 0, fc = ti == 0 && (acc & 8u) == 0;
                s_flag[i] = (uint8_t)((last ? 1 : 0) | (fcc ? 2 : 0) | (fco ? 4 : 0) | (fc ? 8 : 0));
            }
        }
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += CD_THREADS >> lgsub) {      // (uniform trip count: the list sizes come from wave ballots)
            const int i = i0 + ev_i;
            const bool act = i < n;
            uint32_t fl = 0;
            uint32_t r01 = 0, r23 = 0;                             // ranks, two 16-bit counters per word: ru | rcc << 16, rco | rc << 16
            const uint32_t ai = act ? s_aid[i] : 0u;
            if (act) {
#pragma unroll 2
                for (int j0 = jb; j0 < je; j0 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[j0]);
                    const uint32_t f4 = *reinterpret_cast<const uint32_t*>(&s_flag[j0]);
                    const uint32_t aj[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = j0 + e;
                        const uint32_t fj = j < n ? (f4 >> (8 * e)) & 0xFFu : 0u;
                        const uint32_t lt = aj[e] < ai ? 1u : 0u;
                        r01 += ((j > i ? 1u : 0u) & (fj & 1u)) | ((((fj >> 1) & 1u) & lt) << 16);
                        r23 += (((fj >> 2) & 1u) & lt) | ((((fj >> 3) & 1u) & lt) << 16);
                    }
                }
            }
            r01 = group_add(r01);
            r23 = group_add(r23);
            if (act && sub == 0) {
                fl = s_flag[i];
                if (fl & 1u) s_src[0][r01 & 0xFFFFu] = (uint16_t)i;
                if (fl & 2u) s_src[1][r01 >> 16] = (uint16_t)i;
                if (fl & 4u) s_src[2][r23 & 0xFFFFu] = (uint16_t)i;
                if (fl & 8u) s_src[3][r23 >> 16] = (uint16_t)i;
            }
            // list sizes: one LDS atomic per wave and list instead of one per event on the same address
            const uint32_t c0 = (uint32_t)__popcll(__ballot(fl & 1u)), c1 = (uint32_t)__popcll(__ballot(fl & 2u));
            const uint32_t c2 = (uint32_t)__popcll(__ballot(fl & 4u)), c3 = (uint32_t)__popcll(__ballot(fl & 8u));
            if (lane == 0) {
                if (c0) atomicAdd(&s_nsrc[0], c0);
                if (c1) atomicAdd(&s_nsrc[1], c1);
                if (c2) atomicAdd(&s_nsrc[2], c2);
                if (c3) atomicAdd(&s_nsrc[4], c3);
            }
        }
        __syncthreads();
        // aid of entry j of source list `src` (OTTO_CAND_SRC_*)
        auto src_aid = [&](int src, uint32_t j) -> uint32_t {
            return s_aid[src == OTTO_CAND_SRC_LAST ? (uint32_t)(n - 1) : (uint32_t)s_src[src == OTTO_CAND_SRC_C ? 3 : src][j]];
        };
        CD_PH(1);
        // ---- B ----------------------------------------------------------------------------------------
        uint32_t tstart[OTTO_CAND_MAX_TERMS + 1];
        tstart[0] = 0;
#pragma unroll
        for (int t = 0; t < OTTO_CAND_MAX_TERMS; ++t)
            tstart[t + 1] = tstart[t] + (t < a.p.n_terms ? s_nsrc[a.p.term_source[t]] : 0u);
        const uint32_t Q = tstart[OTTO_CAND_MAX_TERMS];
        uint32_t running = 0;
        for (uint32_t q0 = 0; q0 < Q; q0 += CD_THREADS) {
            const uint32_t q = q0 + tid;
            uint32_t len = 0;
            if (q < Q) {
                int t = 0;
#pragma unroll
                for (int u = 1; u < OTTO_CAND_MAX_TERMS; ++u) t += (q >= tstart[u]) ? 1 : 0;
                const uint32_t x = src_aid(a.p.term_source[t], q - tstart[t]);
                const int m = a.p.term_matrix[t];
                const int Km = a.p.mat_k[m] > 0 ? a.p.mat_k[m] : K;
                const int32_t ln = x < a.p.n_aids ? a.p.d_mat_n[m][x] : 0;
                len = ln < 0 ? 0u : (uint32_t)(ln > Km ? Km : ln);
            }
            uint32_t tot;
            const uint32_t off = block_excl_scan<uint32_t, CD_THREADS>(len, s_scan, &tot);
            if (q < Q) s_base[q] = (running + off) | (len << 24);
            if (len > 32u) atomicMax(&s_maxlen, len);            // neighbour lists (45 entries): a second sweep of the gather
            running += tot;
        }
        const uint32_t TOT = running;
        CD_PH(2);
        // table size follows the concatenation: most sessions are short, and clearing / scanning 4096 slots for a few
        // hundred entries would dominate them
        const int lt0 = TOT <= 192u ? 8 : (TOT <= 768u ? 10 : CD_LOG2T);
        const int lt1 = lt0 < CD_LOG2T ? lt0 : CD_LOG2T;
        const int lt = (1 << lt1) < CD_THREADS ? CD_LOG2T : lt1;       // at least one table slot per thread
        const int Teff = 1 << lt;
        const int mpl = Teff / CD_THREADS;                    // table slots per thread: 1, 4 or 16
        // ---- C + D per hash partition. A partition may hold CD_CAP DISTINCT aids; the lists of a session overlap heavily (the
        //      same source aid in several matrices, neighbouring aids with common partners), so the first level is sized for
        //      half the entries, and a partition that still collects more than CD_CAP distinct aids is split in two (one more
        //      hash bit) and only ITS entries are gathered again. Partitions hold disjoint aids: each one merges its
        //      most_common(n_common) into the running list s_sel, in any order.
        int lg0 = 0;
        // Recipes that read three or more matrices through the same source list (the click recipes) always start at half; the
        // others only where that can save two passes or more (up to two tables' worth of entries: two partitions as before).
        int same_src = 0;
        for (int t = 0; t < a.p.n_terms; ++t) {
            int c = 0;
            for (int u = 0; u < a.p.n_terms; ++u) c += a.p.term_source[u] == a.p.term_source[t] ? 1 : 0;
            same_src = c > same_src ? c : same_src;
        }
        const uint32_t opt_from = same_src >= 3 ? (uint32_t)CD_CAP : 2u * (uint32_t)CD_CAP;
        while (((uint32_t)CD_CAP << lg0) < (TOT <= opt_from ? TOT : TOT / 2u) && lg0 < 6) ++lg0;
        if (tid == 0) {
            s_sp = 1u << lg0;
            for (uint32_t pp = 0; pp < (1u << lg0); ++pp) s_stack[pp] = pp | ((uint32_t)lg0 << 24);
        }
        for (int i = tid; i < NC; i += CD_THREADS) s_sel[i] = 0;
        bool have_sel = false;                                // s_sel holds the merged result of at least one partition
        __syncthreads();
        for (;;) {
            const uint32_t sp = s_sp;
            if (sp == 0) break;
            const uint32_t ent = s_stack[sp - 1];
            const uint32_t part = ent & 0xFFFFFFu;
            const int lgR = (int)(ent >> 24);
            const uint32_t R = 1u << lgR;
            __syncthreads();
            if (tid == 0) { s_sp = sp - 1; s_ovf = 0; s_nfresh = 0; }
            for (int i = tid; i < Teff; i += CD_THREADS) { s_tab[i] = CD_EMPTY; s_fp[i] = 0xFFFFFFFFu; }
            __syncthreads();
            CD_PH(3);
            // Gather + insert. A wave takes 64 lists at a time, ONE LIST PER LANE for the set-up (position | length, source aid,
            // row address: done once per list instead of once per list and 32-lane pass), cuts them into segments of 8 entries
            // (wave scan of the segment counts, segment -> list lane in a byte table private to the wave) and deals the segments
            // to its eight 8-lane groups: a wave-instruction carries up to 64 entries whatever the list lengths are (k = 15: 94 %
            // of the lanes; one list per 32-lane half filled 47 %). The entry's lane fetches the list's words from the list's
            // lane with three ds_bpermute. GU steps are in flight per lane; lists longer than 32 (neighbour lists) take a
            // second sweep. (Round-3 measurements: a fifth of the long-session kernel was the loop skeleton alone -- set-up
            // repeated per pass -- and a third the inserts, both per wave-instruction, not per entry.)
            constexpr int GU = 4;
            const uint32_t g8 = lane >> 3, gl = lane & 7u;
            uint8_t* seg = s_seg + wid * 256;
            const uint32_t sweeps = s_maxlen > 32u ? (s_maxlen + 31u) / 32u : 1u;      // lists longer than 32 (neighbour lists): more sweeps
            for (uint32_t qb = 0; qb < Q; qb += (uint32_t)CD_NW * 64u) {            // lists dealt round-robin to the waves: short sessions keep every wave busy
                if (s_ovf) break;                                 // the partition is being split: its table is not used
                const uint32_t q = qb + lane * (uint32_t)CD_NW + (uint32_t)wid;
                uint32_t b = 0;
                uint64_t row = 0;
                if (q < Q) {
                    b = s_base[q];
#pragma unroll
                    for (int t = 0; t < OTTO_CAND_MAX_TERMS; ++t) {
                        if (t < a.p.n_terms && q >= tstart[t] && q < tstart[t + 1]) {          // t uniform: the term's words are scalars
                            const int m = a.p.term_matrix[t];
                            const uint32_t x = src_aid(a.p.term_source[t], q - tstart[t]);
                            row = (uint64_t)(uintptr_t)a.p.d_mat_y[m] + ((uint64_t)x * (uint32_t)(a.p.mat_k[m] > 0 ? a.p.mat_k[m] : K)) * 4ull;
                        }
                    }
                }
                const int rlo = (int)(uint32_t)row, rhi = (int)(uint32_t)(row >> 32);
                for (uint32_t sw = 0; sw < sweeps; ++sw) {
                    const uint32_t len_all = b >> 24;
                    const uint32_t len = len_all > 32u * sw ? (len_all - 32u * sw > 32u ? 32u : len_all - 32u * sw) : 0u;   // entries of this sweep
                    const uint32_t segs = (len + 7u) >> 3;
                    const uint32_t incl = wave_incl_scan(segs), excl = incl - segs;
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (k < segs) seg[excl + k] = (uint8_t)((lane << 2) | k);
                    wave_lds_sync();
                    const int blen = (int)((b & 0xFFFFFFu) + 32u * sw) | (int)(len << 24);       // position of the sweep's first entry | entries
                    const int nstep = (int)((total + 7u) >> 3);
                    for (int t0 = 0; t0 < nstep; t0 += GU) {
                        uint32_t yv[GU], pv[GU];
                        bool ok[GU];
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            const uint32_t sq = (uint32_t)(t0 + u) * 8u + g8;
                            const uint32_t r = (uint32_t)seg[sq & 255u];
                            const int src = (int)(r & 0xFCu);
                            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, rlo);
                            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, rhi);
                            const uint32_t bl = (uint32_t)__builtin_amdgcn_ds_bpermute(src, blen);
                            const uint32_t off = ((r & 3u) << 3) | gl;                           // entry of the sweep
                            ok[u] = sq < total && off < (bl >> 24);
                            pv[u] = (bl & 0xFFFFFFu) + off;
                            yv[u] = 0;
                            if (ok[u]) {
#ifdef OTTO_PHASE_PROF
                                if (a.debug & 2) yv[u] = (lo * 2654435761u + off * 40503u) % a.p.n_aids;
                                else
#endif
                                yv[u] = (uint32_t)*reinterpret_cast<const __attribute__((address_space(1))) int32_t*>(
                                    (uintptr_t)((((uint64_t)hi << 32) | lo) + (uint64_t)(off + 32u * sw) * 4ull));
                            }
                        }
                        uint32_t nfr = 0;                             // keys this wave entered with this batch
#pragma unroll
                        for (int u = 0; u < GU; ++u) {
                            bool fr = false;
#ifdef OTTO_PHASE_PROF
                            if ((a.debug & 1) && ok[u]) { if (yv[u] == 0xDEADBEEFu) s_ovf = 1; ok[u] = false; }
#endif
                            if (ok[u]) {
                                const uint32_t y = yv[u];
                                const uint32_t h = y * 0x9E3779B1u;
                                if (lgR == 0 || ((h >> (32 - lt - lgR)) & (R - 1u)) == part) {
                                    uint32_t slot = h >> (32 - lt);
                                    // double hashing (odd step from a second hash: every slot is visited): linear probing builds
                                    // clusters at the 3/4 load a partition may reach, and a wave waits for its longest chain
                                    const uint32_t step = ((y * 0x85EBCA6Bu) >> (32 - lt)) | 1u;
                                    bool placed = false;
                                    for (int probe = 0; probe < Teff; ++probe) {
                                        // CAS first: a new aid goes in together with its first count. (Measured and dropped in round 3: a
                                        // plain read before the atomics, so that a hit costs one atomic add instead of a failed CAS + add
                                        // + min on its slot: click recipe 115 -> 127 ms.)
                                        const unsigned long long old = atomicCAS(&s_tab[slot], (unsigned long long)CD_EMPTY, ((unsigned long long)y << 32) | 1ull);
                                        const bool fresh = old == CD_EMPTY;
                                        if (fresh || (uint32_t)(old >> 32) == y) {
                                            if (!fresh) atomicAdd(&s_tab[slot], 1ull);
                                            fr = fresh;
                                            atomicMin(&s_fp[slot], pv[u]);
                                            placed = true;
                                            break;
                                        }
                                        slot = (slot + step) & (uint32_t)(Teff - 1);
                                    }
                                    if (!placed) s_ovf = 1;
                                }
                            }
                            nfr += (uint32_t)__popcll(__ballot(fr));
                        }
                        // distinct aids of the partition so far: ONE LDS atomic per wave and batch instead of one per new key on the
                        // same address
                        if (TOT > (uint32_t)CD_CAP && nfr != 0 && lane == 0 && atomicAdd(&s_nfresh, nfr) + nfr > (uint32_t)CD_CAP) s_ovf = 1;
                    }
                    wave_lds_sync();                                  // the segment table is rewritten by the next sweep / batch
                }
            }
            __syncthreads();
            CD_PH(4);
            if (s_ovf) {                                      // split this partition: one more hash bit, two children
                if (tid == 0) {
                    if (lgR + lt >= 32 || sp + 1 > (uint32_t)CD_STACK) atomicAdd(a.err, 1u << 16);      // cannot happen below 2^20 equal hash prefixes
                    else {
                        s_stack[sp - 1] = (part << 1) | ((uint32_t)(lgR + 1) << 24);
                        s_stack[sp] = ((part << 1) | 1u) | ((uint32_t)(lgR + 1) << 24);
                        s_sp = sp + 1;
                    }
                }
                __syncthreads();
                continue;
            }
            if (a.self_count) {
                // the session's own aids: report their counts, then take them out of the selection (count 0 = tombstone: the
                // slot stays occupied for the probes of the other aids)
                auto find = [&](uint32_t x) -> int {
                    const uint32_t h = x * 0x9E3779B1u;
                    if (lgR != 0 && ((h >> (32 - lt - lgR)) & (R - 1u)) != part) return -1;
                    uint32_t slot = h >> (32 - lt);
                    const uint32_t step = ((x * 0x85EBCA6Bu) >> (32 - lt)) | 1u;
                    for (int probe = 0; probe < Teff; ++probe) {
                        const unsigned long long v = s_tab[slot];
                        if (v == CD_EMPTY) return -1;
                        if ((uint32_t)(v >> 32) == x) return (int)slot;
                        slot = (slot + step) & (uint32_t)(Teff - 1);
                    }
                    return -1;
                };
                for (int i = tid; i < n; i += CD_THREADS) {
                    const int sl = find(s_aid[i]);
                    if (sl >= 0) a.self_count[lo + i] = (int32_t)(s_tab[sl] & 0xFFFFFFFFull);
                }
                __syncthreads();
                for (uint32_t j = tid; j < s_nsrc[0]; j += CD_THREADS) {
                    const int sl = find(s_aid[s_src[0][j]]);
                    if (sl >= 0) s_tab[sl] &= 0xFFFFFFFF00000000ull;
                }
                __syncthreads();
            }
            // ---- most_common(NC) of this partition's table + the list carried over from the earlier partitions ----
            // Order: count descending, first position ascending (first positions are distinct: one aid per position of the
            // concatenation), i.e. the ORDER KEY  count << TB | (2^TB - 1 - first position)  with TB = bits of TOT. The
            // NC-th largest order key is found exactly by a most-significant-digit radix select (8- or 10-bit digits, 2 passes
            // for count < 2^8 and TOT < 2^12): every thread keeps its <= 9 candidates in registers, a pass is one LDS histogram
            // of the candidates that match the digits fixed so far + one 256-bin scan by wave 0. Then the selected keys (exactly
            // NC of them, or every candidate) are compacted and ranked by counting. All waves work in every step; the former
            // path (lane bests -> 64-lane sort -> one-by-one pushes into a sorted list held by wave 0, two rounds of 64) cost
            // several thousand instructions on a single wave per partition.
            {
                constexpr int MAXC = CD_T / CD_THREADS + 1;          // table slots per thread + one carried entry
                const int TB = 32 - __clz((int)(TOT | 1u));             // first positions < TOT <= 2^TB
                const uint64_t fpb = 0xFFFFFFull - ((1ull << TB) - 1ull);
                uint64_t ck[MAXC], ok[MAXC];
#pragma unroll
                for (int q = 0; q < MAXC; ++q) {
                    uint64_t c = 0;
                    if (q < mpl) {
                        const int i = q * CD_THREADS + tid;
                        const unsigned long long v = s_tab[i];
                        if (v != CD_EMPTY && (v & 0xFFFFFFFFull) != 0) c = cand_key(v & 0xFFFFFFFFull, s_fp[i], (uint32_t)(v >> 32));
                    } else if (q == mpl) {
                        if (have_sel && tid < NC) c = s_sel[tid];
                    }
                    ck[q] = c;
                    ok[q] = c ? (((c >> 50) << TB) | (((c >> 26) & 0xFFFFFFull) - fpb)) : 0ull;
                }
                // digits of DB bits: 10 where the histogram fits (512 threads: a 4 KB pool), 8 otherwise -- two passes instead of
                // three for the long sessions (count < 2^8, TOT < 2^12)
                constexpr int DB = CD_THREADS * 8 >= 4096 ? 10 : 8;
                constexpr int NBIN = 1 << DB, BPL = NBIN / 64;         // bins, bins per lane of the scanning wave
                uint32_t* hist = reinterpret_cast<uint32_t*>(s_pool); // NBIN bins
                for (int i = tid; i < NBIN; i += CD_THREADS) hist[i] = 0;
                if (tid == 0) s_nex = 0;
                // a key occurs at most once per list: count <= Q
                const int W = TB + (32 - __clz((int)(Q | 1u)));
                const int P = (W + DB - 1) / DB;
                uint64_t prefix = 0;                                  // the digits fixed so far (order key >> (shift + DB))
                uint32_t need = (uint32_t)NC;
                bool all = false;                                     // fewer candidates than NC: every candidate is selected
                __syncthreads();
                for (int pass = 0; pass < P; ++pass) {
                    const int shift = DB * (P - 1 - pass);
#pragma unroll
                    for (int q = 0; q < MAXC; ++q) {
                        const bool in = ok[q] != 0 && (ok[q] >> (shift + DB)) == prefix;
                        const uint32_t d = (uint32_t)(ok[q] >> shift) & (uint32_t)(NBIN - 1);
                        // most candidates of a wave share the digit (small counts): one add for all of them
                        const uint64_t m_in = __ballot(in);
                        if (m_in == 0) continue;
                        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, __builtin_ctzll(m_in));
                        const uint64_t m0 = __ballot(in && d == d0);
                        if (lane == (unsigned)__builtin_ctzll(m_in)) atomicAdd(&hist[d0], (uint32_t)__popcll(m0));
                        if (in && d != d0) atomicAdd(&hist[d], 1u);
                    }
                    __syncthreads();
                    if (wid == 0) {
                        uint32_t h[BPL];                              // bins BPL lane .. BPL lane + BPL - 1
                        uint32_t sum = 0;
#pragma unroll
                        for (int v = 0; v < BPL / 4; ++v) {
                            const uint4 h4 = reinterpret_cast<const uint4*>(hist)[lane * (BPL / 4) + v];
                            reinterpret_cast<uint4*>(hist)[lane * (BPL / 4) + v] = make_uint4(0u, 0u, 0u, 0u);   // cleared for the next pass
                            h[4 * v] = h4.x; h[4 * v + 1] = h4.y; h[4 * v + 2] = h4.z; h[4 * v + 3] = h4.w;
                            sum += h4.x + h4.y + h4.z + h4.w;
                        }
                        const uint32_t incl = wave_incl_scan(sum);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        if (pass == 0 && total < need) {
                            if (lane == 0) s_more = 0xFFFFFFFFu;       // take every candidate
                        } else {
                            uint32_t above = total - incl;             // candidates in the bins above this lane's
                            int b = -1;
                            uint32_t above_b = 0;
#pragma unroll
                            for (int v = BPL - 1; v >= 0; --v) {
                                if (b < 0 && above < need && need <= above + h[v]) { b = v; above_b = above; }
                                above += h[v];
                            }
                            if (b >= 0) { s_more = (uint32_t)BPL * lane + (uint32_t)b; s_keep[0] = need - above_b; }
                        }
                    }
                    __syncthreads();
                    const uint32_t dsel = s_more;
                    if (dsel == 0xFFFFFFFFu) { all = true; break; }
                    prefix = (prefix << DB) | dsel;
                    need = s_keep[0];
                }
                const uint64_t thr = all ? 0ull : prefix;              // the NC-th largest order key (the last pass fixes its last digit)
                uint32_t mine = 0;
#pragma unroll
                for (int q = 0; q < MAXC; ++q) mine += (ok[q] != 0 && ok[q] >= thr) ? 1u : 0u;
                uint32_t at = mine ? atomicAdd(&s_nex, mine) : 0u;    // ONE returning LDS atomic per thread (a round trip), not one per key
#pragma unroll
                for (int q = 0; q < MAXC; ++q)
                    if (ok[q] != 0 && ok[q] >= thr) s_ex[at++] = ck[q];
                __syncthreads();
                const uint32_t nsel = s_nex;                           // <= NC: order keys are distinct
                if (tid < NC) {
                    if ((uint32_t)tid < nsel) {
                        const uint64_t key = s_ex[tid];
                        uint32_t rank = 0;
                        for (uint32_t i = 0; i < nsel; ++i) rank += s_ex[i] > key ? 1u : 0u;
                        s_sel[rank] = key;
                    } else s_sel[tid] = 0;
                }
                have_sel = true;
                __syncthreads();
            }
            CD_PH(5);
        }
        __syncthreads();
        // ---- drop the session's own aids, compact, write ---------------------------------------------------
        bool keep = false;
        uint32_t y = 0, cnt = 0;
        if (tid < NC) {
            const uint64_t c = s_sel[tid];
            if (c != 0) {
                y = (uint32_t)(c & 0x3FFFFFFull);
                cnt = (uint32_t)(c >> 50);
                uint32_t own = 0;                                 // y among the session's aids (= its unique aids), four events per LDS read
#pragma unroll 2
                for (int j0 = 0; j0 < n; j0 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[j0]);
                    own |= (a4.x == y ? 1u : 0u) | ((a4.y == y && j0 + 1 < n) ? 1u : 0u) | ((a4.z == y && j0 + 2 < n) ? 1u : 0u) | ((a4.w == y && j0 + 3 < n) ? 1u : 0u);
                }
                keep = own == 0;
            }
        }
        const uint64_t bal = __ballot(keep);
        if (wid < 2 && lane == 0) s_keep[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (wid < 2) {
            const uint32_t pos = (wid == 1 ? s_keep[0] : 0u) + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (keep) {
                a.cand[s * NC + pos] = (int32_t)y;
                a.count[s * NC + pos] = (int32_t)cnt;
            }
        }
        const uint32_t total = s_keep[0] + (NC > 64 ? s_keep[1] : 0u);
        for (int i = (int)total + tid; i < NC; i += CD_THREADS) { a.cand[s * NC + i] = -1; a.count[s * NC + i] = 0; }
        if (tid == 0) a.n_out[s] = (int32_t)total;
        __syncthreads();
        item = s_next[(it + 1u) & 1u];
        CD_PH(6);
    }
#ifdef OTTO_PHASE_PROF
    if (threadIdx.x == 0 && a.prof)
        for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[i], ph[i]);
#endif
}

// final predictions: unique session aids (most recent first) + candidates + global most frequent aids
__global__ __launch_bounds__(64) void k_predictions(const uint32_t* aid, const int64_t* off, int64_t n_sess, const int32_t* cand,
                                                   const int32_t* n_cand, int n_common, const int32_t* freq, int n_freq, int n_pred,
                                                   int32_t* pred, int32_t* n_out) {
    const unsigned lane = lane_id();
    for (int64_t s = blockIdx.x; s < n_sess; s += gridDim.x) {
        const int64_t lo = off[s], hi = off[s + 1];
        int filled = 0;
        // walk the session backwards, 64 events at a time; an event enters if no LATER event holds its aid
        for (int64_t top = hi; top > lo && filled < n_pred; top -= 64) {
            const int64_t i = top - 1 - (int64_t)lane;
            bool first = false;
            uint32_t x = 0;
            if (i >= lo) {
                x = aid[i];
                first = true;
                for (int64_t j = i + 1; j < hi; ++j)
                    if (aid[j] == x) { first = false; break; }
            }
            const uint64_t m = __ballot(first);
            const int pos = filled + (int)__popcll(m & ((1ull << lane) - 1ull));
            if (first && pos < n_pred) pred[s * n_pred + pos] = (int32_t)x;
            filled += (int)__popcll(m);
        }
        if (filled > n_pred) filled = n_pred;
        const int nu = filled;
        // candidates (the session's aids are already removed from them): [:n_pred - len(unique)]
        const int nc = n_cand[s] < n_common ? n_cand[s] : n_common;
        const int take = nc < n_pred - nu ? nc : n_pred - nu;
        if ((int)lane < take) pred[s * n_pred + nu + lane] = cand[s * n_common + lane];
        filled += take > 0 ? take : 0;
        // global most frequent aids: [:n_pred - len(predictions)] -- NOT de-duplicated against the row (as in the reference)
        const int rest = n_pred - filled < n_freq ? n_pred - filled : n_freq;
        if ((int)lane < rest) pred[s * n_pred + filled + lane] = freq[lane];
        filled += rest > 0 ? rest : 0;
        for (int p = filled + (int)lane; p < n_pred; p += 64) pred[s * n_pred + p] = -1;
        if (lane == 0) n_out[s] = filled;
    }
}

// ---- the ranker's candidate table (src/ranker/regular_candidate_generation.py:160-193) ---------------------------------
// One wave per session. A row block = the session's unique aids, most recent first, scored u, u-1, .., 1, followed by the
// lookup's candidates scored with their Counter counts; label = the aid is in the session's label list of the type.
// COUNT: only the number of rows (rows[s] = u + candidates kept). The unique walk is the one of k_predictions: backwards,
// 64 events at a time, an event enters if no LATER event holds its aid.
template <bool COUNT>
__global__ __launch_bounds__(64) void k_ranker_table(const uint32_t* aid, const int64_t* off, int64_t n_sess, const int32_t* cand,
                                                      const int32_t* count, const int32_t* n_cand, int n_common, uint32_t* rows,
                                                      const uint64_t* row_off, const int64_t* label_off, const int32_t* label_aid,
                                                      const int64_t* session_ids, int64_t* o_session, int32_t* o_cand, float* o_score,
                                                      uint8_t* o_label) {
    const unsigned lane = lane_id();
    for (int64_t s = blockIdx.x; s < n_sess; s += gridDim.x) {
        const int64_t lo = off[s], hi = off[s + 1];
        const int nc = n_cand[s] < n_common ? (n_cand[s] > 0 ? n_cand[s] : 0) : n_common;
        const int64_t l0 = (!COUNT && label_off) ? label_off[s] : 0, l1 = (!COUNT && label_off) ? label_off[s + 1] : 0;
        auto is_label = [&](uint32_t x) {
            bool hit = false;
            for (int64_t q = l0; q < l1; ++q) hit = hit || (uint32_t)label_aid[q] == x;
            return hit;
        };
        int u = 0;
        if (!COUNT) {                    // scores of the unique aids need u first: the fill pass reads it from the row offsets
            u = (int)(row_off[s + 1] - row_off[s]) - nc;
        }
        int filled = 0;
        const int64_t base = COUNT ? 0 : (int64_t)row_off[s];
        const int64_t sid = (!COUNT && session_ids) ? session_ids[s] : s;
        for (int64_t top = hi; top > lo; top -= 64) {
            const int64_t i = top - 1 - (int64_t)lane;
            bool first = false;
            uint32_t x = 0;
            if (i >= lo) {
                x = aid[i];
                first = true;
                for (int64_t j = i + 1; j < hi; ++j)
                    if (aid[j] == x) { first = false; break; }
            }
            const uint64_t m = __ballot(first);
            if (!COUNT && first) {
                const int pos = filled + (int)__popcll(m & ((1ull << lane) - 1ull));
                o_session[base + pos] = sid;
                o_cand[base + pos] = (int32_t)x;
                o_score[base + pos] = (float)(u - pos);                 // np.arange(1, u + 1)[::-1]
                if (o_label) o_label[base + pos] = is_label(x) ? 1 : 0;
            }
            filled += (int)__popcll(m);
        }
        if (COUNT) {
            if (lane == 0) rows[s] = (uint32_t)(filled + nc);
            continue;
        }
        for (int c = (int)lane; c < nc; c += 64) {
            const int32_t y = cand[s * n_common + c];
            o_session[base + u + c] = sid;
            o_cand[base + u + c] = y;
            o_score[base + u + c] = (float)count[s * n_common + c];
            if (o_label) o_label[base + u + c] = is_label((uint32_t)y) ? 1 : 0;
        }
    }
}
struct RowsU32 {
    const uint32_t* r;
    __device__ uint64_t operator()(int64_t i) const { return r[i]; }
};

// ---- recency-weighted candidates (section 8 f3) ------------------------------------------------------------------
struct RecencyArgs {
    otto_recency_params p;
    const uint32_t* aid;
    const uint8_t* type;
    const int64_t* sess_off;
    int64_t n_sess;
    int64_t n_events;
    int32_t* out_aid;
    double* out_w;
    int32_t* n_out;
    uint32_t* err;
    const uint32_t* list;          // this launch's sessions (k_cand_classify at 64 events) / their number / dequeue counter
    const uint32_t* list_n;
    uint32_t* work;
};

// One workgroup per session (THREADS = 64 for sessions up to 64 events, 256 beyond). Event i: weight of every curve
// (NumPy's linspace arithmetic: i * step + start, two roundings, last element = stop), first-occurrence flag; a first
// occurrence sums its aid's weighted events in session order (what Counter += does) and ranks itself among the
// other first occurrences by (weight desc, first position asc) = Counter.most_common's stable order.
template <int MAXL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_recency(RecencyArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_aid[MAXL];
    __shared__ __attribute__((aligned(16))) double s_wc[OTTO_RECENCY_MAX_CURVES][MAXL];     // curve weight x type coefficient of event i
    __shared__ __attribute__((aligned(16))) double s_acc[OTTO_RECENCY_MAX_CURVES][MAXL];    // Counter value of the aid first seen at i
    __shared__ __attribute__((aligned(16))) uint8_t s_first[MAXL];
    __shared__ uint32_t s_nu;
    const int tid = threadIdx.x;
    const int NCV = a.p.n_curves;
    __shared__ uint32_t s_next[2];
    const uint32_t n_list = *a.list_n;
    WorkPool wp;
    if (tid == 0) { wp.init(a.work); s_next[0] = wp.take(a.work); }
    __syncthreads();
    uint32_t item = s_next[0];
    for (uint32_t it = 0; item < n_list; ++it) {
        if (tid == 0) s_next[(it + 1u) & 1u] = wp.take(a.work);         // read at the end of this session
        const int64_t s = (int64_t)a.list[item];
        const int64_t lo = a.sess_off[s], hi = a.sess_off[s + 1];
        const int n = (int)(hi - lo);
        if (n > MAXL) {
            if (tid == 0) atomicAdd(a.err, 1u);
            __syncthreads();
            item = s_next[(it + 1u) & 1u];
            continue;
        }
        if (tid == 0) s_nu = 0;
        for (int i = tid; i < n; i += THREADS) {
            s_aid[i] = a.aid[lo + i];
            const uint32_t ty = a.type[lo + i];
            const double coef = ty < 3u ? a.p.type_coef[ty] : 0.0;
            for (int c = 0; c < NCV; ++c) {
                const double start = a.p.start[c], stop = a.p.stop[c];
                double y = start;
                if (n > 1) {
                    const double step = __ddiv_rn(__dsub_rn(stop, start), (double)(n - 1));
                    y = i == n - 1 ? stop : __dadd_rn(__dmul_rn((double)i, step), start);
                }
                s_wc[c][i] = __dmul_rn(__dsub_rn(exp2(y), 1.0), coef);
            }
        }
        __syncthreads();
        // The three session loops read FOUR events per LDS operation (one 16-byte read of aids / two of weights, one 4-byte read of
        // flags): one dependent LDS read + branch per event ran at one LDS round trip per event.
        for (int i = tid; i < n; i += THREADS) {
            const uint32_t ai = s_aid[i];
            uint32_t seen = 0;
#pragma unroll 2
            for (int j0 = 0; j0 < i; j0 += 4) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[j0]);
                seen |= (a4.x == ai ? 1u : 0u) | ((a4.y == ai && j0 + 1 < i) ? 1u : 0u) | ((a4.z == ai && j0 + 2 < i) ? 1u : 0u) |
                        ((a4.w == ai && j0 + 3 < i) ? 1u : 0u);
            }
            const bool first = seen == 0;
            s_first[i] = first ? 1 : 0;
            if (first) {
                atomicAdd(&s_nu, 1u);
                double acc[OTTO_RECENCY_MAX_CURVES];
#pragma unroll
                for (int c = 0; c < OTTO_RECENCY_MAX_CURVES; ++c) acc[c] = 0.0;
                for (int j0 = i & ~3; j0 < n; j0 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[j0]);
                    const uint32_t aj[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                            // in event order: Counter += is a sequence of double adds
                        const int j = j0 + e;
                        if (aj[e] == ai && j >= i && j < n)
                            for (int c = 0; c < NCV; ++c) acc[c] = __dadd_rn(acc[c], s_wc[c][j]);
                    }
                }
                for (int c = 0; c < NCV; ++c) s_acc[c][i] = acc[c];
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += THREADS) {
            if (!s_first[i]) continue;
            for (int c = 0; c < NCV; ++c) {
                const double wi = s_acc[c][i];
                int rank = 0;
#pragma unroll 2
                for (int r0 = 0; r0 < n; r0 += 4) {
                    const uint32_t f4 = *reinterpret_cast<const uint32_t*>(&s_first[r0]);
                    const double2 w01 = *reinterpret_cast<const double2*>(&s_acc[c][r0]);
                    const double2 w23 = *reinterpret_cast<const double2*>(&s_acc[c][r0 + 2]);
                    const double wr[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = r0 + e;
                        const bool fr = ((f4 >> (8 * e)) & 1u) != 0 && r < n;
                        rank += (fr && (wr[e] > wi || (wr[e] == wi && r < i))) ? 1 : 0;
                    }
                }
                const size_t o = (size_t)c * (size_t)a.n_events + (size_t)lo + (size_t)rank;
                a.out_aid[o] = (int32_t)s_aid[i];
                a.out_w[o] = wi;
            }
        }
        if (tid == 0) a.n_out[s] = (int32_t)s_nu;
        __syncthreads();
        item = s_next[(it + 1u) & 1u];
    }
}

// ---- recency branch of the standalone model (covisitation/inference.py:143-199) ----------------------------------------
// One workgroup per session with at least min_unique unique aids. Entries of a target's Counter: the session's first
// occurrences (insertion order = event order; weight = recency sum, then `self count` additions of the bump) and the
// best n_common aids outside the session from otto_cand_lookup_self (inserted after every session aid, in most_common
// order; weight = `count` additions of the bump to 0.0). most_common(n_pred) = rank by (weight desc, insertion asc),
// every entry counts the entries ahead of it.
constexpr int RP_THREADS = 256;
constexpr int RP_MAXC = 64;
struct RecPredArgs {
    otto_recency_pred_params p;
    const uint32_t* aid;
    const uint8_t* type;
    const int64_t* sess_off;
    int64_t n_sess;
    int32_t* pred;
    double* weight;
    int32_t* n_out;
    uint32_t* err;
};

__global__ __launch_bounds__(RP_THREADS) void k_recency_pred(RecPredArgs a) {
    constexpr int MAXL = OTTO_CAND_MAX_SESSION;
    __shared__ uint32_t s_aid[MAXL];
    __shared__ uint8_t s_ty[MAXL];
    __shared__ uint8_t s_first[MAXL];
    __shared__ double s_wc[MAXL];                       // curve weight x type coefficient of event i (current target)
    __shared__ double s_w[MAXL + RP_MAXC];              // final weight of entry e: first occurrence at e, or candidate MAXL + r
    __shared__ uint8_t s_live[MAXL + RP_MAXC];
    __shared__ uint32_t s_nu;
    const int tid = threadIdx.x;
    const int NT = a.p.n_targets, NC = a.p.n_common, NP = a.p.n_pred;
    for (int64_t s = blockIdx.x; s < a.n_sess; s += gridDim.x) {
        const int64_t lo = a.sess_off[s], hi = a.sess_off[s + 1];
        const int n = (int)(hi - lo);
        if (n > MAXL) {
            if (tid == 0) atomicAdd(a.err, 1u);
            continue;
        }
        if (tid == 0) s_nu = 0;
        for (int i = tid; i < n; i += RP_THREADS) { s_aid[i] = a.aid[lo + i]; s_ty[i] = a.type[lo + i]; }
        __syncthreads();
        for (int i = tid; i < n; i += RP_THREADS) {
            const uint32_t ai = s_aid[i];
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && s_aid[j] != ai;
            s_first[i] = first ? 1 : 0;
            if (first) atomicAdd(&s_nu, 1u);
        }
        __syncthreads();
        const int nu = (int)s_nu;
        if (nu < a.p.min_unique) {
            if (tid < NT) a.n_out[(size_t)tid * a.n_sess + s] = -1;
            for (int t = 0; t < NT; ++t)
                for (int q = tid; q < NP; q += RP_THREADS) {
                    a.pred[((size_t)t * a.n_sess + s) * NP + q] = -1;
                    if (a.weight) a.weight[((size_t)t * a.n_sess + s) * NP + q] = 0.0;
                }
            __syncthreads();
            continue;
        }
        for (int t = 0; t < NT; ++t) {
            const double start = a.p.start[t], stop = a.p.stop[t], bump = a.p.bump[t];
            for (int i = tid; i < n; i += RP_THREADS) {
                const uint32_t ty = s_ty[i];
                const double coef = ty < 3u ? a.p.type_coef[ty] : 0.0;
                double y = start;
                if (n > 1) {
                    const double step = __ddiv_rn(__dsub_rn(stop, start), (double)(n - 1));
                    y = i == n - 1 ? stop : __dadd_rn(__dmul_rn((double)i, step), start);
                }
                s_wc[i] = __dmul_rn(__dsub_rn(exp2(y), 1.0), coef);
            }
            for (int e = tid; e < MAXL + RP_MAXC; e += RP_THREADS) s_live[e] = 0;
            __syncthreads();
            const int ncand = min(min(a.p.d_n_cand[t][s], NC), RP_MAXC);
            for (int e = tid; e < MAXL + RP_MAXC; e += RP_THREADS) {
                double w = 0.0;
                int reps = 0;
                bool live = false;
                if (e < n) {
                    if (s_first[e]) {
                        const uint32_t ai = s_aid[e];
                        for (int j = e; j < n; ++j)
                            if (s_aid[j] == ai) w = __dadd_rn(w, s_wc[j]);
                        reps = a.p.d_self_count[t][lo + e];
                        live = true;
                    }
                } else if (e >= MAXL && e - MAXL < ncand) {
                    reps = a.p.d_count[t][(size_t)s * NC + (e - MAXL)];
                    live = true;
                }
                for (int r = 0; r < reps; ++r) w = __dadd_rn(w, bump);
                if (live) { s_w[e] = w; s_live[e] = 1; }
            }
            __syncthreads();
            const size_t ob = ((size_t)t * a.n_sess + s) * NP;
            for (int e = tid; e < MAXL + RP_MAXC; e += RP_THREADS) {
                if (!s_live[e]) continue;
                const double we = s_w[e];
                int rank = 0;
                for (int r = 0; r < n; ++r)
                    if (s_live[r] && (s_w[r] > we || (s_w[r] == we && r < e))) ++rank;
                for (int r = MAXL; r < MAXL + ncand; ++r)
                    if (s_w[r] > we || (s_w[r] == we && r < e)) ++rank;
                if (rank < NP) {
                    a.pred[ob + rank] = e < MAXL ? (int32_t)s_aid[e] : a.p.d_cand[t][(size_t)s * NC + (e - MAXL)];
                    if (a.weight) a.weight[ob + rank] = we;
                }
            }
            const int total = nu + ncand;
            for (int q = total + tid; q < NP; q += RP_THREADS) {
                a.pred[ob + q] = -1;
                if (a.weight) a.weight[ob + q] = 0.0;
            }
            if (tid == 0) a.n_out[(size_t)t * a.n_sess + s] = total < NP ? total : NP;
            __syncthreads();
        }
    }
}

}  // namespace otto

using namespace otto;

extern "C" int otto_recency_candidates(const otto_recency_params* p, const uint32_t* d_aid, const uint8_t* d_type,
                                       const int64_t* d_sess_off, int64_t n_sess, int64_t n_events, int32_t* d_out_aid,
                                       double* d_out_w, int32_t* d_n, void* stream) {
    OTTO_REQUIRE(p && d_sess_off && d_out_aid && d_out_w && d_n, "otto_recency_candidates: null argument");
    OTTO_REQUIRE(p->n_curves >= 1 && p->n_curves <= OTTO_RECENCY_MAX_CURVES, "n_curves must be in [1, %d]", OTTO_RECENCY_MAX_CURVES);
    OTTO_REQUIRE(n_events >= 0, "n_events < 0");
    if (n_sess <= 0) return 0;
    OTTO_REQUIRE(d_aid && d_type, "null event arrays");
    hipStream_t s = (hipStream_t)stream;
    // scratch: err | sessions per class [2] | dequeue counters [2] | ... | ids of the sessions of <= 64 events | ids of the others
    OTTO_REQUIRE(n_sess < (1ll << 32), "more than 2^32 sessions");
    uint32_t* d_hdr = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_RECENCY, 64 + 8 * (size_t)n_sess, (void**)&d_hdr, s));
    OTTO_HIP(hipMemsetAsync(d_hdr, 0, 64, s));
    uint32_t* d_err = d_hdr;
    uint32_t* d_short = d_hdr + 16;
    uint32_t* d_long = d_short + n_sess;
    k_cand_classify<<<(unsigned)((n_sess + 255) / 256), 256, 0, s>>>(d_sess_off, n_sess, 64, d_hdr + 1, d_short, d_long);
    RecencyArgs a;
    memset(&a, 0, sizeof a);
    a.p = *p;
    a.aid = d_aid; a.type = d_type; a.sess_off = d_sess_off; a.n_sess = n_sess; a.n_events = n_events;
    a.out_aid = d_out_aid; a.out_w = d_out_w; a.n_out = d_n; a.err = d_err;
    a.list = d_short; a.list_n = d_hdr + 1; a.work = d_hdr + 3;
    k_recency<64, 64><<<(int)(n_sess < 256 * 32 ? n_sess : 256 * 32), 64, 0, s>>>(a);
    a.list = d_long; a.list_n = d_hdr + 2; a.work = d_hdr + 4;
    k_recency<OTTO_CAND_MAX_SESSION, 256><<<(int)(n_sess < 256 * 8 ? n_sess : 256 * 8), 256, 0, s>>>(a);
    hipError_t le = hipGetLastError();
    uint32_t h_err = 0;
    if (le == hipSuccess) le = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, s);
    if (le == hipSuccess) le = hipStreamSynchronize(s);
    if (le != hipSuccess) { set_error("otto_recency_candidates: %s", hipGetErrorString(le)); return -5; }
    OTTO_REQUIRE(h_err == 0, "%u session(s) longer than %d events", h_err, OTTO_CAND_MAX_SESSION);
    return 0;
}


static int cand_lookup(const otto_cand_params* p, const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off,
                       int64_t n_sess, int32_t* d_cand, int32_t* d_count, int32_t* d_n, int32_t* d_self, void* stream) {
    OTTO_REQUIRE(p && d_sess_off && d_cand && d_count && d_n, "otto_cand_lookup: null argument");
    OTTO_REQUIRE(p->n_aids > 0 && p->n_aids <= (1u << 26), "n_aids must be in [1, 2^26]");
    OTTO_REQUIRE(p->k >= 1 && p->k <= 32, "k must be in [1, 32]");
    for (int m = 0; m < p->n_matrices; ++m) OTTO_REQUIRE(p->mat_k[m] >= 0 && p->mat_k[m] <= 64, "mat_k[%d] must be in [0, 64]", m);
    OTTO_REQUIRE(p->n_matrices >= 1 && p->n_matrices <= OTTO_CAND_MAX_MATRICES, "n_matrices out of range");
    OTTO_REQUIRE(p->n_terms >= 1 && p->n_terms <= OTTO_CAND_MAX_TERMS, "n_terms out of range");
    OTTO_REQUIRE(p->n_common >= 1 && p->n_common <= OTTO_CAND_MAX_COMMON, "n_common must be in [1, 128]");
    for (int m = 0; m < p->n_matrices; ++m) OTTO_REQUIRE(p->d_mat_y[m] && p->d_mat_n[m], "matrix %d is null", m);
    for (int t = 0; t < p->n_terms; ++t) {
        OTTO_REQUIRE(p->term_matrix[t] >= 0 && p->term_matrix[t] < p->n_matrices, "term %d: bad matrix", t);
        OTTO_REQUIRE(p->term_source[t] >= 0 && p->term_source[t] <= OTTO_CAND_SRC_C, "term %d: bad source", t);
    }
    if (n_sess <= 0) return 0;
    OTTO_REQUIRE(d_aid && d_type, "null event arrays");
    hipStream_t s = (hipStream_t)stream;
    // scratch: err | sessions per class [2] | dequeue counters [2] | ... | ids of the short sessions | ids of the long sessions
    OTTO_REQUIRE(n_sess < (1ll << 32), "more than 2^32 sessions");
    uint32_t* d_hdr = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_CAND, 64 + 8 * (size_t)n_sess, (void**)&d_hdr, s));
    OTTO_HIP(hipMemsetAsync(d_hdr, 0, 64, s));
    uint32_t* d_err = d_hdr;
    uint32_t* d_short = d_hdr + 16;
    uint32_t* d_long = d_short + n_sess;
    k_cand_classify<<<(unsigned)((n_sess + 255) / 256), 256, 0, s>>>(d_sess_off, n_sess, CD_SMALL_MAXL, d_hdr + 1, d_short, d_long);
    CandArgs a;
    memset(&a, 0, sizeof a);
    a.p = *p;
    a.aid = d_aid; a.type = d_type; a.sess_off = d_sess_off; a.n_sess = n_sess;
    a.cand = d_cand; a.count = d_count; a.n_out = d_n; a.err = d_err; a.self_count = d_self;
#ifdef OTTO_PHASE_PROF
    static unsigned long long* d_prof = nullptr;
    static hipEvent_t pe[3];
    if (!d_prof) { OTTO_HIP(hipMalloc(&d_prof, 128)); for (auto& e : pe) OTTO_HIP(hipEventCreate(&e)); }
    OTTO_HIP(hipMemsetAsync(d_prof, 0, 128, s));
    a.prof = d_prof;
    a.debug = getenv("OTTO_CAND_DEBUG") ? atoi(getenv("OTTO_CAND_DEBUG")) : 0;
    (void)hipEventRecord(pe[0], s);
#endif
    // short sessions first (most of them), then the long ones, each from its own work list
    a.list = d_short; a.list_n = d_hdr + 1; a.work = d_hdr + 3;
    const int grid_s = (int)(n_sess < 256 * 10 ? n_sess : 256 * 10);
    k_cand<CD_SMALL_MAXL, 10, 128><<<grid_s, 128, 0, s>>>(a);
#ifdef OTTO_PHASE_PROF
    a.prof = d_prof + 8;
    (void)hipEventRecord(pe[1], s);
#endif
    a.list = d_long; a.list_n = d_hdr + 2; a.work = d_hdr + 4;
    const int grid = (int)(n_sess < 256 * 2 ? n_sess : 256 * 2);
    k_cand<OTTO_CAND_MAX_SESSION, 12, 512><<<grid, 512, 0, s>>>(a);
#ifdef OTTO_PHASE_PROF
    {
        (void)hipEventRecord(pe[2], s);
        (void)hipStreamSynchronize(s);
        unsigned long long h[16];
        (void)hipMemcpy(h, d_prof, 128, hipMemcpyDeviceToHost);
        float ms0 = 0.f, ms1 = 0.f;
        (void)hipEventElapsedTime(&ms0, pe[0], pe[1]);
        (void)hipEventElapsedTime(&ms1, pe[1], pe[2]);
        const char* names[7] = {"skip", "A flags+ranks", "B lengths+scan", "clear", "gather+insert", "selection", "filter+write"};
        for (int v = 0; v < 2; ++v) {
            unsigned long long tot = 0;
            for (int i = 0; i < 7; ++i) tot += h[v * 8 + i];
            fprintf(stderr, "[phase-prof] k_cand %s  %.2f ms:", v == 0 ? "short sessions" : "long sessions", v == 0 ? ms0 : ms1);
            for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.1f%%", names[i], tot ? 100.0 * h[v * 8 + i] / tot : 0.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    hipError_t le = hipGetLastError();
    uint32_t err = 0;
    hipError_t ce = hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s);
    hipError_t se = hipStreamSynchronize(s);
    OTTO_REQUIRE(le == hipSuccess && ce == hipSuccess && se == hipSuccess, "k_cand failed: %s", hipGetErrorString(le != hipSuccess ? le : (ce != hipSuccess ? ce : se)));
    // low half: sessions over the length limit; high half: partitions that could not be split any further
    OTTO_REQUIRE((err & 0xFFFFu) == 0, "%u session(s) longer than %d events", err & 0xFFFFu, OTTO_CAND_MAX_SESSION);
    OTTO_REQUIRE((err >> 16) == 0, "%u hash partition(s) of the candidate table could not be split (more than %d stacked partitions or no hash bits left)",
                 err >> 16, CD_STACK);
    return 0;
}

extern "C" int otto_cand_lookup(const otto_cand_params* p, const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off,
                                int64_t n_sess, int32_t* d_cand, int32_t* d_count, int32_t* d_n, void* stream) {
    return cand_lookup(p, d_aid, d_type, d_sess_off, n_sess, d_cand, d_count, d_n, nullptr, stream);
}

extern "C" int otto_cand_lookup_self(const otto_cand_params* p, const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off,
                                     int64_t n_sess, int32_t* d_cand, int32_t* d_count, int32_t* d_n, int32_t* d_self_count,
                                     void* stream) {
    OTTO_REQUIRE(d_self_count, "otto_cand_lookup_self: null d_self_count");
    return cand_lookup(p, d_aid, d_type, d_sess_off, n_sess, d_cand, d_count, d_n, d_self_count, stream);
}

extern "C" int otto_recency_predictions(const otto_recency_pred_params* p, const uint32_t* d_aid, const uint8_t* d_type,
                                        const int64_t* d_sess_off, int64_t n_sess, int32_t* d_pred, double* d_weight, int32_t* d_n,
                                        void* stream) {
    OTTO_REQUIRE(p && d_sess_off && d_pred && d_n, "otto_recency_predictions: null argument");
    OTTO_REQUIRE(p->n_targets >= 1 && p->n_targets <= OTTO_RECENCY_MAX_TARGETS, "n_targets must be in [1, %d]", OTTO_RECENCY_MAX_TARGETS);
    OTTO_REQUIRE(p->n_pred >= 1 && p->n_pred <= 64 && p->n_common >= 1 && p->n_common <= RP_MAXC, "n_pred in [1, 64], n_common in [1, %d]", RP_MAXC);
    for (int t = 0; t < p->n_targets; ++t)
        OTTO_REQUIRE(p->d_cand[t] && p->d_count[t] && p->d_n_cand[t] && p->d_self_count[t], "target %d: null input", t);
    if (n_sess <= 0) return 0;
    OTTO_REQUIRE(d_aid && d_type, "null event arrays");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* d_err = nullptr;
    OTTO_TRY(device_scratch(SCRATCH_RECENCY_PRED, 4, (void**)&d_err, s));
    OTTO_HIP(hipMemsetAsync(d_err, 0, 4, s));
    RecPredArgs a;
    memset(&a, 0, sizeof a);
    a.p = *p; a.aid = d_aid; a.type = d_type; a.sess_off = d_sess_off; a.n_sess = n_sess;
    a.pred = d_pred; a.weight = d_weight; a.n_out = d_n; a.err = d_err;
    k_recency_pred<<<(int)(n_sess < 256 * 8 ? n_sess : 256 * 8), RP_THREADS, 0, s>>>(a);
    hipError_t le = hipGetLastError();
    uint32_t h_err = 0;
    if (le == hipSuccess) le = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, s);
    if (le == hipSuccess) le = hipStreamSynchronize(s);
    if (le != hipSuccess) { set_error("otto_recency_predictions: %s", hipGetErrorString(le)); return -5; }
    OTTO_REQUIRE(h_err == 0, "%u session(s) longer than %d events", h_err, OTTO_CAND_MAX_SESSION);
    return 0;
}

extern "C" int otto_cand_predictions(const uint32_t* d_aid, const int64_t* d_sess_off, int64_t n_sess, const int32_t* d_cand,
                                     const int32_t* d_n_cand, int32_t n_common, const int32_t* d_frequent, int32_t n_frequent,
                                     int32_t n_pred, int32_t* d_pred, int32_t* d_n_pred, void* stream) {
    OTTO_REQUIRE(n_sess >= 0 && n_pred >= 1 && n_pred <= 64 && n_common >= 1 && n_common <= 64 && n_frequent >= 0 && n_frequent <= 64,
                 "n_pred, n_common must be in [1, 64], n_frequent in [0, 64]");
    if (n_sess == 0) return 0;
    OTTO_REQUIRE(d_aid && d_sess_off && d_cand && d_n_cand && d_pred && d_n_pred && (d_frequent || n_frequent == 0), "otto_cand_predictions: null argument");
    k_predictions<<<(unsigned)(n_sess < 256 * 64 ? n_sess : 256 * 64), 64, 0, (hipStream_t)stream>>>(
        d_aid, d_sess_off, n_sess, d_cand, d_n_cand, n_common, d_frequent, n_frequent, n_pred, d_pred, d_n_pred);
    OTTO_HIP(hipGetLastError());
    return 0;
}

extern "C" int64_t otto_cand_ranker_workspace(int64_t n_sess) {
    return (int64_t)(n_sess > 0 ? n_sess : 1) * 4 + (int64_t)scan_partial_bytes(n_sess > 0 ? n_sess : 1) + 64;
}

extern "C" int otto_cand_ranker_rows(const uint32_t* d_aid, const int64_t* d_sess_off, int64_t n_sess, const int32_t* d_n_cand,
                                     int32_t n_common, int64_t* d_row_off, int64_t* h_n_rows, void* d_ws, int64_t ws_bytes,
                                     void* stream) {
    OTTO_REQUIRE(n_sess >= 0 && n_common >= 0 && d_row_off && h_n_rows, "otto_cand_ranker_rows: bad argument");
    hipStream_t s = (hipStream_t)stream;
    *h_n_rows = 0;
    if (n_sess == 0) {
        OTTO_HIP(hipMemsetAsync(d_row_off, 0, 8, s));
        return 0;
    }
    OTTO_REQUIRE(d_aid && d_sess_off && d_n_cand && d_ws, "otto_cand_ranker_rows: null argument");
    OTTO_REQUIRE(ws_bytes >= otto_cand_ranker_workspace(n_sess), "workspace too small");
    uint32_t* rows = (uint32_t*)d_ws;
    uint64_t* partial = (uint64_t*)((char*)d_ws + (((size_t)n_sess * 4 + 63) & ~(size_t)63));
    k_ranker_table<true><<<(unsigned)(n_sess < 256 * 64 ? n_sess : 256 * 64), 64, 0, s>>>(
        d_aid, d_sess_off, n_sess, nullptr, nullptr, d_n_cand, n_common, rows, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
        nullptr, nullptr);
    OTTO_HIP(hipGetLastError());
    OTTO_TRY(device_scan(RowsU32{rows}, n_sess, (uint64_t*)d_row_off, partial, s));
    OTTO_HIP(hipMemcpyAsync(h_n_rows, d_row_off + n_sess, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    return 0;
}

extern "C" int otto_cand_ranker_table(const uint32_t* d_aid, const int64_t* d_sess_off, int64_t n_sess, const int32_t* d_cand,
                                      const int32_t* d_count, const int32_t* d_n_cand, int32_t n_common, const int64_t* d_row_off,
                                      const int64_t* d_label_off, const int32_t* d_label_aid, const int64_t* d_session_ids,
                                      int64_t* d_out_session, int32_t* d_out_cand, float* d_out_score, uint8_t* d_out_label,
                                      void* stream) {
    OTTO_REQUIRE(n_sess >= 0 && n_common >= 0, "otto_cand_ranker_table: bad argument");
    if (n_sess == 0) return 0;
    OTTO_REQUIRE(d_aid && d_sess_off && d_n_cand && d_row_off && d_out_session && d_out_cand && d_out_score && (n_common == 0 || (d_cand && d_count)),
                 "otto_cand_ranker_table: null argument");
    OTTO_REQUIRE(!d_out_label || !d_label_off || d_label_aid, "otto_cand_ranker_table: label lists without aids");
    k_ranker_table<false><<<(unsigned)(n_sess < 256 * 64 ? n_sess : 256 * 64), 64, 0, (hipStream_t)stream>>>(
        d_aid, d_sess_off, n_sess, d_cand, d_count, d_n_cand, n_common, nullptr, (const uint64_t*)d_row_off, d_label_off, d_label_aid,
        d_session_ids, d_out_session, d_out_cand, d_out_score, d_out_label);
    OTTO_HIP(hipGetLastError());
    return 0;
}

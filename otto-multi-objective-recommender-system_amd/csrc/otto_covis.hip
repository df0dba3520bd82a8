// Covisitation-matrix builder for MI355X (gfx950): pair-expand -> inverted run index
// -> per-aid LDS hash reduce + top-k.  C-ABI in include/otto_covis.h; semantics =
// SPEC-COVIS (restating SURVEY.md App. A).  The reference has no builder
// (SURVEY.md F1); the only related reference code is the session self-join of
// src/matrix_factorization/torch_trainer.py:198-223.
//
// Pipeline:
//   winscan  : per session n = min(len, W); exclusive scans of n(n-1) (record slots) and n (run slots)
//   K1 expand: no filter kinds -> k_expand_fused: one launch over the sessions in memory order, windows of 8 / 16 / 32
//              lanes held in registers, row loop with wave-mask predicates, gap-free shortcut; filter kinds ->
//              class-sorted k_expand / k_expand_fast with the LDS ds_min matrix M[class_x][class_y] and filter bits.
//              Either way: one 4-byte record per deduped pair, grouped by aid_x inside the window (a "run"), plus
//              one run descriptor per window event
//   index    : runs split into buckets of 1024 consecutive aids, then counted and placed per bucket with LDS atomics
//              (k_bkt_*; fallback: one 64-bit memory-side atomic per run, k_hist_runs / k_scatter_runs); work-item
//              lists in three size bins (S/M/L; L items are hash partitions of one heavy aid_x, three table layouts)
//   K2 reduce: per work item gather the runs' records (or read the partition bucket), aggregate by aid_y in an LDS
//              hash table (no global atomics), then top-k per kind; heavy aids: partial top-k per partition (one pass
//              when a sibling partition's threshold is known) + exact merge
#include "common.h"
#include "topk.h"
#include "scan.h"
#include "../../include/otto_covis.h"

#include <stdarg.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace otto {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ---------------------------------------------------------------------------
// winscan functors
// ---------------------------------------------------------------------------
struct WinPairs {   // n(n-1) record slots of session i
    const int64_t* off;
    int W;
    __device__ uint64_t operator()(int64_t i) const {
        int64_t n = off[i + 1] - off[i];
        n = n < W ? n : W;
        return n >= 2 ? (uint64_t)(n * (n - 1)) : 0;
    }
};
struct WinEvents {  // n run slots of session i
    const int64_t* off;
    int W;
    __device__ uint64_t operator()(int64_t i) const {
        int64_t n = off[i + 1] - off[i];
        n = n < W ? n : W;
        return n >= 2 ? (uint64_t)n : 0;
    }
};

// ---------------------------------------------------------------------------
// K1: pair-expand
// ---------------------------------------------------------------------------
constexpr uint32_t M_EMPTY = 0xFFFFFFFFu;
constexpr int REC_AID_BITS = 26;
constexpr uint32_t REC_AID_MASK = (1u << REC_AID_BITS) - 1;

// window classes: size (n <= 8 / 16 / 32 events -> G = 8 / 16 / 32 lanes, 8 / 4 / 2 windows per wave) x
// "gap-free" (time span of the whole window <= max_gap: every pair of distinct aids is valid, so the first
// valid pair of (x, y) is (first x, first y) and neither the n^2 ds_min phase nor the M matrix is needed).
constexpr int N_WIN_CLASSES = 6;
__global__ void k_classify(const int64_t* off, const int32_t* ts, int W, int max_gap, int use_fast, int64_t n_sess,
                           uint8_t* cls_out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sess) return;
    const int64_t lo = off[s], hi = off[s + 1];
    int64_t n = hi - lo;
    n = n < W ? n : W;
    uint8_t c = 255;
    if (n >= 2) {
        c = n <= 8 ? 0 : (n <= 16 ? 1 : 2);
        if (use_fast && (int64_t)ts[hi - 1] - (int64_t)ts[hi - n] <= (int64_t)max_gap) c += 3;
    }
    cls_out[s] = c;
}
struct WinClass {   // 1 if session i falls in class `cls`
    const uint8_t* c;
    int cls;
    __device__ uint64_t operator()(int64_t i) const { return c[i] == cls ? 1ull : 0ull; }
};
struct ClassFill {
    const uint64_t* pos[N_WIN_CLASSES];
    uint64_t base[N_WIN_CLASSES];
};
// sess_list[base[c] + rank within class] = session index
__global__ void k_fill_classes(const uint8_t* cls, int64_t n_sess, ClassFill f, uint32_t* sess_list) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sess) return;
    const int c = cls[s];
    if (c < N_WIN_CLASSES) sess_list[f.base[c] + f.pos[c][s]] = (uint32_t)s;
}

// run_x of a run slot without records (a repeated aid of a window): the index passes skip it without reading its descriptor
constexpr uint32_t RUN_X_EMPTY = 0xFFFFFFFFu;

// Run descriptor (u64): len (bits 0-5, <= 32) | first record slot << 8 (40 bits) | sp << 48 (6 bits); 0 = empty run.
// sp: the run reads a SHARED list (the distinct aids of one time-connected component of the window, written once and
// read by the run of every aid in it) and sp is the position of the run's own aid inside that list -- the gather
// skips it; the time extra of such a run is ONE value, kept at tw[slot + sp]. sp = 63: a private row (every record
// is a pair, the time extra is per record).
constexpr int DESC_SLOT_BITS = 40;
constexpr uint64_t DESC_SLOT_MASK = (1ull << DESC_SLOT_BITS) - 1ull;
constexpr uint32_t DESC_SP_NONE = 63u;
__host__ __device__ __forceinline__ uint64_t make_desc(uint64_t slot, uint32_t len, uint32_t sp = DESC_SP_NONE) {
    return len ? ((uint64_t)len | (slot << 8) | ((uint64_t)sp << 48)) : 0ull;
}
__host__ __device__ __forceinline__ uint32_t desc_len(uint64_t d) { return (uint32_t)d & 0x3Fu; }
__host__ __device__ __forceinline__ uint64_t desc_slot(uint64_t d) { return (d >> 8) & DESC_SLOT_MASK; }
__host__ __device__ __forceinline__ uint32_t desc_sp(uint64_t d) { return (uint32_t)(d >> 48) & 63u; }
// pairs the run contributes (the list entry of the run's own aid is not a pair)
__host__ __device__ __forceinline__ uint32_t desc_pairs(uint64_t d) {
    const uint32_t len = desc_len(d);
    return len - ((len && desc_sp(d) != DESC_SP_NONE) ? 1u : 0u);
}

struct ExpandArgs {
    const uint32_t* aid;
    const int32_t* ts;
    const uint8_t* type;
    const int64_t* sess_off;
    const uint64_t* pair_base;   // [S+1] chunk-local exclusive scans
    const uint64_t* ev_base;
    const uint32_t* sess_list;   // sessions of this size class
    int64_t n_list;
    uint32_t* rec;
    uint32_t* tw;
    uint32_t* run_x;
    uint64_t* run_desc;
    uint64_t rec_base;           // first record / run slot of this chunk
    uint64_t run_base;
    uint64_t list_base;          // k_expand_lists: first list slot of this chunk (behind its pair slots)
    int window;
    int max_gap;
    int64_t t0;
    int64_t tspan;
    uint32_t fmask[4];
    int debug;                   // timing diagnostics (wrong results): 1 no record stores, 2 no row loops
};

// (wave_lds_sync: common.h) -- no workgroup barrier anywhere in this kernel.

// G lanes per window, 64/G windows per wave, 4 independent waves per workgroup.
//  A: load the window; class id of an event = first position holding the same aid.
//  B: every ordered pair (i, lane): valid pairs race with ds_min for M[class_x][class_y] = first (i << 5 | j);
//     filter kinds OR their mask bit into FB[class_x][class_y].
//  C: per class row: ballot/popcount compaction of the non-empty columns -> contiguous 4-byte records
//     (a run), one (aid_x, slot << 8 | len) descriptor per window event.
template <int G, bool TIME, bool FILT>
__global__ __launch_bounds__(256) void k_expand(ExpandArgs a) {
    constexpr int WPW = 64 / G;                    // windows per wave
    constexpr int FBW = (G + 7) / 8;               // FB words per row (4 bits per column)
    __shared__ uint4 s_ev[4][WPW][G];              // aid, ts, type | cls << 8, time extra
    __shared__ uint32_t s_M[4][WPW][G * G];
    __shared__ uint32_t s_FB[4][WPW][FILT ? G * FBW : 1];

    const int wv = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    const int grp = lane / G, g = lane % G;
    const unsigned grp_shift = grp * G;
    const uint64_t gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    uint4* ev = s_ev[wv][grp];
    uint32_t* M = s_M[wv][grp];
    uint32_t* FB = s_FB[wv][FILT ? grp : 0];

    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int64_t w0 = wave_global * WPW; w0 < a.n_list; w0 += wave_stride * WPW) {
        const int64_t li = w0 + grp;
        int n = 0;
        int64_t wstart = 0;
        uint64_t pbase = 0, ebase = 0;
        if (li < a.n_list) {
            const int64_t s = a.sess_list[li];
            const int64_t lo = a.sess_off[s], hi = a.sess_off[s + 1];
            const int64_t len = hi - lo;
            n = (int)(len < a.window ? len : a.window);
            wstart = hi - n;
            pbase = a.pair_base[s];
            ebase = a.ev_base[s];
        }
        // wave-uniform trip count: the longest window of this wave
        int nmax = 0;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int nq = __builtin_amdgcn_readlane(n, q * G);
            nmax = nq > nmax ? nq : nmax;
        }
        const int nmax4 = (nmax + 3) & ~3;
        // ---- A ----
        uint32_t aid = 0xFFFFFFFFu, ty = 0, extra = 0;
        int32_t t = 0;
        if (g < n) {
            aid = a.aid[wstart + g];
            t = a.ts[wstart + g];
            ty = a.type[wstart + g];
            if (TIME) extra = a.tspan > 0 ? (uint32_t)((uint64_t)(196608ull * (uint64_t)((int64_t)t - a.t0)) / (uint64_t)a.tspan) : 0u;
        }
        ev[g] = make_uint4(aid, (uint32_t)t, ty, extra);
        wave_lds_sync();
        int cls = g;
        for (int j0 = nmax4 - 4; j0 >= 0; j0 -= 4) {      // LDS reads four at a time: latency paid once per group
            uint32_t ax[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ax[u] = ev[(j0 + u) & (G - 1)].x;
#pragma unroll
            for (int u = 3; u >= 0; --u)
                if (j0 + u < n && ax[u] == aid) cls = j0 + u;
        }
        if (g < n) {
            ev[g].z = ty | ((uint32_t)cls << 8);
            if (FILT)
                for (int q = 0; q < FBW; ++q) FB[g * FBW + q] = 0;
        }
        for (int r = 0; r < nmax; ++r)
            if (r < n) M[r * G + g] = M_EMPTY;
        wave_lds_sync();
        // ---- B ----
        for (int i0 = 0; i0 < nmax; i0 += 4) {
            uint4 e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e4[u] = ev[(i0 + u) & (G - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                const uint4 e = e4[u];
                int dt = (int)e.y - t;
                dt = dt < 0 ? -dt : dt;
                if (i < n && g < n && e.x != aid && dt <= a.max_gap) {
                    const uint32_t ci = e.z >> 8;
                    atomicMin(&M[ci * G + cls], (uint32_t)((i << 5) | g));
                    if (FILT) {
                        const uint32_t bit = (e.z & 0xFFu) * 3 + ty;
                        const uint32_t fb = ((a.fmask[0] >> bit) & 1u) | (((a.fmask[1] >> bit) & 1u) << 1) |
                                            (((a.fmask[2] >> bit) & 1u) << 2) | (((a.fmask[3] >> bit) & 1u) << 3);
                        if (fb) atomicOr(&FB[ci * FBW + (cls >> 3)], fb << ((cls & 7) * 4));
                    }
                }
            }
        }
        wave_lds_sync();
        // ---- C ----
        uint32_t off = 0, my_off = 0, my_cnt = 0;
        for (int r0 = 0; r0 < nmax; r0 += 4) {
            uint32_t m4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) m4[u] = (r0 + u < n && g < n) ? M[((r0 + u) & (G - 1)) * G + g] : M_EMPTY;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + u;
                const uint32_t e = m4[u];
                const bool has = e != M_EMPTY;
                const uint32_t hm = (uint32_t)((__ballot(has) >> grp_shift) & gmask);
                const uint32_t cnt = __popc(hm);
                if (has) {
                    const uint32_t rank = __popc(hm & ((1u << g) - 1u));
                    const uint32_t i = e >> 5, j = e & 31u;
                    const uint32_t tyj = ev[j].z & 0xFFu;
                    const uint32_t fb = FILT ? (FB[r * FBW + (g >> 3)] >> ((g & 7) * 4)) & 0xFu : 0u;
                    const uint64_t slot = a.rec_base + pbase + off + rank;
                    a.rec[slot] = aid | (tyj << REC_AID_BITS) | (fb << 28);
                    if (TIME) a.tw[slot] = ev[i].w;
                }
                if (g == r) { my_off = off; my_cnt = cnt; }
                off += cnt;
            }
        }
        if (g < n) {
            a.run_x[a.run_base + ebase + g] = my_cnt ? aid : RUN_X_EMPTY;
            a.run_desc[a.run_base + ebase + g] = make_desc(a.rec_base + pbase + my_off, my_cnt);
        }
        wave_lds_sync();
    }
}

// Gap-free windows (no filter kinds): class ids, then per class row r the columns are simply the other class
// representatives -- record = aid_y | type(first y) << 26, time extra of the first x. No M, no LDS atomics.
template <int G, bool TIME>
__global__ __launch_bounds__(256) void k_expand_fast(ExpandArgs a) {
    constexpr int WPW = 64 / G;
    __shared__ uint32_t s_aid[4][WPW][G];
    const int wv = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    const int grp = lane / G, g = lane % G;
    const unsigned grp_shift = grp * G;
    const uint64_t gmask = (1ull << G) - 1ull;
    uint32_t* av = s_aid[wv][grp];
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    for (int64_t w0 = wave_global * WPW; w0 < a.n_list; w0 += wave_stride * WPW) {
        const int64_t li = w0 + grp;
        int n = 0;
        int64_t wstart = 0;
        uint64_t pbase = 0, ebase = 0;
        if (li < a.n_list) {
            const int64_t s = a.sess_list[li];
            const int64_t lo = a.sess_off[s], hi = a.sess_off[s + 1];
            const int64_t len = hi - lo;
            n = (int)(len < a.window ? len : a.window);
            wstart = hi - n;
            pbase = a.pair_base[s];
            ebase = a.ev_base[s];
        }
        int nmax = 0;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int nq = __builtin_amdgcn_readlane(n, q * G);
            nmax = nq > nmax ? nq : nmax;
        }
        const int nmax4 = (nmax + 3) & ~3;
        uint32_t aid = 0xFFFFFFFFu, ty = 0, extra = 0;
        if (g < n) {
            aid = a.aid[wstart + g];
            ty = a.type[wstart + g];
            if (TIME) {
                const int32_t t = a.ts[wstart + g];
                extra = a.tspan > 0 ? (uint32_t)((uint64_t)(196608ull * (uint64_t)((int64_t)t - a.t0)) / (uint64_t)a.tspan) : 0u;
            }
        }
        av[g] = aid;
        wave_lds_sync();
        int cls = g;
        for (int j0 = nmax4 - 4; j0 >= 0; j0 -= 4) {
            uint32_t ax[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ax[u] = av[(j0 + u) & (G - 1)];
#pragma unroll
            for (int u = 3; u >= 0; --u)
                if (j0 + u < n && ax[u] == aid) cls = j0 + u;
        }
        const bool rep = g < n && cls == g;
        const uint32_t repm = (uint32_t)((__ballot(rep) >> grp_shift) & gmask);     // class representatives of my window
        const uint32_t d = __popc(repm);
        const uint32_t below = __popc(repm & ((1u << g) - 1u));                      // representatives before me
        const uint32_t rec_word = aid | (ty << REC_AID_BITS);
        // row r (a representative) lists every other representative in position order: d - 1 records
        for (int r = 0; r < nmax; ++r) {
            const bool row = (repm >> r) & 1u;                // uniform inside the window
            // time extra of the row's event: read with ALL lanes active (the source lane g == r sits out the branch)
            const uint32_t xr = TIME ? (uint32_t)__shfl((int)extra, (int)(grp_shift + r), 64) : 0u;
            if (row && rep && g != r) {
                const uint32_t rowrank = __popc(repm & ((1u << r) - 1u));
                const uint32_t rank = below - (g > r ? 1u : 0u);
                const uint64_t slot = a.rec_base + pbase + (uint64_t)rowrank * (d - 1) + rank;
                a.rec[slot] = rec_word;
                if (TIME) a.tw[slot] = xr;
            }
        }
        if (g < n) {
            const bool has = rep && d > 1;
            a.run_x[a.run_base + ebase + g] = has ? aid : RUN_X_EMPTY;
            a.run_desc[a.run_base + ebase + g] = make_desc(a.rec_base + pbase + (uint64_t)below * (d - 1), has ? d - 1 : 0u);
        }
        wave_lds_sync();
    }
}

// ---- fused, register-only expansion (no filter kinds) -------------------------------------------------
// Sessions are taken in MEMORY ORDER, 64 per wave round: the wave reads their offsets / slot bases coalesced,
// sorts them by window class (size n <= 8 / 16 / 32  x  gap-free or not) into a wave-private LDS task list
// with ballots, then runs the six classes back to back with G = 8 / 16 / 32 lanes per window. Neighbouring
// sessions are expanded by the same wave, so every event cache line is fetched once (the class-sorted launches
// above fetch it up to 6x), and there is one launch instead of six.
//
// A window lives in registers: lane g = event j; `same` = lanes of my window holding my aid (my class),
// class id = first such lane, reps = first lanes. Row class r owns slots [rank(r) * (d - 1), +d - 1) of the
// window (d distinct aids). Masks are kept in WAVE bit space (lane numbers), so a ballot is used as it comes.
//
// gap-free windows (span <= max_gap: every pair of distinct aids is valid, first pair = (first x, first y)):
//   loop over row ranks k: every rep lane stores its own word into row k (skipping its own row).
// general windows: row loop over i (uniform): pair (i, g) valid if aids differ and |dt| <= max_gap. The class
//   pair (class(i), my class) is taken by the FIRST valid (i, j) in lexicographic order = first row whose valid
//   mask meets my class (`done` bit per x class: identical in all lanes of a class because it only depends on
//   ballot & same), lowest valid lane of the class. The row's fill level = number of column classes that
//   already took the pair. Run length of class r = popc(done of r): the relation "x and y have a valid pair" is
//   symmetric (no filter masks here). No M matrix, no LDS atomics: LDS only broadcasts one uint4 per row.
struct TaskPre {            // one task of a lane's window, its event already requested from memory
    int n;
    uint64_t pb, eb;
    uint32_t aid, ty;
    int32_t t;
};

// task `t0 + lane / G` of the list segment [b0, b0 + c): window descriptor from LDS, then the lane's event loads
// are ISSUED here -- the caller runs the previous task's row loop before touching the values.
template <int G, bool NEED_TS>
__device__ __forceinline__ TaskPre fetch_task(const ExpandArgs& a, const uint4* ta, const uint2* tb, int b0, int c, int t0,
                                              unsigned lane) {
    TaskPre p;
    p.n = 0; p.pb = p.eb = 0; p.aid = 0xFFFFFFFFu; p.ty = 0; p.t = 0;
    const int ti = t0 + (int)(lane / G);
    int64_t ws = 0;
    if (ti < c) {
        const uint4 A = ta[b0 + ti];
        const uint2 B = tb[b0 + ti];
        p.n = (int)(A.y >> 16);
        ws = (int64_t)(((uint64_t)(A.y & 0xFFFFu) << 32) | A.x);
        p.pb = ((uint64_t)A.w << 32) | A.z;
        p.eb = ((uint64_t)B.y << 32) | B.x;
    }
    const int g = (int)(lane & (G - 1));
    if (g < p.n) {
        p.aid = a.aid[ws + g];
        p.ty = a.type[ws + g];
        if (NEED_TS) p.t = a.ts[ws + g];
    }
    return p;
}

template <int G, bool TIME, bool FAST, bool DBG>
__device__ __forceinline__ void expand_task_reg(const ExpandArgs& a, uint4* ev, uint32_t* xs, unsigned lane, const TaskPre& tp) {
    constexpr int WPW = 64 / G;
    const int n = tp.n;
    const uint64_t pbase = tp.pb, ebase = tp.eb;
    const int g = (int)(lane & (G - 1));
    const unsigned w0 = lane - g;                               // first lane of my window
    uint4* evw = ev + w0;
    int nmax = 0;
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int nq = __builtin_amdgcn_readlane(n, q * G);
        nmax = nq > nmax ? nq : nmax;
    }
    const int nmax4 = (nmax + 3) & ~3;
    const bool act = g < n;
    const uint32_t aid = tp.aid, ty = tp.ty;
    const int32_t t = tp.t;
    uint32_t extra = 0;
    if (TIME && act) extra = a.tspan > 0 ? (uint32_t)((uint64_t)(196608ull * (uint64_t)((int64_t)t - a.t0)) / (uint64_t)a.tspan) : 0u;
    ev[lane] = make_uint4(aid, (uint32_t)t, 0u, extra);
    wave_lds_sync();
    uint32_t same = 0;                                          // window-relative bits
    for (int j0 = 0; j0 < nmax4; j0 += 4) {
        uint32_t ax[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ax[u] = evw[(j0 + u) & (G - 1)].x;
#pragma unroll
        for (int u = 0; u < 4; ++u) same |= (ax[u] == aid ? 1u : 0u) << ((j0 + u) & 31);
    }
    same &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    const int cls = act ? (int)__builtin_ctz(same) : g;
    const bool rep = act && cls == g;
    // wave-space 64-bit masks of my window / the lanes of my window below me
    const uint64_t win64 = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << w0;
    const uint64_t lw64 = win64 & ((1ull << lane) - 1ull);
    const uint64_t rep64 = __ballot(rep);
    const uint32_t d1 = (uint32_t)__popcll(rep64 & win64) - 1u;                  // records a full row holds
    uint32_t* recp = a.rec + (a.rec_base + pbase);
    uint32_t* twp = TIME ? a.tw + (a.rec_base + pbase) : nullptr;
    const uint32_t word = aid | (ty << REC_AID_BITS);
    if (FAST) {
        const uint32_t below = (uint32_t)__popcll(rep64 & lw64);                 // my rank among the reps
        if (TIME) {
            if (rep) xs[w0 + below] = extra;
            wave_lds_sync();
        }
        int dmax = 0;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int dq = __builtin_amdgcn_readlane((int)d1, q * G) + 1;
            dmax = dq > dmax ? dq : dmax;
        }
        if (DBG && (a.debug & 2)) dmax = 0;
        for (int k = 0; k < dmax; ++k) {
            if (!(DBG && (a.debug & 1)) && rep && (uint32_t)k <= d1 && (uint32_t)k != below) {
                const uint32_t o = (uint32_t)k * d1 + below - (below > (uint32_t)k ? 1u : 0u);
                recp[o] = word;
                if (TIME) twp[o] = xs[w0 + k];
            }
        }
        if (act) {
            a.run_x[a.run_base + ebase + g] = (rep && d1) ? aid : RUN_X_EMPTY;
            a.run_desc[a.run_base + ebase + g] = make_desc(a.rec_base + pbase + (uint64_t)below * d1, rep ? d1 : 0u);
        }
        wave_lds_sync();
        return;
    }
    const uint32_t rb = (uint32_t)__popcll(rep64 & win64 & ((1ull << (w0 + cls)) - 1ull)) * d1;   // first slot of my class's row
    reinterpret_cast<uint32_t*>(&ev[lane])[2] = (uint32_t)cls | (rb << 5);
    wave_lds_sync();
    const uint64_t same64 = (uint64_t)same << w0;
    const uint64_t samelow64 = same64 & lw64;
    const uint32_t nlim = act ? (uint32_t)n : 0u;
    uint32_t done = 0;
    uint4 e = evw[0];
    if (DBG && (a.debug & 2)) nmax = 0;
    // Predicates are kept as wave masks (v_cmp writes them, s_and combines them, exec takes them): the whole wave is
    // active here, so a mask IS the ballot -- no select/compare round trip per ballot.
    constexpr int NE = 33, EQ = 32, ULT = 36, ULE = 37;
    for (int i = 0; i < nmax; ++i) {
        const uint4 en = evw[(i + 1) & (G - 1)];                // next row in flight
        const uint32_t dA = e.y - (uint32_t)t, dB = (uint32_t)t - e.y;
        const uint32_t ad = dA < dB ? dA : dB;                    // |dt| (|dt| < 2^31)
        const uint64_t V = __builtin_amdgcn_uicmp((uint32_t)i, nlim, ULT) & __builtin_amdgcn_uicmp(e.x, aid, NE) &
                           __builtin_amdgcn_uicmp(ad, (uint32_t)a.max_gap, ULE);        // valid pairs (i, lane)
        if (V != 0) {                                             // uniform: rows without any valid pair cost nothing more
            const uint32_t bit = 1u << (e.z & 31u);               // x class of this row
            const uint64_t S = __builtin_amdgcn_uicmp(done & bit, 0u, NE);              // my class already took (x class, me)
            const uint64_t M = __builtin_amdgcn_uicmpl(V & same64, 0ull, NE);           // my class meets this row
            const uint64_t F = __builtin_amdgcn_uicmpl(V & samelow64, 0ull, EQ);        // no lower lane of my class is valid
            const uint64_t E = V & ~S & F;                                              // lanes that emit
            const uint64_t D = rep64 & S;                                               // column classes already in the row
            if (__builtin_amdgcn_inverse_ballot_w64(M)) done |= bit;
            if (__builtin_amdgcn_inverse_ballot_w64(E) && !(DBG && (a.debug & 1))) {
                const uint32_t o = (e.z >> 5) + (uint32_t)__popcll(D & win64) + (uint32_t)__popcll(E & lw64);
                recp[o] = word;
                if (TIME) twp[o] = e.w;
            }
        }
        e = en;
    }
    if (act) {
        const uint32_t len = rep ? __popc(done) : 0u;
        a.run_x[a.run_base + ebase + g] = len ? aid : RUN_X_EMPTY;
        a.run_desc[a.run_base + ebase + g] = make_desc(a.rec_base + pbase + rb, len);
    }
    wave_lds_sync();
}

// DBG: the timing diagnostics of ExpandArgs::debug are compiled into a second instantiation only
template <bool TIME, bool DBG>
__global__ __launch_bounds__(256) void k_expand_fused(ExpandArgs a, int64_t n_sess, int use_fast) {
    __shared__ uint4 s_ev[4][64];
    __shared__ uint32_t s_xs[4][TIME ? 64 : 1];
    __shared__ uint4 s_ta[4][64];       // task: wstart lo, wstart hi | n << 16, pair_base lo, hi
    __shared__ uint2 s_tb[4][64];       //       ev_base lo, hi
    const int wv = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    uint4* ev = s_ev[wv];
    uint32_t* xs = s_xs[wv];
    uint4* ta = s_ta[wv];
    uint2* tb = s_tb[wv];
    const int64_t n_tiles = (n_sess + 63) / 64;
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // session metadata of a tile (one session per lane), requested one tile ahead
    struct TilePre { int64_t lo, hi; uint64_t pb, eb; };
    auto fetch_tile = [&](int64_t tile) {
        TilePre q;
        q.lo = q.hi = 0; q.pb = q.eb = 0;
        const int64_t s = tile * 64 + lane;
        if (tile < n_tiles && s < n_sess) {
            q.lo = a.sess_off[s]; q.hi = a.sess_off[s + 1];
            q.pb = a.pair_base[s]; q.eb = a.ev_base[s];
        }
        return q;
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    TilePre cur = fetch_tile(tile);
    for (; tile < n_tiles; tile += tile_stride) {
        const TilePre nxt = fetch_tile(tile + tile_stride);
        const int64_t len = cur.hi - cur.lo;
        const int n = (int)(len < a.window ? len : a.window);
        const int64_t wstart = cur.hi - n;
        int c6 = 6;
        if (n >= 2) {
            c6 = n <= 8 ? 0 : (n <= 16 ? 1 : 2);
            if (use_fast && (int64_t)a.ts[cur.hi - 1] - (int64_t)a.ts[wstart] <= (int64_t)a.max_gap) c6 += 3;
        }
        int cnt[6], base[6];
        int below = 0, acc = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const uint64_t m = __ballot(c6 == q);
            cnt[q] = __popcll(m);
            base[q] = acc;
            if (c6 == q) below = acc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            acc += cnt[q];
        }
        if (c6 < 6) {
            ta[below] = make_uint4((uint32_t)wstart, (uint32_t)((uint64_t)wstart >> 32) | ((uint32_t)n << 16), (uint32_t)cur.pb, (uint32_t)(cur.pb >> 32));
            tb[below] = make_uint2((uint32_t)cur.eb, (uint32_t)(cur.eb >> 32));
        }
        wave_lds_sync();
        // classes run back to back; the events of the NEXT task (same class, or the first task of the next class)
        // are requested before the current task's row loop
        auto run_class = [&](auto gtag, auto ftag, auto gntag, auto fntag, int q, const TaskPre& first) {
            constexpr int G = decltype(gtag)::value;
            constexpr bool FAST = decltype(ftag)::value;
            constexpr int GN = decltype(gntag)::value;
            constexpr bool FASTN = decltype(fntag)::value;
            constexpr int WPW = 64 / G;
            const int b0 = base[q], c = cnt[q];
            const int bn = q < 5 ? base[q < 5 ? q + 1 : 5] : 0, cn = q < 5 ? cnt[q < 5 ? q + 1 : 5] : 0;
            TaskPre tp = first;
            for (int t0 = 0; t0 < c; t0 += WPW) {
                TaskPre nx;
                if (t0 + WPW < c) nx = fetch_task<G, !FAST || TIME>(a, ta, tb, b0, c, t0 + WPW, lane);
                else nx = fetch_task<GN, !FASTN || TIME>(a, ta, tb, bn, cn, 0, lane);
                expand_task_reg<G, TIME, FAST, DBG>(a, ev, xs, lane, tp);
                tp = nx;
            }
            if (c == 0) tp = fetch_task<GN, !FASTN || TIME>(a, ta, tb, bn, cn, 0, lane);
            return tp;
        };
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        using I32 = std::integral_constant<int, 32>;
        TaskPre tp = fetch_task<8, true>(a, ta, tb, base[0], cnt[0], 0, lane);
        tp = run_class(I8{}, std::false_type{}, I16{}, std::false_type{}, 0, tp);
        tp = run_class(I16{}, std::false_type{}, I32{}, std::false_type{}, 1, tp);
        tp = run_class(I32{}, std::false_type{}, I8{}, std::true_type{}, 2, tp);
        tp = run_class(I8{}, std::true_type{}, I16{}, std::true_type{}, 3, tp);
        tp = run_class(I16{}, std::true_type{}, I32{}, std::true_type{}, 4, tp);
        tp = run_class(I32{}, std::true_type{}, I32{}, std::true_type{}, 5, tp);
        cur = nxt;
    }
}

// ---- component lists (no filter kinds): a window's pairs WITHOUT writing one record per pair ---------------------
// Events of a session arrive sorted by time, so "valid pair" (|dt| <= max_gap) is a band over the window positions.
// Cut the window where two neighbouring events are more than max_gap apart: pairs across a cut are invalid, and a
// piece ("component") whose whole time span is <= max_gap is a clique -- every two events of it are a valid pair
// (OTTO shape: 99 % of the windows consist of cliques only; a gap-free window is the one-component case). Then
//   * the first valid pair (i, j) of (x, y) in lexicographic order lies in the EARLIEST component that holds both:
//     i = first x there, j = first y there;
//   * so the row of x inside a component is the component's list of distinct aids (each with the type of its first
//     event in the component) minus x itself, minus the aids y that already met x in an earlier component.
// The list is written ONCE per component (one store per first occurrence: <= n words per window instead of n(n-1)),
// and the run descriptor of every aid in it points at the shared list with the position of its own entry (sp), which
// the gather skips. Only an aid with a repeated partner (x and y together in two components: 2.4 % of the runs of
// multi-component windows) gets a private row, written by the component's lanes in one step. Windows with a component
// that is not a clique, or with unsorted timestamps (1 % of the windows), take the general row loop of
// expand_task_reg into private rows. The time extra of a shared run is ONE value (the first event of x in the
// component): tw[list slot of x's own entry].
// Slots: lists live behind the chunk's pair slots at list_base + ev_base[s] (n per window, dense: neighbouring
// windows share cache lines); private rows keep the window's old region [pair_base[s], +n(n-1)).
template <int G, bool TIME, bool DBG>
__device__ __forceinline__ void expand_task_lists(const ExpandArgs& a, uint4* ev, unsigned lane, const TaskPre& tp) {
    constexpr int WPW = 64 / G;
    const int n = tp.n;
    const uint64_t pbase = tp.pb, ebase = tp.eb;
    const int g = (int)(lane & (G - 1));
    const unsigned w0 = lane - g;                               // first lane of my window
    uint4* evw = ev + w0;
    int nmax = 0;
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int nq = __builtin_amdgcn_readlane(n, q * G);
        nmax = nq > nmax ? nq : nmax;
    }
    const int nmax4 = (nmax + 3) & ~3;
    const bool act = g < n;
    const uint32_t aid = tp.aid, ty = tp.ty;
    const int32_t t = tp.t;
    uint32_t extra = 0;
    if (TIME && act) extra = a.tspan > 0 ? (uint32_t)((uint64_t)(196608ull * (uint64_t)((int64_t)t - a.t0)) / (uint64_t)a.tspan) : 0u;
    ev[lane] = make_uint4(aid, (uint32_t)t, 0u, extra);
    wave_lds_sync();
    uint32_t same = 0;                                          // window-relative bits: events holding my aid
    if (!(DBG && (a.debug & 4))) {
        // from the last event down: the compare result is shifted in as the carry of same + same, two VALU
        // instructions per event (compare, add-with-carry) instead of compare / select / shift-or
        for (int j0 = nmax4 - 4; j0 >= 0; j0 -= 4) {
            uint32_t ax[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ax[u] = evw[(j0 + u) & (G - 1)].x;
#pragma unroll
            for (int u = 3; u >= 0; --u)
                asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(same) : "v"(ax[u]), "v"(aid) : "vcc");
        }
    } else same = 1u << g;
    same &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    // a wave mask restricted to my window, bit 0 = the window's first lane
    const bool hi_half = lane >= 32u;
    auto win32 = [&](uint64_t m) -> uint32_t {
        uint32_t h = hi_half ? (uint32_t)(m >> 32) : (uint32_t)m;
        if (G < 32) h = (h >> (w0 & 31u)) & ((1u << (G & 31)) - 1u);
        return h;
    };
    const uint32_t lem = (2u << g) - 1u, ltm = (1u << g) - 1u;     // window bits <= me / < me
    // components: a new one starts where the previous event is more than max_gap older
    const int32_t tprev = (int32_t)ev[(lane - 1u) & 63u].y;
    const bool unsorted = act && g > 0 && t < tprev;
    const bool bnd = act && (g == 0 || (uint32_t)(t - tprev) > (uint32_t)a.max_gap);
    const uint32_t Bw = win32(__ballot(bnd));
    const uint32_t cs = act ? 31u - (uint32_t)__clz((int)(Bw & lem)) : 0u;                 // first position of my component
    const uint32_t above = (Bw & ~lem) | (n < 32 ? (1u << (n & 31)) : 0u);
    const uint32_t ce = above ? (uint32_t)__builtin_ctz(above) : 32u;                      // one past its last position
    const uint32_t ltc = (1u << cs) - 1u;                                                  // window bits before my component
    const uint32_t cm = act ? ((0xFFFFFFFFu >> ((32u - ce) & 31u)) & ~ltc) : 0u;
    const int32_t tcs = (int32_t)evw[cs].y;
    const bool viol = unsorted || (act && (uint32_t)(t - tcs) > (uint32_t)a.max_gap);      // my component is not a clique
    const bool general = win32(__ballot(viol)) != 0u;                                      // whole window: general row loop
    const bool repc = act && !general && ((same & cm) & ltm) == 0u;                        // first event of my aid in my component
    const uint32_t Rw = win32(__ballot(repc));
    const uint32_t dc = (uint32_t)__popc(Rw & cm);                                         // entries of my component's list
    const uint32_t lpos = (uint32_t)__popc(Rw & ltm);                                      // my entry inside the window's lists
    const uint32_t lb = (uint32_t)__popc(Rw & ltc);                                        // first entry of my component's list
    const uint64_t L0 = a.list_base + ebase;
    const uint32_t word = aid | (ty << REC_AID_BITS);
    if (repc && !(DBG && (a.debug & 1))) {
        a.rec[L0 + lpos] = word;
        if (TIME) a.tw[L0 + lpos] = extra;
    }
    // repeated partners: x (first event in component c) met y in an earlier component
    const uint32_t E = same & ltc;                                                         // my aid in earlier components
    const bool reocc = repc && E != 0u;
    uint64_t RO = __ballot(reocc);
    bool priv = false;
    uint32_t poff = 0, pslot = 0, plen = 0;                                                // private rows of my window so far
    if (RO != 0 && !(DBG && (a.debug & 2))) {
        const uint32_t key = (uint32_t)__popc(Bw & lem) | (w0 << 5);                       // (window, component)
        uint32_t PC = 0;                                                                   // earlier components holding my aid
        uint32_t Er = reocc ? E : 0u;
        while (__ballot(Er != 0u) != 0) {
            if (Er) {
                const uint32_t e = (uint32_t)__builtin_ctz(Er);
                Er &= Er - 1u;
                PC |= 1u << (((uint32_t)__popc(Bw & ((2u << e) - 1u)) - 1u) & 31u);
            }
        }
        while (RO != 0) {
            const int L = __builtin_ctzll(RO);
            RO &= RO - 1ull;
            const uint32_t PCx = (uint32_t)__builtin_amdgcn_readlane((int)PC, L);
            const uint32_t kx = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
            const bool incomp = repc && key == kx && (int)lane != L;                       // the other entries of x's list
            const bool hit = incomp && (PC & PCx) != 0u;
            if (__ballot(hit) == 0) continue;
            const bool put = incomp && !hit;
            const uint64_t EP = __ballot(put);
            const uint32_t len = (uint32_t)__popcll(EP);
            if (put && !(DBG && (a.debug & 1))) {
                const uint64_t o = a.rec_base + pbase + poff + (uint32_t)__popcll(EP & ((1ull << lane) - 1ull));
                a.rec[o] = word;
                if (TIME) a.tw[o] = (uint32_t)__builtin_amdgcn_readlane((int)extra, L);
            }
            if ((int)lane == L) { priv = true; pslot = poff; plen = len; }
            if ((lane ^ (unsigned)L) < (unsigned)G) poff += len;                            // lanes of x's window
        }
    }
    // general windows: the row loop of expand_task_reg (first valid pair per (x class, y class), private rows)
    uint32_t glen = 0, grb = 0;
    const bool gact = act && general;
    if (__ballot(gact) != 0) {
        const uint64_t win64 = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << w0;
        const uint64_t lw64 = win64 & ((1ull << lane) - 1ull);
        const int cls = act ? (int)__builtin_ctz(same) : g;
        const bool rep = gact && cls == g;
        const uint64_t rep64 = __ballot(rep);
        const uint32_t d1 = (uint32_t)__popcll(rep64 & win64) - 1u;
        const uint32_t rb = (uint32_t)__popcll(rep64 & win64 & ((1ull << (w0 + cls)) - 1ull)) * d1;
        reinterpret_cast<uint32_t*>(&ev[lane])[2] = (uint32_t)cls | (rb << 5);
        wave_lds_sync();
        const uint64_t same64 = (uint64_t)same << w0;
        const uint64_t samelow64 = same64 & lw64;
        const uint32_t nlim = gact ? (uint32_t)n : 0u;
        int gmax = 0;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int nq = __builtin_amdgcn_readlane((int)nlim, q * G);
            gmax = nq > gmax ? nq : gmax;
        }
        uint32_t done = 0;
        uint4 e = evw[0];
        uint32_t* recp = a.rec + (a.rec_base + pbase);
        uint32_t* twp = TIME ? a.tw + (a.rec_base + pbase) : nullptr;
        constexpr int NE = 33, EQ = 32, ULT = 36, ULE = 37;
        for (int i = 0; i < gmax; ++i) {
            const uint4 en = evw[(i + 1) & (G - 1)];
            const uint32_t dA = e.y - (uint32_t)t, dB = (uint32_t)t - e.y;
            const uint32_t ad = dA < dB ? dA : dB;
            const uint64_t V = __builtin_amdgcn_uicmp((uint32_t)i, nlim, ULT) & __builtin_amdgcn_uicmp(e.x, aid, NE) &
                               __builtin_amdgcn_uicmp(ad, (uint32_t)a.max_gap, ULE);
            if (V != 0) {
                const uint32_t bit = 1u << (e.z & 31u);
                const uint64_t S = __builtin_amdgcn_uicmp(done & bit, 0u, NE);
                const uint64_t M = __builtin_amdgcn_uicmpl(V & same64, 0ull, NE);
                const uint64_t F = __builtin_amdgcn_uicmpl(V & samelow64, 0ull, EQ);
                const uint64_t Em = V & ~S & F;
                const uint64_t D = rep64 & S;
                if (__builtin_amdgcn_inverse_ballot_w64(M)) done |= bit;
                if (__builtin_amdgcn_inverse_ballot_w64(Em) && !(DBG && (a.debug & 1))) {
                    const uint32_t o = (e.z >> 5) + (uint32_t)__popcll(D & win64) + (uint32_t)__popcll(Em & lw64);
                    recp[o] = word;
                    if (TIME) twp[o] = e.w;
                }
            }
            e = en;
        }
        glen = rep ? (uint32_t)__popc(done) : 0u;
        grb = rb;
    }
    if (act) {
        uint64_t d;
        if (general) d = make_desc(a.rec_base + pbase + grb, glen);
        else if (priv) d = make_desc(a.rec_base + pbase + pslot, plen);
        else d = make_desc(L0 + lb, (repc && dc > 1u) ? dc : 0u, lpos - lb);
        if (!(DBG && (a.debug & 8))) {
            a.run_x[a.run_base + ebase + g] = d ? aid : RUN_X_EMPTY;
            a.run_desc[a.run_base + ebase + g] = d;
        }
    }
    wave_lds_sync();
}

// One launch over the sessions in memory order (as k_expand_fused): 64 sessions per wave round, sorted with ballots
// into three window size classes (n <= 8 / 16 / 32 -> 8 / 16 / 32 lanes per window) in a wave-private LDS task list.
template <bool TIME, bool DBG>
__global__ __launch_bounds__(256, 4) void k_expand_lists(ExpandArgs a, int64_t n_sess) {
    __shared__ uint4 s_ev[4][64];
    __shared__ uint4 s_ta[4][64];       // task: wstart lo, wstart hi | n << 16, pair_base lo, hi
    __shared__ uint2 s_tb[4][64];       //       ev_base lo, hi
    const int wv = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    uint4* ev = s_ev[wv];
    uint4* ta = s_ta[wv];
    uint2* tb = s_tb[wv];
    const int64_t n_tiles = (n_sess + 63) / 64;
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    struct TilePre { int64_t lo, hi; uint64_t pb, eb; };
    auto fetch_tile = [&](int64_t tile) {
        TilePre q;
        q.lo = q.hi = 0; q.pb = q.eb = 0;
        const int64_t s = tile * 64 + lane;
        if (tile < n_tiles && s < n_sess) {
            q.lo = a.sess_off[s]; q.hi = a.sess_off[s + 1];
            q.pb = a.pair_base[s]; q.eb = a.ev_base[s];
        }
        return q;
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    TilePre cur = fetch_tile(tile);
    for (; tile < n_tiles; tile += tile_stride) {
        const TilePre nxt = fetch_tile(tile + tile_stride);
        const int64_t len = cur.hi - cur.lo;
        const int n = (int)(len < a.window ? len : a.window);
        const int64_t wstart = cur.hi - n;
        const int c3 = n >= 2 ? (n <= 8 ? 0 : (n <= 16 ? 1 : 2)) : 3;
        int cnt[3], base[3];
        int below = 0, acc = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const uint64_t m = __ballot(c3 == q);
            cnt[q] = __popcll(m);
            base[q] = acc;
            if (c3 == q) below = acc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            acc += cnt[q];
        }
        if (c3 < 3) {
            ta[below] = make_uint4((uint32_t)wstart, (uint32_t)((uint64_t)wstart >> 32) | ((uint32_t)n << 16), (uint32_t)cur.pb, (uint32_t)(cur.pb >> 32));
            tb[below] = make_uint2((uint32_t)cur.eb, (uint32_t)(cur.eb >> 32));
        }
        wave_lds_sync();
        auto run_class = [&](auto gtag, auto gntag, int q, const TaskPre& first) {
            constexpr int G = decltype(gtag)::value;
            constexpr int GN = decltype(gntag)::value;
            constexpr int WPW = 64 / G;
            const int b0 = base[q], c = cnt[q];
            const int bn = q < 2 ? base[q < 2 ? q + 1 : 2] : 0, cn = q < 2 ? cnt[q < 2 ? q + 1 : 2] : 0;
            TaskPre tp = first;
            for (int t0 = 0; t0 < c; t0 += WPW) {
                TaskPre nx;
                if (t0 + WPW < c) nx = fetch_task<G, true>(a, ta, tb, b0, c, t0 + WPW, lane);
                else nx = fetch_task<GN, true>(a, ta, tb, bn, cn, 0, lane);
                expand_task_lists<G, TIME, DBG>(a, ev, lane, tp);
                tp = nx;
            }
            if (c == 0) tp = fetch_task<GN, true>(a, ta, tb, bn, cn, 0, lane);
            return tp;
        };
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        using I32 = std::integral_constant<int, 32>;
        TaskPre tp = fetch_task<8, true>(a, ta, tb, base[0], cnt[0], 0, lane);
        tp = run_class(I8{}, I16{}, 0, tp);
        tp = run_class(I16{}, I32{}, 1, tp);
        tp = run_class(I32{}, I32{}, 2, tp);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------
// index: histogram / scatter of runs by aid_x, work items
// ---------------------------------------------------------------------------
constexpr uint64_t REC_PAD = 64;                      // words past the last record slot that may be READ (for_each_record_seg: lanes without a record)
constexpr int CNT_REC_BITS = 36;                      // cnt64[x] = runs << 36 | records
constexpr uint64_t CNT_REC_MASK = (1ull << CNT_REC_BITS) - 1;

// One returning 64-bit atomic per run: it accumulates (runs, records) of aid_x AND hands the run its rank
// inside the aid (the previous run count), so the scatter below needs no second atomic.
__global__ void k_hist_runs(const uint32_t* run_x, const uint64_t* run_desc, int64_t n_slots, uint64_t* cnt64,
                            uint32_t* run_rank, uint32_t n_aids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = run_desc[i];
        if (desc_len(d)) {
            const uint32_t x = run_x[i];
            if (x < n_aids) {
                const unsigned long long old =
                    atomicAdd((unsigned long long*)&cnt64[x], (unsigned long long)((1ull << CNT_REC_BITS) | desc_pairs(d)));
                run_rank[i] = (uint32_t)(old >> CNT_REC_BITS);
            }
        }
    }
}

// ---- bucketed index: the same (cnt64, run_start, sorted_desc) without one global atomic per run -------------
// The histogram / scatter pair above pays one returning memory-side atomic and two random 8-byte accesses per run.
// Here the non-empty runs are first split into buckets of 2^SH consecutive aids (count -> scan -> scatter with one
// cursor bump per (chunk, bucket)), then one workgroup per bucket counts and places its runs with LDS atomics
// only; a bucket's slice of run_start / sorted_desc is a window of a few hundred KB, so those writes combine in L2.
constexpr int BKT_THREADS = 1024;
constexpr int BKT_CHUNK = 16384;                      // run slots per pass-1 work item (staged in LDS: 128 KB of 8-byte records)
constexpr int BKT_MAX_NB = 2048;                      // buckets (LDS histogram / offsets / cursors of pass 1)
constexpr int BKT_MAX_SH = 13;                        // aids per bucket <= 8192: 13 bits of the bucketed record, 96 KB of LDS in k_bkt_fused

// Bucketed run record (8 bytes; the bucket is implied by the position): len - 1 (5 bits; only non-empty runs travel) |
// first record slot << 5 (40 bits) | sp << 45 (6 bits) | (aid_x & (bucket size - 1)) << 51 (13 bits)
__device__ __forceinline__ uint64_t bkt_pack(uint64_t d, uint32_t xlow) {
    return (uint64_t)(desc_len(d) - 1u) | (desc_slot(d) << 5) | ((uint64_t)desc_sp(d) << 45) | ((uint64_t)xlow << 51);
}
__device__ __forceinline__ uint64_t bkt_desc(uint64_t r) {
    return make_desc((r >> 5) & DESC_SLOT_MASK, ((uint32_t)r & 31u) + 1u, (uint32_t)(r >> 45) & 63u);
}
__device__ __forceinline__ uint32_t bkt_xlow(uint64_t r) { return (uint32_t)(r >> 51); }

struct BktArgs {
    const uint32_t* run_x;
    const uint64_t* run_desc;
    int64_t n_slots;
    uint32_t n_aids;
    int sh;                        // aids per bucket = 1 << sh
    uint32_t nb;
    uint32_t* bcnt_blk;            // [nb][split grid] runs per (bucket, split workgroup)
    const uint64_t* bscan;         // [nb * split grid + 1] its bucket-major exclusive scan: bucket b starts at bscan[b * bstride]
    uint32_t bstride;              // = split grid
    uint64_t* tmp;                 // bucketed run records
    uint64_t* cnt64;
    const uint64_t* run_start;
    uint64_t* sorted_desc;
};

// Split without global atomics: workgroup w always takes the same range of consecutive chunks, so the count pass leaves ONE counter
// per (bucket, workgroup) and their bucket-major scan is every workgroup's private, contiguous piece of every bucket.
//   count  : s_cnt = histogram over all the workgroup's chunks (only run_x is read: empty runs carry RUN_X_EMPTY)
//   scatter: a chunk of 16 k run slots is SORTED BY BUCKET IN LDS (histogram with returning LDS atomics = rank inside the
//            (chunk, bucket) piece, block scan = the piece's place in the stage) and each piece is copied by one wave to the
//            workgroup's cursor of that bucket: consecutive lanes, consecutive addresses, pieces of ~70 records (~600 bytes)
//            at OTTO shape. (Round 2 stored every run straight to its cursor, 16 bytes each: a 32-byte sector of HBM
//            traffic per run, 4.1 GB written for 2.05 GB.)
template <bool SCATTER>
__global__ __launch_bounds__(BKT_THREADS) void k_bkt_split(BktArgs a) {
    __shared__ uint32_t s_cnt[BKT_MAX_NB];                       // count: histogram over the chunks; scatter: the workgroup's cursors
    __shared__ uint32_t s_hist[SCATTER ? BKT_MAX_NB : 1];        // runs of this chunk per bucket
    __shared__ uint32_t s_loc[SCATTER ? BKT_MAX_NB : 1];         // first stage position of the bucket's piece
    __shared__ uint64_t s_stage[SCATTER ? BKT_CHUNK : 1];
    __shared__ uint32_t s_sc[BKT_THREADS / 64 + 1];
    const int64_t n_chunks = (a.n_slots + BKT_CHUNK - 1) / BKT_CHUNK;
    constexpr int PER = BKT_CHUNK / BKT_THREADS;
    constexpr int BPT = BKT_MAX_NB / BKT_THREADS;                // buckets per thread in the scan
    for (uint32_t b = threadIdx.x; b < a.nb; b += BKT_THREADS)
        s_cnt[b] = SCATTER ? (uint32_t)a.bscan[(size_t)b * gridDim.x + blockIdx.x] : 0u;      // absolute position (run slots < 2^32)
    __syncthreads();
    const int64_t per_wg = (n_chunks + gridDim.x - 1) / gridDim.x;
    const int64_t ch_end = min(n_chunks, (int64_t)(blockIdx.x + 1) * per_wg);
    const uint32_t xmask = (1u << a.sh) - 1u;
    for (int64_t ch = (int64_t)blockIdx.x * per_wg; ch < ch_end; ++ch) {
        const int64_t i0 = ch * BKT_CHUNK + threadIdx.x;
        uint32_t xs[PER];
        uint64_t ds[SCATTER ? PER : 1];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = i0 + (int64_t)u * BKT_THREADS;
            uint32_t x = RUN_X_EMPTY;
            if (i < a.n_slots) {
                x = a.run_x[i];
                if (SCATTER) ds[u] = a.run_desc[i];       // requested together with run_x: one round trip per chunk, not two
            }
            if (x >= a.n_aids) x = RUN_X_EMPTY;
            xs[u] = x;
        }
        if (!SCATTER) {
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (xs[u] != RUN_X_EMPTY) atomicAdd(&s_cnt[xs[u] >> a.sh], 1u);
            continue;
        }
        for (uint32_t b = threadIdx.x; b < a.nb; b += BKT_THREADS) s_hist[SCATTER ? b : 0] = 0;
        __syncthreads();
        uint32_t rk[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            rk[u] = 0;
            if (xs[u] != RUN_X_EMPTY) rk[u] = atomicAdd(&s_hist[SCATTER ? xs[u] >> a.sh : 0], 1u);
        }
        __syncthreads();
        {   // exclusive scan of the chunk's histogram: thread t owns buckets t * BPT .. + BPT
            uint32_t h[BPT], mine = 0;
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const uint32_t b = threadIdx.x * BPT + q;
                h[q] = b < a.nb ? s_hist[SCATTER ? b : 0] : 0u;
                mine += h[q];
            }
            uint32_t tot;
            uint32_t run = block_excl_scan<uint32_t, BKT_THREADS>(mine, s_sc, &tot);
#pragma unroll
            for (int q = 0; q < BPT; ++q) {
                const uint32_t b = threadIdx.x * BPT + q;
                if (b < a.nb) s_loc[SCATTER ? b : 0] = run;
                run += h[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (xs[u] != RUN_X_EMPTY)
                s_stage[SCATTER ? s_loc[SCATTER ? xs[u] >> a.sh : 0] + rk[u] : 0] = bkt_pack(ds[SCATTER ? u : 0], xs[u] & xmask);
        __syncthreads();
        // one wave per piece: stage -> the workgroup's cursor of the bucket
        const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        for (uint32_t b = w; b < a.nb; b += BKT_THREADS / 64) {
            const uint32_t n = s_hist[SCATTER ? b : 0];
            if (n == 0) continue;
            const uint32_t src = s_loc[SCATTER ? b : 0], dst = s_cnt[b];
            for (uint32_t l = lane; l < n; l += 64u) a.tmp[dst + l] = s_stage[SCATTER ? src + l : 0];
            if (lane == 0) s_cnt[b] = dst + n;
        }
        __syncthreads();
    }
    if (!SCATTER) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < a.nb; b += BKT_THREADS) a.bcnt_blk[(size_t)b * gridDim.x + blockIdx.x] = s_cnt[b];
    }
}

// count + scan + place of one bucket in ONE workgroup: the first run of aid x is the bucket's first run (bstart[b], known from
// the split) + the runs of the bucket's aids before x, so the prefix is a block scan in LDS; the second walk over the bucket's
// runs (4.6 MB of 8-byte records at OTTO shape) is served by the Infinity Cache.
__global__ __launch_bounds__(BKT_THREADS) void k_bkt_fused(BktArgs a, uint64_t* run_start_out) {
    extern __shared__ unsigned long long s_dyn[];          // [1 << sh]: runs << 36 | records, then start positions
    uint32_t* s_cur = reinterpret_cast<uint32_t*>(s_dyn + ((size_t)1 << a.sh));   // [1 << sh] cursors
    __shared__ uint32_t s_sc[BKT_THREADS / 64 + 1];
    const uint32_t ab = 1u << a.sh;
    const uint32_t per = ab / BKT_THREADS > 0 ? ab / BKT_THREADS : 1u;
    for (uint32_t b = blockIdx.x; b < a.nb; b += gridDim.x) {
        const uint32_t x0 = b << a.sh;
        for (uint32_t i = threadIdx.x; i < ab; i += BKT_THREADS) { s_dyn[i] = 0; s_cur[i] = 0; }
        __syncthreads();
        const uint64_t e0 = a.bscan[(size_t)b * a.bstride], e1 = a.bscan[(size_t)(b + 1) * a.bstride];
        for (uint64_t i0 = e0 + threadIdx.x; i0 < e1; i0 += 4 * BKT_THREADS) {
            uint64_t r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + (uint64_t)u * BKT_THREADS;
                r[u] = i < e1 ? a.tmp[i] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r[u] != ~0ull) atomicAdd(&s_dyn[bkt_xlow(r[u])], (1ull << CNT_REC_BITS) | desc_pairs(bkt_desc(r[u])));
        }
        __syncthreads();
        // counts out, exclusive scan of the run counts over the bucket's aids (thread t: aids t * per .. + per)
        uint32_t mine = 0;
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t i = threadIdx.x * per + q;
            if (i < ab) {
                const unsigned long long c64 = s_dyn[i];
                if (x0 + i < a.n_aids) a.cnt64[x0 + i] = c64;
                mine += (uint32_t)(c64 >> CNT_REC_BITS);
            }
        }
        uint32_t tot;
        uint32_t run = block_excl_scan<uint32_t, BKT_THREADS>(mine, s_sc, &tot);
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t i = threadIdx.x * per + q;
            if (i < ab) {
                const uint32_t n = (uint32_t)(s_dyn[i] >> CNT_REC_BITS);
                s_dyn[i] = e0 + run;
                if (x0 + i < a.n_aids) run_start_out[x0 + i] = e0 + run;
                run += n;
            }
        }
        if (b == a.nb - 1 && threadIdx.x == 0) run_start_out[a.n_aids] = e1;
        __syncthreads();
        for (uint64_t i0 = e0 + threadIdx.x; i0 < e1; i0 += 4 * BKT_THREADS) {
            uint64_t r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + (uint64_t)u * BKT_THREADS;
                r[u] = i < e1 ? a.tmp[i] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r[u] != ~0ull) a.sorted_desc[s_dyn[bkt_xlow(r[u])] + atomicAdd(&s_cur[bkt_xlow(r[u])], 1u)] = bkt_desc(r[u]);
        }
        __syncthreads();
    }
}

struct BktCount {
    const uint32_t* c;
    __device__ uint64_t operator()(int64_t i) const { return c[i]; }
};

struct RunCount {
    const uint64_t* cnt64;
    __device__ uint64_t operator()(int64_t x) const { return cnt64[x] >> CNT_REC_BITS; }
};
struct RecCount {
    const uint64_t* cnt64;
    __device__ uint64_t operator()(int64_t x) const { return cnt64[x] & CNT_REC_MASK; }
};

__global__ void k_scatter_runs(const uint32_t* run_x, const uint64_t* run_desc, const uint32_t* run_rank, int64_t n_slots,
                               const uint64_t* run_start, uint64_t* sorted_desc, uint32_t n_aids) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = run_desc[i];
        if (desc_len(d)) {
            const uint32_t x = run_x[i];
            if (x < n_aids) sorted_desc[run_start[x] + run_rank[i]] = d;
        }
    }
}

// size bins of the reduce kernel (table slots / threads / max records that are guaranteed to fit)
constexpr int S_LOG2T = 9, S_THREADS = 64, S_CAP = 384;
constexpr int M_LOG2T = 12, M_THREADS = 256, M_CAP = 3072;
constexpr int L_LOG2T = 13, L_THREADS = 1024, L_CAP = 6144;

// Heavy aids come in two table layouts. An (x, y) pair gains at most one record per session that holds x, so a
// counter of aid x never exceeds runs(x): heavy aids with fewer than 4096 runs (82 % of the heavy pair mass at OTTO
// shape) are safe in the PACKED layout (12-bit counters) -- twice the slots in the same LDS (2^14), partitions of
// twice the records, half the work items. The others, and the time group, keep the wide layout (2^13 slots).
constexpr int LP_LOG2T = 14;
constexpr uint64_t PACKED_MAX_RUNS = 4096;
// Layouts ("modes") of the heavy bin:  0: wide, 2^13 slots, partitions of l_cap records;
//   1: packed, 2^14 slots, 16 waves, one workgroup per CU: aids of l_cap < n <= 2 * l_cap records as ONE item (no partition pass);
//   2: packed, 2^13 slots, 8 waves, two workgroups per CU: the whole aid (n <= l_cap) or partitions of l_cap records
//      (n > 2 * l_cap). Partitioned aids were in layout 1 with partitions of 2 * l_cap at first: half the items, but one
//      workgroup per CU exposes every barrier, and with twice the partitions three of four items instead of one of two
//      take the one-pass guess path (reduce L 12.5 -> 12.1 ms, partition 4.0 -> 3.7 ms; option packed_heavy = 1 is that rule).
//   (Smaller tables for the small aids of the M bin were measured and did not pay.)
__device__ __forceinline__ int heavy_mode(uint64_t c64, int allow_packed, uint32_t l_cap) {
    if (!allow_packed || (c64 >> CNT_REC_BITS) >= PACKED_MAX_RUNS) return 0;
    const uint64_t n = c64 & CNT_REC_MASK;
    if (allow_packed == 2) return (n <= (uint64_t)l_cap || n > 2ull * l_cap) ? 2 : 1;   // partitioned aids: 2^13 tables, partitions of l_cap
    return n <= (uint64_t)l_cap ? 2 : 1;
}
__device__ __forceinline__ int heavy_log2t(uint64_t c64, int allow_packed, uint32_t l_cap) {
    return heavy_mode(c64, allow_packed, l_cap) == 1 ? LP_LOG2T : L_LOG2T;
}

// log2 of the number of hash partitions of a heavy aid (records n, layout from its run count)
__device__ __forceinline__ int l_log2r(uint64_t c64, int boost, uint32_t l_cap, int allow_packed) {
    const uint64_t n = c64 & CNT_REC_MASK;
    const uint64_t cap = heavy_mode(c64, allow_packed, l_cap) == 1 ? 2ull * l_cap : (uint64_t)l_cap;
    uint64_t parts = (n + cap - 1) / cap;
    int lg = 0;
    while ((1ull << lg) < parts) ++lg;
    lg += boost;
    const int maxlg = 32 - heavy_log2t(c64, allow_packed, l_cap);
    return lg > maxlg ? maxlg : lg;
}

struct ItemCount {   // number of work items aid x contributes to bin `bin`
    const uint64_t* cnt64;
    const uint8_t* boost;
    const uint32_t* flag;
    int bin;
    int only_flagged;
    uint32_t l_cap;
    int allow_packed;
    int mode;                      // bins 1 and 2: -1 every aid of the bin, else only the aids of this tier / layout
    __device__ uint64_t operator()(int64_t x) const {
        const uint64_t n = cnt64[x] & CNT_REC_MASK;
        if (n == 0) return 0;
        if (only_flagged && !flag[x]) return 0;
        const int b = n <= (uint64_t)S_CAP ? 0 : (n <= (uint64_t)M_CAP ? 1 : 2);
        if (b != bin) return 0;
        if (bin == 0) return 1ull;
        if (bin == 1) return 1ull;
        if (mode >= 0 && heavy_mode(cnt64[x], allow_packed, l_cap) != mode) return 0;
        return 1ull << l_log2r(cnt64[x], boost[x], l_cap, allow_packed);
    }
};

struct BinRecs {     // expanded pairs of aid x if it falls in bin `bin`
    const uint64_t* cnt64;
    int bin;
    __device__ uint64_t operator()(int64_t x) const {
        const uint64_t n = cnt64[x] & CNT_REC_MASK;
        const int b = n <= (uint64_t)S_CAP ? 0 : (n <= (uint64_t)M_CAP ? 1 : 2);
        return b == bin ? n : 0;
    }
};

struct BinRuns {     // runs of aid x if it falls in bin `bin`
    const uint64_t* cnt64;
    int bin;
    __device__ uint64_t operator()(int64_t x) const {
        const uint64_t n = cnt64[x] & CNT_REC_MASK;
        const int b = n <= (uint64_t)S_CAP ? 0 : (n <= (uint64_t)M_CAP ? 1 : 2);
        return b == bin ? (cnt64[x] >> CNT_REC_BITS) : 0;
    }
};

struct ItemAny {     // 1 if aid x has work items in this bin
    ItemCount f;
    __device__ uint64_t operator()(int64_t x) const { return f(x) ? 1ull : 0ull; }
};

// Processing order of one layout's heavy aids: the first partition of every aid ("pilot"), then all other
// partitions. The pilots leave their top-k threshold in tau[x], which lets the aid's other partitions finish in one
// table pass. f counts this layout's items; item_start (all heavy aids) gives the item index.
__global__ void k_fill_order(ItemCount f, uint32_t n_aids, const uint64_t* item_start, const uint64_t* mode_start,
                             const uint64_t* aid_rank, uint64_t n_pilots, uint32_t* order) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_aids) return;
    const uint64_t c = f((int64_t)x);
    if (!c) return;
    const uint64_t s = item_start[x], sm = mode_start[x], r = aid_rank[x];
    order[r] = (uint32_t)s;
    for (uint64_t p = 1; p < c; ++p) order[n_pilots + sm + p - r - 1] = (uint32_t)(s + p);
}

// item = x | part << 26 | log2R << 50
__global__ void k_fill_items(ItemCount f, uint32_t n_aids, const uint64_t* item_start, uint64_t* items) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_aids) return;
    const uint64_t c = f((int64_t)x);
    if (!c) return;
    const uint64_t s = item_start[x];
    uint64_t lg = 0;
    while ((1ull << lg) < c) ++lg;
    for (uint64_t p = 0; p < c; ++p) items[s + p] = (uint64_t)x | (p << 26) | (lg << 50);
}

// ---------------------------------------------------------------------------
// K2: per-item LDS hash reduce + top-k
// ---------------------------------------------------------------------------
constexpr int MAX_K = 32;
constexpr int MAX_KINDS = 4;
#ifndef OTTO_PK
#define OTTO_PK 3
#endif
constexpr int PK = OTTO_PK;                    // kinds reduced per pass over the records
constexpr int PART_CHUNK_RUNS = 256;    // runs per partition-pass work item (<= 256 * 31 records: fits the LDS stage)
constexpr int PART_STAGE = 8192;        // records staged in LDS per chunk
constexpr int PART_STAGE_LOG2R = 9;     // staged (coalesced) scatter up to 512 partitions (aids of <= 3.1 M pairs), direct scatter above; 10 would cost the 4th workgroup per CU (LDS)
constexpr int PART_LDS_LOG2R = 12;      // partitions whose histogram / cursors fit LDS

struct ReduceArgs {
    const uint64_t* items;
    const uint32_t* order;         // processing order (dequeue index -> item index), null = identity
    uint32_t n_items;
    uint32_t n_work;               // entries of `order` (= n_items without an order)
    int allow_packed;              // heavy aids with < 4096 runs use the packed layout (not in the time group)
    const uint64_t* cnt64;         // [n_aids] runs << 36 | records
    const uint64_t* run_start;     // [n_aids+1]
    const uint64_t* sorted_desc;
    const uint32_t* rec;
    const uint32_t* tw;
    const uint64_t* pstart;        // [n_items+1] partition buckets of L items with R > 1 (null: filter mode)
    const uint32_t* pcursor;       // [n_items] records the scatter put into each bucket
    const uint32_t* prec;
    const uint32_t* ptw;
    int group;
    int nk;                        // kinds selected in this pass (<= MAX_KINDS)
    int k;
    int chan_shift;                // FILTER: first filter bit of this pass
    uint32_t coef[MAX_KINDS][3];   // unit weight of kind j = sum_c v[c] * coef[j][c]
    uint32_t n_aids;
    int kind_base;                 // first output kind index of this pass
    uint32_t* out_y;               // [kinds][n_aids][k]
    uint64_t* out_w;
    int32_t* out_n;                // [kinds][n_aids]
    uint32_t* part_y;              // [item][nk][k] partial top-k of L items with R > 1
    uint64_t* part_w;
    uint32_t* flag;                // [n_aids] overflow -> redo with more partitions
    uint8_t* boost;
    uint32_t* ovf_count;
    uint32_t* work_counter;
    uint64_t* tau_w;               // [PK][n_aids] threshold guess of a heavy aid's partitions: a lower bound of the 32nd
    uint32_t* tau_y;               //   best key of a partition already reduced (0 = none yet); tau_y only for GROUP_TIME
    uint32_t l_cap;                // records per L partition the item lists were sized for
    int hot_ok;                    // GROUP_TYPE: every kind ranks a key made of ONE click record below every other key (see k_reduce)
    int debug_skip;                // diagnostics only (wrong results): 1 skip gather, 2 skip top-k, 4 skip table init, 8 skip inserts
#ifdef OTTO_PHASE_PROF
    unsigned long long* prof;      // [16] summed shader-clock ticks of thread 0 per phase + path counters
#endif
};

// candidate keys and the wave-level top-k primitives live in topk.h; the covisitation-specific key encodings:
__device__ __forceinline__ void kmake(KeyN& k, uint64_t unit_w, uint64_t, uint32_t y) { k.c = unit_w ? (unit_w << REC_AID_BITS) | (uint64_t)(REC_AID_MASK - y) : 0; }
__device__ __forceinline__ void kmake(KeyW& k, uint64_t, uint64_t q16_w, uint32_t y) { k.w = q16_w; k.y = q16_w ? y : KEY_EMPTY; }
__device__ __forceinline__ uint64_t kweight(KeyN k) { return (k.c >> REC_AID_BITS) * 65536ull; }
__device__ __forceinline__ uint64_t kweight(KeyW k) { return k.w; }
__device__ __forceinline__ uint32_t kaid(KeyN k) { return REC_AID_MASK - (uint32_t)(k.c & REC_AID_MASK); }
__device__ __forceinline__ uint32_t kaid(KeyW k) { return k.y; }
__device__ __forceinline__ void ktau_store(KeyN k, uint64_t* w, uint32_t*, size_t o) { w[o] = k.c; }
__device__ __forceinline__ void ktau_store(KeyW k, uint64_t* w, uint32_t* y, size_t o) { w[o] = k.w; y[o] = k.y; }

// NK independent 64-lane sorts, networks interleaved stage by stage (NK dependency chains in flight)
template <int NK, typename K>
__device__ __forceinline__ void wave_bitonic_sort_multi(K (&v)[NK]) {
    const unsigned l = lane_id();
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            const bool keep_better = ((l & jj) == 0) == ((l & kk) == 0);
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const K o = kshfl_xor(v[j], jj);
                if (keep_better ? kbetter(o, v[j]) : kbetter(v[j], o)) v[j] = o;
            }
        }
    }
}

// NK independent selections at once (the kinds of one pass): the sorting networks are interleaved stage by
// stage so each wave has NK independent dependency chains in flight (a single chain is latency-bound).
template <int MPL, int NK, typename K>
__device__ __forceinline__ void wave_topk_select_multi(K (&c)[NK][MPL], int nk, int k, K (&best)[NK]) {
    const unsigned l = lane_id();
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        K lb = c[j][0];
        int bi = 0;
#pragma unroll
        for (int i = 1; i < MPL; ++i)
            if (kbetter(c[j][i], lb)) { lb = c[j][i]; bi = i; }
#pragma unroll
        for (int i = 0; i < MPL; ++i)
            if (i == bi) kclear(c[j][i]);
        best[j] = lb;
    }
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1) {
            const bool keep_better = ((l & jj) == 0) == ((l & kk) == 0);
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const K o = kshfl_xor(best[j], jj);
                if (keep_better ? kbetter(o, best[j]) : kbetter(best[j], o)) best[j] = o;
            }
        }
    }
    if (MPL > 1) {
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            if (j >= nk) continue;
            for (;;) {
                const K thr = kshfl(best[j], k - 1);
                K lb;
                kclear(lb);
                int bi = -1;
#pragma unroll
                for (int i = 0; i < MPL; ++i)
                    if (kvalid(c[j][i]) && (bi < 0 || kbetter(c[j][i], lb))) { lb = c[j][i]; bi = i; }
                const bool q = bi >= 0 && kbetter(lb, thr);
                if (__ballot(q) == 0) break;
#pragma unroll
                for (int i = 0; i < MPL; ++i)
                    if (i == bi) kclear(c[j][i]);
                if (!q) kclear(lb);
                wave_topk_push(best[j], lb, k);
            }
        }
    }
}

// Visit every record of the runs [r0, r1). Runs are dealt round-robin to the NW waves of the workgroup
// (wave w owns runs r0 + w, r0 + w + NW, ...): a wave fetches 64 of its run descriptors with one load, then
// walks them two at a time (one per 32-lane half) with GATHER_U record loads in flight per lane, so an item
// of a few dozen runs costs two or three dependent memory round trips instead of one per run.
// fb(rc, slot, ok) receives GATHER_U records at a time (arrays; ok[u] = lane holds a record), so that the consumer can
// put GATHER_U independent LDS operations in flight before it looks at any result.
template <int NW, int GATHER_U, typename FB>
__device__ __forceinline__ void for_each_record_batch(const uint64_t* sorted_desc, const uint32_t* rec, uint64_t r0, uint64_t r1,
                                                      int wid, FB fb) {
    const unsigned lane = lane_id();
    const uint32_t l = lane & 31u;
    const int half = (int)(lane >> 5);
    if (r0 >= r1) return;
    auto load_desc = [&](uint64_t cb) {
        const uint64_t mine = cb + (uint64_t)lane * NW + wid;
        return mine < r1 ? sorted_desc[mine] : 0ull;
    };
    uint64_t d = load_desc(r0);
    for (uint64_t cb = r0; cb < r1; cb += (uint64_t)NW * 64) {
        const uint64_t cbn = cb + (uint64_t)NW * 64;
        const uint64_t dn = cbn < r1 ? load_desc(cbn) : 0ull;              // next descriptors in flight
        const uint64_t left = r1 - cb;                                     // runs in this chunk (all waves)
        const int nrun = left >= (uint64_t)NW * 64 ? 64 : (int)((left + NW - 1 - wid) / NW);   // this wave's share
        uint32_t rcA[GATHER_U], rcB[GATHER_U];
        uint64_t slA[GATHER_U], slB[GATHER_U];
        bool okA[GATHER_U], okB[GATHER_U];
        auto issue = [&](int t, uint32_t (&rc)[GATHER_U], uint64_t (&sl)[GATHER_U], bool (&ok)[GATHER_U]) {
#pragma unroll
            for (int u = 0; u < GATHER_U; ++u) {
                const int src = t + 2 * u + half;
                const uint64_t dd = (uint64_t)__shfl((unsigned long long)d, src & 63, 64);
                const uint32_t sp = desc_sp(dd);
                const uint64_t s0 = desc_slot(dd);
                ok[u] = src < 64 && l < desc_len(dd) && l != sp;               // the run's own list entry is not a pair
                rc[u] = ok[u] ? rec[s0 + l] : 0u;
                sl[u] = s0 + (sp != DESC_SP_NONE ? sp : l);                      // where the record's time extra lives
            }
        };
        auto consume = [&](uint32_t (&rc)[GATHER_U], uint64_t (&sl)[GATHER_U], bool (&ok)[GATHER_U]) { fb(rc, sl, ok); };
        // two batches of 2*GATHER_U runs alternate: the loads of one are in flight while the other is consumed
        if (nrun > 0) issue(0, rcA, slA, okA);
        for (int t = 0; t < nrun; t += 4 * GATHER_U) {
            const bool hasB = t + 2 * GATHER_U < nrun;
            if (hasB) issue(t + 2 * GATHER_U, rcB, slB, okB);
            consume(rcA, slA, okA);
            if (t + 4 * GATHER_U < nrun) issue(t + 4 * GATHER_U, rcA, slA, okA);
            if (hasB) consume(rcB, slB, okB);
        }
        d = dn;
    }
}

// Same visit, packed 8 lanes per SEGMENT (8 consecutive records of a run): a run of len records is ceil(len / 8) segments,
// the segments of a wave's 64 runs are numbered by a wave scan and dealt to the eight 8-lane groups, so a wave-instruction
// carries up to 64 records whatever the run lengths are (two 32-lane runs per instruction fill ~40 % of the lanes at the
// OTTO run-length mix; the consumer's insert rounds -- its cost -- scale with the instructions, not the records).
// s_seg: 256 bytes of LDS private to the wave (segment -> lane that holds the run's descriptor | segment number << 6).
// NEED_SL: the consumer wants the record's slot (time channel lookup); otherwise the slot arrays are not kept
// d0p: the caller already holds the wave's first 64 descriptors (requested while it was busy with something else)
template <int NW, int GATHER_U, bool NEED_SL, typename FB>
__device__ __forceinline__ void for_each_record_seg(const uint64_t* sorted_desc, const uint32_t* rec, uint64_t r0, uint64_t r1,
                                                    int wid, uint8_t* s_seg, FB fb, const uint64_t* d0p = nullptr) {
    const unsigned lane = lane_id();
    const uint32_t g = lane >> 3, gl4 = (lane & 7u) << 2;
    if (r0 >= r1) return;
    auto load_desc = [&](uint64_t cb) {
        const uint64_t mine = cb + (uint64_t)lane * NW + wid;
        return mine < r1 ? sorted_desc[mine] : 0ull;
    };
    uint64_t d = d0p ? *d0p : load_desc(r0);
    for (uint64_t cb = r0; cb < r1; cb += (uint64_t)NW * 64) {
        const uint64_t cbn = cb + (uint64_t)NW * 64;
        const uint64_t dn = cbn < r1 ? load_desc(cbn) : 0ull;              // next descriptors in flight
        const uint32_t len = desc_len(d);
        const uint32_t segs = (len + 7u) >> 3;
        const uint32_t incl = wave_incl_scan(segs), excl = incl - segs;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (k < segs) s_seg[excl + k] = (uint8_t)((lane << 2) | k);      // descriptor lane << 2 | segment of the run (len <= 32: 4 segments)
        wave_lds_sync();
        const int nstep = (int)((total + 7u) >> 3);
        // what a record's lane needs of its run, in the two words it fetches from the descriptor lane: the ADDRESS of the
        // run's first record (48 bits) with len and sp in the 16 bits above it -- per record 2 field extracts, 1 mask and
        // 1 add instead of unpacking the descriptor (slot shift, masks, base add, offset add) 64 times per run chunk
        const uint64_t recb = (uint64_t)(uintptr_t)rec;
        const uint64_t pre = (recb + (desc_slot(d) << 2)) | ((uint64_t)desc_len(d) << 50) | ((uint64_t)desc_sp(d) << 58);   // len * 4, sp * 4 in bytes 6, 7
        const int plo = (int)(uint32_t)pre, phi = (int)(uint32_t)(pre >> 32);
        uint32_t rcA[GATHER_U], rcB[GATHER_U];
        uint64_t slA[NEED_SL ? GATHER_U : 1], slB[NEED_SL ? GATHER_U : 1];
        uint64_t okA[GATHER_U], okB[GATHER_U];                              // wave masks (SGPR pairs): lanes that hold a record
        // Branch-free on purpose -- one basic block, so the GATHER_U segment reads, the 2 x GATHER_U descriptor permutes and
        // the loads go out back to back instead of one dependent LDS round trip after the other: the segment byte is read
        // whether or not the step has a segment for this group (stale bytes name some lane of the wave, whose descriptor is a
        // real one or 0: the step bound masks the result), and a lane without a record still loads the word its (run, offset)
        // names -- at most 31 words past a run, inside the 64-word pad every rec allocation carries (REC_PAD; tw is only
        // read where a record is). The three conditions are compares straight into wave masks, combined on the scalar unit.
        constexpr int ICMP_NE = 33, ICMP_ULT = 36, ICMP_SLT = 40;
        auto issue = [&](int t, uint32_t (&rc)[GATHER_U], uint64_t (&sl)[NEED_SL ? GATHER_U : 1], uint64_t (&ok)[GATHER_U]) {
            const uint32_t qb = (uint32_t)t * 8u + g;
            const int lim = (int)total - 8 * t;                              // group g of step t + u has a segment: g < lim - 8 u
#pragma unroll
            for (int u = 0; u < GATHER_U; ++u) {
                const uint32_t r = (uint32_t)s_seg[qb + 8u * u];             // (t + u) * 8 + g < 256 + 8 * GATHER_U: the array is padded
                const int src = (int)(r & 0xFCu);                            // ds_bpermute: byte address = lane * 4
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, plo);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, phi);
                const uint32_t off4 = ((r & 3u) << 5) | gl4;                 // byte offset of the lane's record in the run
                const uint32_t len4 = (hi >> 16) & 0xFFu, sp4 = hi >> 24;
                const uint64_t base = ((uint64_t)(hi & 0xFFFFu) << 32) | lo;
                if (NEED_SL) sl[NEED_SL ? u : 0] = (base - recb + (sp4 != DESC_SP_NONE * 4u ? sp4 : off4)) >> 2;   // where the record's time extra lives
                ok[u] = __builtin_amdgcn_sicmp((int)g, lim - 8 * u, ICMP_SLT) & __builtin_amdgcn_uicmp(off4, len4, ICMP_ULT) &
                        __builtin_amdgcn_uicmp(off4, sp4, ICMP_NE);          // the run's own list entry is not a pair
                rc[u] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>((uintptr_t)(base + off4));
            }
        };
        if (nstep > 0) issue(0, rcA, slA, okA);
        for (int t = 0; t < nstep; t += 2 * GATHER_U) {
            const bool hasB = t + GATHER_U < nstep;
            if (hasB) issue(t + GATHER_U, rcB, slB, okB);
            fb(rcA, slA, okA);
            if (t + 2 * GATHER_U < nstep) issue(t + 2 * GATHER_U, rcA, slA, okA);
            if (hasB) fb(rcB, slB, okB);
        }
        wave_lds_sync();                                                   // s_seg is rewritten by the next chunk
        d = dn;
    }
}

template <int NW, int GATHER_U, typename F>
__device__ __forceinline__ void for_each_record(const uint64_t* sorted_desc, const uint32_t* rec, uint64_t r0, uint64_t r1,
                                                int wid, F f) {
    for_each_record_batch<NW, GATHER_U>(sorted_desc, rec, r0, r1, wid,
                                        [&](uint32_t (&rc)[GATHER_U], uint64_t (&sl)[GATHER_U], bool (&ok)[GATHER_U]) {
#pragma unroll
        for (int u = 0; u < GATHER_U; ++u)
            if (ok[u]) f(rc[u], sl[u]);
    });
}

__device__ __forceinline__ uint32_t rec_hash(uint32_t rc) { return (rc & REC_AID_MASK) * 0x9E3779B1u; }

// ---- partition pass for heavy aids (L items with R > 1): count, then scatter into per-partition buckets ----
struct PartArgs {
    const uint64_t* chunks;        // x | chunk << 26
    uint32_t n_chunks;
    const uint64_t* cnt64;
    const uint8_t* boost;
    const uint64_t* run_start;
    const uint64_t* sorted_desc;
    const uint32_t* rec;
    const uint32_t* tw;
    const uint64_t* litem_start;   // [n_aids+1] first L item of aid x
    uint32_t* pcount;              // [n_items_L]
    uint32_t* pcursor;
    const uint64_t* pstart;        // [n_items_L + 1]
    uint32_t* prec;
    uint32_t* ptw;
    uint32_t l_cap;
    int window_max;                // largest window (records per run <= window_max - 1)
    int allow_packed;
    uint32_t* flag;                // [n_aids] bucket overflow (capacity-sized buckets): the aid is redone with exact bucket sizes
    uint32_t* ovf_count;
    uint32_t* work;                // scatter pass: chunk dequeue counter (zeroed before the launch)
};

// Capacity of a partition bucket when the buckets are sized WITHOUT the count pass: twice the mean partition size + a
// margin. A partition can exceed it only through one hot aid_y (equal records share a partition); the scatter detects
// that, flags the aid, and the host redoes the flagged aids with counted buckets.
__host__ __device__ __forceinline__ uint64_t bucket_cap(uint64_t n_records, int lgR) {
    return lgR == 0 ? 0ull : 2ull * ((n_records + (1ull << lgR) - 1ull) >> lgR) + 256ull;
}
struct ItemCap {     // bucket capacity of L item i
    const uint64_t* items;
    const uint64_t* cnt64;
    __device__ uint64_t operator()(int64_t i) const {
        const uint64_t item = items[i];
        return bucket_cap(cnt64[item & REC_AID_MASK] & CNT_REC_MASK, (int)(item >> 50));
    }
};

// TW: the time channel travels with the records (GROUP_TIME): staged and scattered alongside
// A chunk is a chain of dependent memory round trips (chunk word -> the aid's counts and run range -> its first bucket and the
// run descriptors -> the records -> the bucket cursors) around ~13 records per thread: the SCATTER pass keeps the chain
// of the NEXT chunks in flight while the current one is gathered and scattered (chunk word two chunks ahead, the aid's
// words one chunk ahead, bucket base + descriptors from the middle of the current chunk). These prefetches go through the
// VECTOR memory path on purpose (an opaque zero in the address): vector loads retire in order under vmcnt, scalar loads
// share lgkmcnt with the LDS operations and would be waited for at the next LDS access.
template <bool SCATTER, bool TW>
__global__ __launch_bounds__(256) void k_partition(PartArgs a) {
    constexpr int NW = 4;
    constexpr int RL = 1 << (SCATTER ? PART_STAGE_LOG2R : PART_LDS_LOG2R);
    constexpr int WSTAGE = PART_STAGE / NW;            // stage region of one wave (its 64 runs x window_max records fit: `direct` below)
    __shared__ uint32_t s_cnt[RL];                     // histogram, then staging cursors
    __shared__ uint32_t s_delta[SCATTER ? RL : 1];     // (global bucket position - position in the stage) per partition
    __shared__ uint32_t s_stage[SCATTER ? PART_STAGE : 1];
    __shared__ uint32_t s_stage_tw[(SCATTER && TW) ? PART_STAGE : 1];
    __shared__ uint32_t s_full;                        // a capacity-sized bucket of this chunk's aid is full
    __shared__ uint8_t s_seg[NW * 256 + 32];           // gather: segment -> descriptor lane, per wave (+ pad: for_each_record_seg reads past a wave's last segment)
    const int wid = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    struct Chunk {                                     // everything a chunk needs before its records (uniform)
        uint32_t x; int lgR, pshift; uint64_t g0, rb, re; bool direct;
    };
    auto derive = [&](uint64_t ch, uint64_t c64, uint32_t boost, uint64_t rs0, uint64_t rs1, uint64_t g0) {
        Chunk k;
        k.x = (uint32_t)(ch & REC_AID_MASK);
        const uint64_t c = ch >> 26;
        k.lgR = l_log2r(c64, (int)boost, a.l_cap, a.allow_packed);
        k.pshift = 32 - heavy_log2t(c64, a.allow_packed, a.l_cap) - k.lgR;
        k.g0 = g0;
        k.rb = rs0 + c * PART_CHUNK_RUNS;
        k.re = k.rb + PART_CHUNK_RUNS;
        if (k.re > rs1) k.re = rs1;
        k.direct = k.lgR > (SCATTER ? PART_STAGE_LOG2R : PART_LDS_LOG2R) || (c64 & CNT_REC_MASK) >= (1ull << 31) ||
                   (SCATTER && a.window_max * PART_CHUNK_RUNS > PART_STAGE);
        return k;
    };
    // no LDS staging (giant aids): global cursor per record
    auto run_direct = [&](const Chunk& k) {
        const uint32_t pmask = (1u << k.lgR) - 1u;
        for_each_record<NW, 8>(a.sorted_desc, a.rec, k.rb, k.re, wid, [&](uint32_t rc, uint64_t slot) {
            const uint64_t g = k.g0 + ((rec_hash(rc) >> k.pshift) & pmask);
            if (!SCATTER) atomicAdd(&a.pcount[g], 1u);
            else {
                const uint64_t pos = a.pstart[g] + atomicAdd(&a.pcursor[g], 1u);
                if (pos >= a.pstart[g + 1]) {                      // capacity-sized bucket is full
                    if (atomicExch(&a.flag[k.x], 1u) == 0u) atomicAdd(a.ovf_count, 1u);
                } else {
                    a.prec[pos] = rc;
                    if (a.ptw) a.ptw[pos] = a.tw[slot];
                }
            }
        });
    };
    if constexpr (!SCATTER) {
        // count pass (retry rounds only): histogram per chunk in LDS, one global add per (chunk, partition)
        for (uint32_t ci = blockIdx.x; ci < a.n_chunks; ci += gridDim.x) {
            const uint64_t ch = a.chunks[ci];
            const uint32_t x = (uint32_t)(ch & REC_AID_MASK);
            const Chunk k = derive(ch, a.cnt64[x], a.boost[x], a.run_start[x], a.run_start[x + 1], a.litem_start[x]);
            if (k.direct) { run_direct(k); continue; }
            const uint32_t R = 1u << k.lgR, pmask = R - 1u;
            for (uint32_t p = threadIdx.x; p < R; p += 256) s_cnt[p] = 0;
            __syncthreads();
            for_each_record<NW, 8>(a.sorted_desc, a.rec, k.rb, k.re, wid, [&](uint32_t rc, uint64_t) {
                atomicAdd(&s_cnt[(rec_hash(rc) >> k.pshift) & pmask], 1u);
            });
            __syncthreads();
            for (uint32_t p = threadIdx.x; p < R; p += 256) {
                const uint32_t n = s_cnt[p];
                if (n) atomicAdd(&a.pcount[k.g0 + p], n);
            }
            __syncthreads();
        }
    } else {
        uint32_t vz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vz));     // opaque zero: keeps the prefetches below on the vector memory path
        auto uni64 = [&](uint64_t v) -> uint64_t {
            return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        };
        constexpr uint64_t NO_CHUNK = ~0ull;
        auto ld_chunk = [&](uint64_t ci) -> uint64_t { return ci < a.n_chunks ? a.chunks[ci + vz] : NO_CHUNK; };
        struct Aid { uint64_t c64, rs0, rs1, g0; uint32_t boost; };
        auto ld_aid = [&](uint64_t ch) {                // ch uniform
            Aid m{0, 0, 0, 0, 0};
            if (ch != NO_CHUNK) {
                const uint32_t x = (uint32_t)(ch & REC_AID_MASK) + vz;
                m.c64 = a.cnt64[x]; m.boost = a.boost[x]; m.rs0 = a.run_start[x]; m.rs1 = a.run_start[x + 1]; m.g0 = a.litem_start[x];
            }
            return m;
        };
        auto mk = [&](uint64_t ch, const Aid& m) { return derive(ch, uni64(m.c64), (uint32_t)__builtin_amdgcn_readfirstlane((int)m.boost), uni64(m.rs0), uni64(m.rs1), uni64(m.g0)); };
        auto ld_desc = [&](const Chunk& k) -> uint64_t {
            const uint64_t mine = k.rb + (uint64_t)lane * NW + wid;
            return mine < k.re ? a.sorted_desc[mine] : 0ull;
        };
        // chunks are dequeued dynamically, CQ per counter bump (thread 0; the index three chunks ahead is published through LDS
        // and read after the barriers of the chunk in between): a static deal left the workgroups' sums uneven
        __shared__ uint32_t s_ci[4];
        constexpr uint32_t CQ = 4;
        uint32_t pool = 0, pool_end = 0, pool_next = 0;
        auto take = [&]() {
            const uint32_t r = pool++;
            if (pool == pool_end) {
                pool = pool_next;
                pool_end = pool + CQ;
                pool_next = atomicAdd(a.work, CQ);
            }
            return r;
        };
        if (threadIdx.x == 0) {
            pool = atomicAdd(a.work, 2 * CQ);
            pool_end = pool + CQ;
            pool_next = pool + CQ;
            s_ci[2] = take(); s_ci[3] = take(); s_ci[0] = take();
        }
        __syncthreads();
        uint64_t ci = s_ci[2], ci1 = s_ci[3], ci2 = s_ci[0];
        __syncthreads();
        if (ci >= a.n_chunks) return;
        uint64_t ch_cur = uni64(ld_chunk(ci));
        uint64_t ch_nxt_v = ld_chunk(ci1);
        Chunk cur = mk(ch_cur, ld_aid(ch_cur));
        uint64_t xb_cur_v = a.pstart[cur.g0 + vz];
        uint64_t d_cur = ld_desc(cur);
        for (uint32_t it = 0; ci < a.n_chunks; ++it) {
            if (threadIdx.x == 0) s_ci[it & 1u] = take();   // the chunk after ci2: read at the end of this iteration
            const uint64_t ch_nxt = uni64(ch_nxt_v);     // requested one chunk ago
            const uint64_t ch_n2_v = ld_chunk(ci2);
            const Aid aid_nxt = ld_aid(ch_nxt);          // lands while this chunk's records are gathered
            Chunk nxt{};
            uint64_t xb_nxt_v = 0, d_nxt = 0;
            const uint32_t x = cur.x;
            const uint32_t R = 1u << cur.lgR, pmask = R - 1u;
            const int pshift = cur.pshift;
            const uint64_t g0 = cur.g0;
            auto stage3 = [&]() {                        // bucket base + run descriptors of the next chunk
                if (ch_nxt != NO_CHUNK) {
                    nxt = mk(ch_nxt, aid_nxt);
                    xb_nxt_v = a.pstart[nxt.g0 + vz];
                    d_nxt = ld_desc(nxt);
                }
            };
            if (cur.direct) {
                run_direct(cur);
                stage3();
            } else {
                for (uint32_t p = threadIdx.x; p < R; p += 256) s_cnt[p] = 0;
                if (threadIdx.x == 0) s_full = 0;
                __syncthreads();
                // ONE pass over the chunk's records: each wave appends its records to its own region of the LDS stage
                // (wave-private cursor: no atomics) while the per-partition histogram is taken
                uint32_t wn = 0;
                for_each_record_seg<NW, 4, TW>(a.sorted_desc, a.rec, cur.rb, cur.re, wid, s_seg + wid * 256, [&](uint32_t (&rc)[4], uint64_t (&sl)[TW ? 4 : 1], uint64_t (&ok)[4]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint64_t m = ok[u];
                        if (m == 0) continue;
                        if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                            const uint32_t pos = (uint32_t)wid * WSTAGE + wn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            s_stage[SCATTER ? pos : 0] = rc[u];
                            if (TW) s_stage_tw[(SCATTER && TW) ? pos : 0] = a.tw[sl[TW ? u : 0]];
                            atomicAdd(&s_cnt[(rec_hash(rc[u]) >> pshift) & pmask], 1u);
                        }
                        wn += (uint32_t)__popcll(m);
                    }
                }, &d_cur);
                __syncthreads();
                stage3();
                const uint64_t x_base = uni64(xb_cur_v);
                // one global cursor bump per (chunk, partition) reserves the chunk's piece of every bucket
                for (uint32_t p = threadIdx.x; p < R; p += 256) {
                    const uint32_t n = s_cnt[p];
                    uint32_t gpos = 0;
                    if (n) {
                        const uint32_t old = atomicAdd(&a.pcursor[g0 + p], n);
                        gpos = (uint32_t)(a.pstart[g0 + p] - x_base) + old;
                        if ((uint64_t)old + n > a.pstart[g0 + p + 1] - a.pstart[g0 + p]) s_full = 1u;   // full: nothing of this chunk is written
                    }
                    s_delta[SCATTER ? p : 0] = gpos;
                    s_cnt[p] = 0;
                }
                __syncthreads();
                if (s_full) {                                              // block-uniform
                    if (threadIdx.x == 0 && atomicExch(&a.flag[x], 1u) == 0u) atomicAdd(a.ovf_count, 1u);
                } else {
                    // a partition's records of this chunk go to one contiguous piece of its bucket (order inside is free)
                    for (uint32_t i = lane; i < wn; i += 64) {
                        const uint32_t si = (uint32_t)wid * WSTAGE + i;
                        const uint32_t rc = s_stage[SCATTER ? si : 0];
                        const uint32_t p = (rec_hash(rc) >> pshift) & pmask;
                        const uint64_t o = x_base + (uint64_t)(s_delta[SCATTER ? p : 0] + atomicAdd(&s_cnt[p], 1u));
                        a.prec[o] = rc;
                        if (TW) a.ptw[o] = s_stage_tw[(SCATTER && TW) ? si : 0];
                    }
                }
            }
            __syncthreads();
            ci = ci1; ci1 = ci2; ci2 = s_ci[it & 1u];
            cur = nxt;
            xb_cur_v = xb_nxt_v;
            d_cur = d_nxt;
            ch_nxt_v = ch_n2_v;
        }
    }
}

struct PCount {
    const uint32_t* pcount;
    __device__ uint64_t operator()(int64_t i) const { return pcount[i]; }
};

struct ChunkCount {   // partition-pass work items of aid x: ceil(runs / PART_CHUNK_RUNS) if it is an L aid with R > 1
    const uint64_t* cnt64;
    const uint8_t* boost;
    const uint32_t* flag;
    int only_flagged;
    uint32_t l_cap;
    int allow_packed;
    __device__ uint64_t operator()(int64_t x) const {
        const uint64_t n = cnt64[x] & CNT_REC_MASK;
        if (n <= (uint64_t)M_CAP) return 0;
        if (only_flagged && !flag[x]) return 0;
        if (l_log2r(cnt64[x], boost[x], l_cap, allow_packed) == 0) return 0;
        return ((cnt64[x] >> CNT_REC_BITS) + PART_CHUNK_RUNS - 1) / PART_CHUNK_RUNS;
    }
};

__global__ void k_fill_chunks(ChunkCount f, uint32_t n_aids, const uint64_t* chunk_start, uint64_t* chunks) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_aids) return;
    const uint64_t c = f((int64_t)x);
    const uint64_t s = chunk_start[x];
    for (uint64_t q = 0; q < c; ++q) chunks[s + q] = (uint64_t)x | (q << 26);
}

// Every total the index build needs on the host (buffer sizes, statistics) in ONE reduction over the aids, so the build
// synchronises with the host once instead of after each of its scans:
//   [0] records  [1] runs  [2..4] records per bin  [5..7] runs per bin  [8..10] work items per bin
//   [11..13] heavy items per layout  [14..16] heavy aids per layout (pilots)  [17] partition chunks
constexpr int N_TOTALS = 18;
constexpr int ITEM_BLOCK_AIDS = 2048;      // aids per workgroup of k_aid_totals / k_items_fill (same split in both)
constexpr int N_ITEM_SCANS = 10;           // totals [8, 18): the counters whose prefixes place the work items
// block_part (nullable): [gridDim.x][N_ITEM_SCANS] the block's own sums of totals [8, 18) -- k_items_fill adds up the
// blocks before it instead of running ten device-wide scans.
__global__ __launch_bounds__(256) void k_aid_totals(const uint64_t* cnt64, const uint8_t* boost, const uint32_t* flag, uint32_t n_aids,
                                                    uint32_t l_cap, int allow_packed, unsigned long long* totals,
                                                    unsigned long long* block_part) {
    __shared__ unsigned long long s_t[4][N_TOTALS];
    unsigned long long t[N_TOTALS];
#pragma unroll
    for (int q = 0; q < N_TOTALS; ++q) t[q] = 0;
    const uint32_t x_end = min(n_aids, (blockIdx.x + 1) * (uint32_t)ITEM_BLOCK_AIDS);
    for (uint32_t x = blockIdx.x * ITEM_BLOCK_AIDS + threadIdx.x; x < x_end; x += 256) {
        const uint64_t c64 = cnt64[x], n = c64 & CNT_REC_MASK, r = c64 >> CNT_REC_BITS;
        if (n == 0) continue;
        const int b = n <= (uint64_t)S_CAP ? 0 : (n <= (uint64_t)M_CAP ? 1 : 2);
        t[0] += n; t[1] += r; t[2 + b] += n; t[5 + b] += r;
        ItemCount f{cnt64, boost, flag, b, 0, l_cap, allow_packed, -1};
        const uint64_t items = f((int64_t)x);
        t[8 + b] += items;
        if (b == 2) {
            const int mode = heavy_mode(c64, allow_packed, l_cap);
            t[11 + mode] += items;
            t[14 + mode] += 1;
            ChunkCount cf{cnt64, boost, flag, 0, l_cap, allow_packed};
            t[17] += cf((int64_t)x);
        }
    }
#pragma unroll
    for (int q = 0; q < N_TOTALS; ++q) {
        unsigned long long v = t[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane_id() == 0) s_t[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < N_TOTALS) {
        const unsigned long long v = s_t[0][threadIdx.x] + s_t[1][threadIdx.x] + s_t[2][threadIdx.x] + s_t[3][threadIdx.x];
        if (v) atomicAdd(&totals[threadIdx.x], v);
        if (block_part && threadIdx.x >= N_TOTALS - N_ITEM_SCANS)
            block_part[(size_t)blockIdx.x * N_ITEM_SCANS + threadIdx.x - (N_TOTALS - N_ITEM_SCANS)] = v;
    }
}

// All work lists of a build in ONE launch (replaces ten device-wide scans and seven fill launches): work items of the
// three bins, the heavy bin's processing order per layout (pilots first), the partition-pass chunks and litem_start.
// A workgroup owns the same ITEM_BLOCK_AIDS aids as in k_aid_totals; its base prefixes come from k_item_part_scan over the
// workgroups' partial sums, the prefixes inside the block are block scans in aid order.
struct ItemFillArgs {
    const uint64_t* cnt64;
    const uint8_t* boost;
    uint32_t n_aids;
    uint32_t l_cap;
    int allow_packed;
    const unsigned long long* block_part;   // [workgroup][N_ITEM_SCANS] exclusive prefixes
    uint64_t* items[3];
    uint32_t* order[3];
    uint64_t n_pilots[3];
    uint64_t* chunks;
    uint64_t* litem_start;     // [n_aids + 1]
};
// block_part[b][q] -> sum over the workgroups before b (in place): ten columns, one wave each walks its column in strides of 64
__global__ __launch_bounds__(64 * N_ITEM_SCANS) void k_item_part_scan(unsigned long long* block_part, uint32_t nblk) {
    const int q = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    unsigned long long run = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
        const uint32_t b = b0 + lane;
        const unsigned long long v = b < nblk ? block_part[(size_t)b * N_ITEM_SCANS + q] : 0ull;
        const unsigned long long inc = wave_incl_scan(v);
        if (b < nblk) block_part[(size_t)b * N_ITEM_SCANS + q] = run + inc - v;
        run += (unsigned long long)__shfl((unsigned long long)inc, 63, 64);
    }
}

__global__ __launch_bounds__(256) void k_items_fill(ItemFillArgs a) {
    __shared__ unsigned long long s_base[N_ITEM_SCANS];
    __shared__ uint64_t s_sc[256 / 64 + 1];
    if (threadIdx.x < N_ITEM_SCANS) s_base[threadIdx.x] = a.block_part[(size_t)blockIdx.x * N_ITEM_SCANS + threadIdx.x];   // exclusive prefixes (k_item_part_scan)
    __syncthreads();
    // running prefixes (order of totals [8, 18)): [0..2] items per bin, [3..5] heavy items per layout, [6..8] heavy aids per layout, [9] chunks
    uint64_t run[N_ITEM_SCANS];
#pragma unroll
    for (int q = 0; q < N_ITEM_SCANS; ++q) run[q] = s_base[q];
    for (int it = 0; it < ITEM_BLOCK_AIDS / 256; ++it) {
        const uint32_t x = blockIdx.x * ITEM_BLOCK_AIDS + it * 256 + threadIdx.x;
        int b = -1, mode = -1;
        uint64_t items = 0, nch = 0;
        if (x < a.n_aids) {
            const uint64_t c64 = a.cnt64[x], n = c64 & CNT_REC_MASK;
            if (n) {
                b = n <= (uint64_t)S_CAP ? 0 : (n <= (uint64_t)M_CAP ? 1 : 2);
                items = 1;
                if (b == 2) {
                    mode = heavy_mode(c64, a.allow_packed, a.l_cap);
                    const int lg = l_log2r(c64, a.boost[x], a.l_cap, a.allow_packed);
                    items = 1ull << lg;
                    if (lg) nch = ((c64 >> CNT_REC_BITS) + PART_CHUNK_RUNS - 1) / PART_CHUNK_RUNS;
                }
            }
        }
        // five 0/1 counters travel in one word (12 bits each: at most 256 per round)
        const uint64_t small = (uint64_t)(b == 0) | (uint64_t)(b == 1) << 12 | (uint64_t)(mode == 0) << 24 | (uint64_t)(mode == 1) << 36 |
                               (uint64_t)(mode == 2) << 48;
        uint64_t tot_small, tot_l, tot_m[3], tot_c;
        const uint64_t ex_small = block_excl_scan<uint64_t, 256>(small, s_sc, &tot_small);
        const uint64_t ex_l = block_excl_scan<uint64_t, 256>(b == 2 ? items : 0ull, s_sc, &tot_l);
        uint64_t ex_m[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) ex_m[m] = block_excl_scan<uint64_t, 256>(mode == m ? items : 0ull, s_sc, &tot_m[m]);
        const uint64_t ex_c = block_excl_scan<uint64_t, 256>(nch, s_sc, &tot_c);
        if (b == 0) a.items[0][run[0] + (ex_small & 0xFFF)] = (uint64_t)x;
        if (b == 1) a.items[1][run[1] + ((ex_small >> 12) & 0xFFF)] = (uint64_t)x;
        const uint64_t s = run[2] + ex_l;
        if (x < a.n_aids) a.litem_start[x] = s;
        if (x == a.n_aids - 1) a.litem_start[a.n_aids] = s + (b == 2 ? items : 0ull);
        if (b == 2) {
            uint64_t lg = 0;
            while ((1ull << lg) < items) ++lg;
            for (uint64_t p = 0; p < items; ++p) a.items[2][s + p] = (uint64_t)x | (p << 26) | (lg << 50);
            const uint64_t r = run[6 + mode] + ((ex_small >> (24 + 12 * mode)) & 0xFFF);
            const uint64_t sm = run[3 + mode] + ex_m[mode];
            uint32_t* ord = a.order[mode];
            ord[r] = (uint32_t)s;
            for (uint64_t p = 1; p < items; ++p) ord[a.n_pilots[mode] + sm + p - r - 1] = (uint32_t)(s + p);
            const uint64_t cs = run[9] + ex_c;
            for (uint64_t q = 0; q < nch; ++q) a.chunks[cs + q] = (uint64_t)x | (q << 26);
        }
        run[0] += tot_small & 0xFFF;
        run[1] += (tot_small >> 12) & 0xFFF;
        run[2] += tot_l;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            run[6 + m] += (tot_small >> (24 + 12 * m)) & 0xFFF;
            run[3 + m] += tot_m[m];
        }
        run[9] += tot_c;
    }
}

// ---- the reduce kernel --------------------------------------------------------------------------------
// PACKED (S / M bins, not the time group): a table slot is ONE 64-bit word  aid_y << 36 | c2 << 24 | c1 << 12 | c0
// (an aid of these bins has <= M_CAP < 4096 records, so no 12-bit counter can overflow): one LDS read per probe,
// one LDS atomic per record, and half the LDS of the wide layout (key word + three 32-bit counters) -> twice the
// resident workgroups per CU. The L bin and the time group keep the wide layout.
constexpr uint64_t TAB_EMPTY = ~0ull;

struct ItemDesc {      // everything a workgroup needs about its item, fetched one item ahead by thread 0
    uint32_t it;
    uint64_t item;
    uint64_t rb, re;   // run range of the aid
    uint64_t ps, pe;   // bucket range (partitioned heavy aids)
};

#ifdef OTTO_PHASE_PROF
#define OTTO_PH(i) do { if (threadIdx.x == 0) { const unsigned long long _t = clock64(); ph[i] += _t - ph_t; ph_t = _t; } } while (0)
#else
#define OTTO_PH(i) do {} while (0)
#endif
// DBG: the timing diagnostics of `debug_skip` (wrong results) are compiled into a second instantiation only
template <int LOG2T, int THREADS, int GROUP, bool PACKED, int MINW, int GU, int INS_CH, bool DBG>
__global__ __launch_bounds__(THREADS, MINW) void k_reduce(ReduceArgs a) {
    constexpr int T = 1 << LOG2T;
    constexpr int NW = THREADS / 64;
    // TP: the time-weighted kind in the packed layout. Its weight 65536 * count + sum of the Q16 time extras is ONE sum
    // of (65536 + extra) per record, < 2^30 for the keys of an S / M aid (<= 3072 records) and of a packed-layout heavy aid: the slot is aid_y << 36 | that sum, one LDS atomic per
    // record instead of three, a 64-bit key, and the 40 KB table (four workgroups per CU) instead of 72 KB of wide counters (two).
    constexpr bool TP = GROUP == OTTO_COVIS_GROUP_TIME && PACKED;
    using K = typename std::conditional<GROUP == OTTO_COVIS_GROUP_TIME && !PACKED, KeyW, KeyN>::type;
    constexpr bool WIDE = GROUP == OTTO_COVIS_GROUP_TIME && !PACKED;
    // (TP in the heavy bin: only the packed layouts' aids, fewer than PACKED_MAX_RUNS = 4096 runs -- a key gains at most one record per
    // run of x, so its sum stays below 4096 * 2^18 = 2^30)
    auto kw = [](K key) -> uint64_t { if constexpr (TP) return key.c >> REC_AID_BITS; else return kweight(key); };   // Q16 weight of a key
    // work items dequeued with an atomic counter. The one-wave bin too (round 3): with a static stride a workgroup's ~220 aids of
    // 1 .. 256 records add up to sums that differ by +-25 % across the 5,120 workgroups, and the kernel lasts as long as the unluckiest
    constexpr bool DYNAMIC = true;
    constexpr int EXCAP = 64;
    constexpr int PKD = GROUP == OTTO_COVIS_GROUP_TIME ? 1 : PK;      // the time-weighted group has a single kind
    // BOUND (type-weighted group): the kinds of a pass are linear in the same three counters with per-type coefficients
    // between lo[t] = min_j coef[j][t] and hi[t] = max_j coef[j][t], so ONE lower-bound key KL and ONE upper-bound key KH
    // per table slot bracket the key of every kind: KL <= key_j <= KH. With lambda <= the k-th largest KL, every kind's
    // top-k lies inside {KH >= lambda}: one selection instead of PKD, then each kind ranks the short candidate list.
    constexpr bool BOUND = GROUP == OTTO_COVIS_GROUP_TYPE && PKD > 1 && NW == 1;   // one-wave bin only: see DESIGN.md (heavy aids: the band is too wide)
    constexpr int CCAP = 256;               // candidate slots per item (more: exact single-wave fallback per kind)
    constexpr bool HOT = NW > 1 && GROUP == OTTO_COVIS_GROUP_TYPE;   // multi-wave bins: top-k walks over the heavy keys only
    __shared__ uint64_t s_tab[PACKED ? T : 1];
    __shared__ uint32_t s_key[PACKED ? 1 : T];
    __shared__ uint32_t s_v[3][PACKED ? 1 : T];
    // Multi-wave bins keep a dense list of the occupied slots, appended to when a key enters the table: the top-k passes
    // walk ceil(occupied / THREADS) entries per lane instead of T / THREADS slots (the tables run 15 - 40 % full), and the
    // table is cleared through the list. More distinct keys than OCAP: the passes fall back to walking the table.
    constexpr int OCAP = NW == 1 ? 1 : (LOG2T == 12 ? 1920 : (LOG2T == 14 ? 10240 : (THREADS == 512 ? 4608 : 6144)));
    constexpr int RCAP = OCAP / NW;         // one list region per wave: appended to with a wave-private counter (no atomics)
    __shared__ uint16_t s_occ[OCAP];
    __shared__ uint32_t s_wcnt[NW];         // keys each wave entered into the table (published after the insert phase)
    __shared__ uint32_t s_hcnt[NW];         // of them, the "heavy" ones (more than one click record), moved to the front of the wave's region
    __shared__ uint16_t s_lbi[NW > 1 ? PKD : 1][NW > 1 ? THREADS : 1];         // slot of every lane's best key per kind (0xFFFF: none)
    __shared__ uint64_t s_exw[(NW > 1 || BOUND) ? PKD : 1][(NW > 1 || BOUND) ? EXCAP : 1];   // candidates above the threshold / rank broadcast
    __shared__ uint16_t s_cand[BOUND ? CCAP : 1];                               // BOUND: slots of the candidates
    __shared__ uint32_t s_exy[1][(NW > 1 && WIDE) ? EXCAP : 1];
    __shared__ uint64_t s_thrw[PK];
    __shared__ uint32_t s_thry[PK];
    __shared__ uint32_t s_nex[PK];
    __shared__ uint32_t s_more;
    __shared__ uint32_t s_ovf;
    constexpr int LISTCAP = BOUND ? S_CAP : 256;
    __shared__ uint16_t s_list[NW == 1 ? LISTCAP : 1];                         // one-wave bins: compacted valid slots
    __shared__ uint8_t s_seg[NW * 256 + 32];                                   // gather: segment -> descriptor lane, per wave (+ pad: the gather reads past a wave's last segment)
    __shared__ ItemDesc s_cur;
    __shared__ ItemDesc s_nxt;                                                // the item after s_cur (for the record prefetch)

    const int wid = threadIdx.x >> 6;
    const unsigned lane = lane_id();
#ifdef OTTO_PHASE_PROF
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = clock64();
#endif

    // ---- item pipeline: (index, item word, run range, bucket range) of the NEXT item are fetched while the
    //      current one is reduced, so the dependent loads index -> item -> run_start are off the critical path
    auto fetch_item = [&](uint32_t idx, ItemDesc& d) {     // stage 1: item word
        d.it = 0xFFFFFFFFu;             // past the end of the work list
        d.item = 0ull;
        if (idx < a.n_work) {
            d.it = a.order ? a.order[idx] : idx;
            d.item = a.items[d.it];
        }
    };
    auto fetch_ranges = [&](ItemDesc& d) {                 // stage 2: ranges (needs the item word)
        d.rb = d.re = d.ps = d.pe = 0;
        if (d.it != 0xFFFFFFFFu) {
            const uint32_t xx = (uint32_t)(d.item & REC_AID_MASK);
            d.rb = a.run_start[xx];
            d.re = a.run_start[xx + 1];
            if ((d.item >> 50) != 0 && a.pstart) {
                d.ps = a.pstart[d.it];
                const uint64_t cap = a.pstart[d.it + 1] - d.ps, got = a.pcursor[d.it];       // a full (flagged) bucket: got > cap
                d.pe = d.ps + (got < cap ? got : cap);
            }
        }
    };
    ItemDesc cur, nx;
    uint32_t idx_next = 0, idx_far = 0;   // DYNAMIC, thread 0: dequeued indices of the next two items
    // Items are reserved DQ at a time (same-address atomics retire at ~90 M/s: one per item would cost 5 ms for the M
    // bin alone); the next chunk is requested when the current one is opened, so its latency is never waited for.
    constexpr uint32_t DQ = THREADS == S_THREADS ? 16 : 4;      // (1.1 M one-wave items: 70 k counter bumps)
    uint32_t pool = 0, pool_end = 0, pool_next = 0;
    auto take = [&]() {
        const uint32_t r = pool++;
        if (pool == pool_end) {
            pool = pool_next;
            pool_end = pool + DQ;
            pool_next = atomicAdd(a.work_counter, DQ);
        }
        return r;
    };
    if (DYNAMIC) {
        if (threadIdx.x == 0) {
            pool = atomicAdd(a.work_counter, 2 * DQ);
            pool_end = pool + DQ;
            pool_next = pool + DQ;
            const uint32_t i0 = take();
            idx_next = take();
            idx_far = take();
            fetch_item(i0, cur);
            fetch_ranges(cur);
            s_cur = cur;
        }
    } else {
        fetch_item(blockIdx.x, cur);
        fetch_ranges(cur);
    }
    uint32_t sidx = blockIdx.x;          // static schedule: position in the work list
    auto clear_slot = [&](int i) {
        if (PACKED) {
            s_tab[PACKED ? i : 0] = TAB_EMPTY;
        } else {
            s_key[PACKED ? 0 : i] = KEY_EMPTY;
            s_v[0][PACKED ? 0 : i] = 0; s_v[1][PACKED ? 0 : i] = 0; s_v[2][PACKED ? 0 : i] = 0;
        }
    };
    auto clear_table = [&]() {
        for (int i = threadIdx.x; i < T; i += THREADS) clear_slot(i);
    };
    if (NW > 1) {
        clear_table();
        if (threadIdx.x == 0) s_ovf = 0;
    }
    // Records of a bucketed heavy-aid partition per thread and round trip: a whole partition (wide: l_cap = 6 * 1024,
    // packed: 12 * 1024) in one. PREF: the first round of the NEXT item is requested during the tail of the current
    // item's top-k (after its last walk over the table), so a one-workgroup-per-CU kernel no longer sits through the whole
    // HBM round trip + the CU's ingest time (48 KB at ~24 GB/s) between two items.
    constexpr int BU = GROUP == OTTO_COVIS_GROUP_TIME ? 4 : ((PACKED && THREADS == L_THREADS) ? 12 : 6);
    constexpr bool PREF = THREADS == L_THREADS && GROUP != OTTO_COVIS_GROUP_TIME;   // the 1024-thread kernels that read partition buckets
    // the 512-thread kernel has no registers to hold a prefetched round (6 more live values spill, and a spill waits for the
    // load): it only touches the next bucket's lines early (WARM), the loads of the item itself then come from L2
    constexpr bool WARM = (THREADS == L_THREADS || THREADS == 512) && GROUP != OTTO_COVIS_GROUP_TIME;
    uint32_t pre[PREF ? BU : 1];
    bool pre_valid = false;                // uniform: pre[] holds the first round of the item that becomes `cur` next
    auto prefetch_next = [&]() {
        if (!PREF) return;
        const uint64_t nitem = s_nxt.item;
        pre_valid = s_nxt.it != 0xFFFFFFFFu && (nitem >> 50) != 0 && a.pstart != nullptr;
        if (!pre_valid) return;
        const uint64_t ps = s_nxt.ps, pe = s_nxt.pe;
#pragma unroll
        for (int u = 0; u < (PREF ? BU : 1); ++u) {
            const uint64_t i = ps + threadIdx.x + (uint64_t)u * THREADS;
            pre[u] = i < pe ? a.prec[i] : KEY_EMPTY;
        }
    };

    for (;;) {
        if (DYNAMIC) {
            __syncthreads();
            cur = s_cur;
        }
        if (cur.it == 0xFFFFFFFFu) break;
        OTTO_PH(0);
        const uint32_t it = cur.it;
        // stage 1 of the next item
        if (DYNAMIC) {
            if (threadIdx.x == 0) fetch_item(idx_next, nx);
#ifdef OTTO_PHASE_PROF
            if (threadIdx.x == 0) { const unsigned long long _t = clock64(); ph[13] += _t - ph_t; }   // part of p1: next item's dependent index -> item loads
#endif
        } else {
            fetch_item(sidx + gridDim.x, nx);
        }
        const uint64_t item = cur.item;
        const uint32_t x = (uint32_t)(item & REC_AID_MASK);
        const uint32_t part = (uint32_t)((item >> 26) & 0xFFFFFFu);
        // only the heavy bin's items are hash partitions: for the M instantiation lgR is the constant 0, so the guess path, the
        // bucket reads, the partial lists and the partition filter of the gather compile away -- the M kernel sits at its 128-VGPR
        // cap, every path it does not need is registers back (scratch 8 -> 0 bytes, 6.6 -> 6.27 ms). (The one-wave S kernel also
        // never sees a partition, but with the constant it got SLOWER, 2.85 -> 2.98 ms at 73 instead of 85 VGPRs: left as it was.)
        constexpr bool CAN_PART = THREADS != M_THREADS;
        const int lgR = CAN_PART ? (int)(item >> 50) : 0;
        const uint32_t pmask = (1u << lgR) - 1u;
        const int pshift = 32 - LOG2T - lgR;

        // Partitions of one heavy aid are equal-probability hash samples of its pairs, so their top-k thresholds
        // agree closely: a partition reduced earlier leaves a lower bound of its 32nd best key in tau[x]; every key
        // above that guess is collected in ONE pass over the table and, if at least k were found (and the list did
        // not overflow), the exact top-k is the sorted list -- no lane-bests, no block-wide selection, no second pass.
        // Any guess is safe: too few / too many candidates fall back to the two-pass path below.
        K guess[PKD];
        const bool use_guess = NW > 1 && lgR > 0 && a.tau_w != nullptr;
#pragma unroll
        for (int j = 0; j < PKD; ++j) {
            kclear(guess[j]);
            if (use_guess && j < (BOUND ? 1 : a.nk)) {
                const size_t o = (size_t)j * a.n_aids + x;
                kload(guess[j], a.tau_w[o], WIDE ? a.tau_y[o] : 0u);
            }
        }
        if (NW == 1 && !(DBG && (a.debug_skip & 4))) clear_table();      // multi-wave bins: cleared at the end of the previous item
        if (NW == 1) {                       // multi-wave bins: s_ovf is reset with the table at the end of the previous item
            s_ovf = 0;
            wave_lds_sync();
        }
        OTTO_PH(1);

        // Records go into the table N at a time: the first-probe CAS of all N is issued back to back (N independent LDS
        // round trips in flight per lane instead of one), then hits (key already there: add) and misses (next probe) are
        // resolved. `e` = time extra (GROUP_TIME only). CAS first, no read: a new key (most records) costs ONE LDS
        // operation in the packed layout -- it goes in together with its first count.
        auto probe_step = [&](uint32_t y) -> uint32_t { return (((y * 0x85EBCA6Bu) ^ (y >> 7)) * 0xC2B2AE35u >> (32 - LOG2T)) | 1u; };
        // a key entered the table at `slot`: append the slot to this wave's region of the dense list. The position comes from
        // a wave-private counter and the ballot rank: no LDS atomic, no round trip (every lane of the wave calls this
        // together; lanes that claim a slot later, inside a probe loop, report it after the loop through the same call)
        uint32_t wc = 0;
        // m: wave mask of the lanes whose key is new (compares go straight into wave masks: a bool would be materialised in a
        // register and compared again)
        auto note_new = [&](uint64_t m, uint32_t slot) {
            if (NW == 1) return;
            if (m == 0) return;
            const uint32_t pos = wc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (__builtin_amdgcn_inverse_ballot_w64(m) && pos < (uint32_t)RCAP) s_occ[NW > 1 ? wid * RCAP + pos : 0] = (uint16_t)slot;
            wc += (uint32_t)__popcll(m);
        };
        constexpr int ICMP_EQ = 32, ICMP_NE = 33;
        // Double hashing: the probe step is a second hash of the key (odd: every slot is visited in T probes). Linear probing
        // builds clusters, and a wave waits for its longest chain: every probe is a dependent LDS round trip.
        // returns the slot it claimed for a NEW key, else 0xFFFFFFFF
        auto probe_on = [&](uint32_t y, uint32_t slot, unsigned long long add, unsigned long long fresh) -> uint32_t {   // packed: slots after the first
            const uint32_t step = probe_step(y);
            for (int probe = 1; probe < T; ++probe) {
                slot = (slot + step) & (T - 1);
                const unsigned long long old = atomicCAS((unsigned long long*)&s_tab[PACKED ? slot : 0], (unsigned long long)TAB_EMPTY, fresh | add);
                if (old == TAB_EMPTY) return slot;
                if ((uint32_t)(old >> 36) == y) {
                    atomicAdd((unsigned long long*)&s_tab[PACKED ? slot : 0], add);
                    return 0xFFFFFFFFu;
                }
            }
            s_ovf = 1;
            return 0xFFFFFFFFu;
        };
        auto wide_add = [&](uint32_t rc, uint32_t found, uint32_t e) {
            if (GROUP == OTTO_COVIS_GROUP_TIME) {
                atomicAdd(&s_v[0][PACKED ? 0 : found], 1u);
                const uint32_t old = atomicAdd(&s_v[1][PACKED ? 0 : found], e);
                if (old + e < old) atomicAdd(&s_v[2][PACKED ? 0 : found], 1u);
            } else if (GROUP == OTTO_COVIS_GROUP_TYPE) {
                const uint32_t tyj = (rc >> REC_AID_BITS) & 3u;            // exactly one counter: one LDS atomic, no divergence
                if (tyj < 3u) atomicAdd(&s_v[tyj][PACKED ? 0 : found], 1u);
            } else {
                const uint32_t fb = (rc >> 28) >> a.chan_shift;
                if (fb & 1u) atomicAdd(&s_v[0][PACKED ? 0 : found], 1u);
                if (fb & 2u) atomicAdd(&s_v[1][PACKED ? 0 : found], 1u);
                if (fb & 4u) atomicAdd(&s_v[2][PACKED ? 0 : found], 1u);
            }
        };
        auto packed_add = [&](uint32_t rc, uint32_t e) -> unsigned long long {
            uint32_t add0, add1, add2;
            if (GROUP == OTTO_COVIS_GROUP_TIME) {
                return 65536ull + (unsigned long long)e;
            } else if (GROUP == OTTO_COVIS_GROUP_TYPE) {
                const uint32_t tyj = (rc >> REC_AID_BITS) & 3u;              // counter tyj: bit 12 * tyj (type 3 does not occur: 36 & 31 = 4 is masked off)
                return (unsigned long long)((1u << ((tyj * 12u) & 31u)) & 0x01001001u);
            } else {
                const uint32_t fb = (rc >> 28) >> a.chan_shift;
                add0 = fb & 1u; add1 = (fb >> 1) & 1u; add2 = (fb >> 2) & 1u;
            }
            return (unsigned long long)add0 | ((unsigned long long)add1 << 12) | ((unsigned long long)add2 << 24);
        };
        // CH records per lane are in flight at a time (more would spill: the 2^14 kernel runs at 128 VGPRs); a chunk no
        // lane of the wave has a record for is skipped as a whole (small aids: most of a gather batch is empty)
        auto insert_batch = [&](auto ntag, const uint32_t* rc, const uint64_t* okin, const uint32_t* e) {     // okin: wave masks
            constexpr int N = decltype(ntag)::value;
            constexpr int CH = INS_CH;
            if (DBG && (a.debug_skip & 8)) {                 // diagnostics: records are fetched but not inserted
#pragma unroll
                for (int u = 0; u < N; ++u)
                    if (__builtin_amdgcn_inverse_ballot_w64(okin[u]) && rc[u] == 0xDEADBEEFu) s_ovf = 1;
                return;
            }
#pragma unroll
            for (int c0 = 0; c0 < N; c0 += CH) {
                bool ok[CH];
                uint64_t okm[CH];
                uint64_t any = 0;
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const int u = c0 + q;
                    uint64_t m = u < N ? okin[u < N ? u : 0] : 0ull;
                    if (GROUP == OTTO_COVIS_GROUP_FILTER) m &= __builtin_amdgcn_uicmp(((rc[u < N ? u : 0] >> 28) >> a.chan_shift) & 7u, 0u, ICMP_NE);
                    okm[q] = m;
                    ok[q] = __builtin_amdgcn_inverse_ballot_w64(m);
                    any |= m;
                }
                if (any == 0) continue;
                if (PACKED) {
                    uint32_t oldhi[CH];                       // high word of the slot before the CAS: 0xFFFFFFFF = was empty
                    unsigned long long addq[CH];              // the record's counter increment, built once
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        const uint32_t r = rc[c0 + q < N ? c0 + q : 0];
                        const uint32_t slot = rec_hash(r) >> (32 - LOG2T);
                        oldhi[q] = 0xFFFFFFFFu;
                        addq[q] = packed_add(r, e[c0 + q < N ? c0 + q : 0]);
                        if (ok[q]) {
                            const unsigned long long o = atomicCAS((unsigned long long*)&s_tab[PACKED ? slot : 0], (unsigned long long)TAB_EMPTY,
                                                                   ((unsigned long long)(r & REC_AID_MASK) << 36) | addq[q]);
                            oldhi[q] = (uint32_t)(o >> 32);
                        }
                        note_new(okm[q] & __builtin_amdgcn_uicmp(oldhi[q], 0xFFFFFFFFu, ICMP_EQ), slot);
                    }
#pragma unroll
                    for (int q = 0; q < CH; ++q) {
                        // oldhi == 0xFFFFFFFF: not a record, or a new key that went in with its first count
                        const uint32_t r = rc[c0 + q < N ? c0 + q : 0];
                        const uint32_t y = r & REC_AID_MASK, slot = rec_hash(r) >> (32 - LOG2T);
                        uint32_t late = 0xFFFFFFFFu;
                        if (oldhi[q] != 0xFFFFFFFFu) {
                            if ((oldhi[q] >> 4) == y) atomicAdd((unsigned long long*)&s_tab[PACKED ? slot : 0], addq[q]);
                            else late = probe_on(y, slot, addq[q], (unsigned long long)y << 36);
                        }
                        note_new(__builtin_amdgcn_uicmp(late, 0xFFFFFFFFu, ICMP_NE), late);
                    }
                    continue;
                }
                uint32_t old[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const uint32_t r = rc[c0 + q < N ? c0 + q : 0];
                    old[q] = KEY_EMPTY;
                    if (ok[q]) old[q] = atomicCAS(&s_key[PACKED ? 0 : rec_hash(r) >> (32 - LOG2T)], KEY_EMPTY, r & REC_AID_MASK);
                    note_new(okm[q] & __builtin_amdgcn_uicmp(old[q], KEY_EMPTY, ICMP_EQ), rec_hash(r) >> (32 - LOG2T));
                }
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const uint32_t r = rc[c0 + q < N ? c0 + q : 0];
                    const uint32_t y = r & REC_AID_MASK;
                    uint32_t sl = rec_hash(r) >> (32 - LOG2T);
                    uint32_t late = 0xFFFFFFFFu;
                    if (ok[q]) {
                        bool found = old[q] == KEY_EMPTY || old[q] == y;
                        const uint32_t step = probe_step(y);
                        for (int probe = 1; !found && probe < T; ++probe) {
                            sl = (sl + step) & (T - 1);
                            const uint32_t o2 = atomicCAS(&s_key[PACKED ? 0 : sl], KEY_EMPTY, y);
                            if (o2 == KEY_EMPTY) late = sl;
                            found = o2 == KEY_EMPTY || o2 == y;
                        }
                        if (!found) s_ovf = 1;
                        else wide_add(r, sl, e[c0 + q < N ? c0 + q : 0]);
                    }
                    note_new(__builtin_amdgcn_uicmp(late, 0xFFFFFFFFu, ICMP_NE), late);
                }
            }
        };

        // keys of table slot i for every kind of this pass (the slot is read once)
        auto slot_keys = [&](int i, K (&out)[PKD]) {
            uint32_t y, v0, v1, v2;
            if (PACKED) {
                const uint64_t v = s_tab[i];
                y = v == TAB_EMPTY ? KEY_EMPTY : (uint32_t)(v >> 36);
                v0 = (uint32_t)v & 0xFFFu; v1 = (uint32_t)(v >> 12) & 0xFFFu; v2 = (uint32_t)(v >> 24) & 0xFFFu;
            } else {
                y = s_key[i];
                v0 = s_v[0][i]; v1 = s_v[1][i]; v2 = s_v[2][i];
            }
#pragma unroll
            for (int j = 0; j < PKD; ++j) {
                uint64_t uw = 0, qw = 0;
                if (y != KEY_EMPTY) {
                    if (TP) uw = (uint64_t)v0 | ((uint64_t)v1 << 12) | ((uint64_t)v2 << 24);
                    else if (GROUP == OTTO_COVIS_GROUP_TIME) qw = 65536ull * v0 + (((uint64_t)v2 << 32) | v1);
                    else if (PACKED) uw = __umul24(v0, a.coef[j][0]) + __umul24(v1, a.coef[j][1]) + __umul24(v2, a.coef[j][2]);   // 12-bit counts x 8-bit weights: v_mad_u32_u24 (v_mul_lo_u32 is quarter rate)
                    else uw = (uint64_t)v0 * a.coef[j][0] + (uint64_t)v1 * a.coef[j][1] + (uint64_t)v2 * a.coef[j][2];
                }
                kmake(out[j], uw, qw, y);
            }
        };
        // sorted list (lane i = i-th best) of kind j -> output rows, or this partition's partial list
        auto emit = [&](int j, K best) {
            const bool valid = (int)lane < a.k && kvalid(best);
            if (lgR == 0) {
                const size_t o = ((size_t)(a.kind_base + j) * a.n_aids + x) * (size_t)a.k + lane;
                if (valid) { a.out_y[o] = kaid(best); a.out_w[o] = kw(best); }
                const int nvalid = __popcll(__ballot(valid));
                if (lane == 0) a.out_n[(size_t)(a.kind_base + j) * a.n_aids + x] = nvalid;
            } else if ((int)lane < a.k) {
                const size_t o = ((size_t)it * a.nk + j) * (size_t)a.k + lane;
                kstore(best, &a.part_w[o], &a.part_y[o]);
            }
        };
        if (DBG && (a.debug_skip & 1)) {
        } else if (lgR > 0 && a.pstart) {
            // heavy aid, records already bucketed by hash partition: contiguous coalesced reads
            const uint64_t ps = cur.ps, pe = cur.pe;
            bool first = true;
            for (uint64_t i0 = ps + threadIdx.x; i0 < pe; i0 += BU * THREADS) {
                uint32_t rc[BU], e[BU];
                if (PREF && first && pre_valid) {
#pragma unroll
                    for (int u = 0; u < BU; ++u) { rc[u] = pre[PREF ? u : 0]; e[u] = 0u; }
                } else {
#pragma unroll
                    for (int u = 0; u < BU; ++u) {
                        const uint64_t i = i0 + (uint64_t)u * THREADS;
                        rc[u] = i < pe ? a.prec[i] : KEY_EMPTY;
                        e[u] = (GROUP == OTTO_COVIS_GROUP_TIME && i < pe) ? a.ptw[i] : 0u;
                    }
                }
                first = false;
#ifdef OTTO_PHASE_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) { const unsigned long long _t = clock64(); ph[12] += _t - ph_t; ph[2] -= _t - ph_t; }   // p2 = inserts only; ph[12] = wait for the records
#endif
                uint64_t okb[BU];
#pragma unroll
                for (int u = 0; u < BU; ++u) okb[u] = __ballot(rc[u] != KEY_EMPTY);
                insert_batch(std::integral_constant<int, BU>{}, rc, okb, e);
                if (s_ovf) break;
            }
        } else {
            constexpr bool NEED_SL = GROUP == OTTO_COVIS_GROUP_TIME;
            for_each_record_seg<NW, GU, NEED_SL>(a.sorted_desc, a.rec, cur.rb, cur.re, wid, s_seg + wid * 256,
                                                 [&](uint32_t (&rc)[GU], uint64_t (&sl)[NEED_SL ? GU : 1], uint64_t (&ok)[GU]) {
                uint32_t e[GU];
                uint64_t okb[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    okb[u] = ok[u];
                    if (lgR != 0) okb[u] &= __ballot(((rec_hash(rc[u]) >> pshift) & pmask) == part);
                    e[u] = (GROUP == OTTO_COVIS_GROUP_TIME && __builtin_amdgcn_inverse_ballot_w64(okb[u])) ? a.tw[sl[NEED_SL ? u : 0]] : 0u;
                }
                insert_batch(std::integral_constant<int, GU>{}, rc, okb, e);
            });
        }
        if (NW > 1 && lane == 0) s_wcnt[NW > 1 ? wid : 0] = wc;
        pre_valid = false;                     // consumed (or not applicable); set again by prefetch_next below
        // stage 2 of the next item (its item word has long arrived)
        if (!DYNAMIC || threadIdx.x == 0) fetch_ranges(nx);
        OTTO_PH(2);
        __syncthreads();
        OTTO_PH(3);
        const bool ovf = s_ovf != 0;
        if constexpr (BOUND) {
        if (ovf) {
            // LDS table full: ask the host to redo this aid with twice the partitions
            if (threadIdx.x == 0) {
                a.flag[x] = 1;
                a.boost[x] = (uint8_t)(lgR - l_log2r(a.cnt64[x], 0, a.l_cap, a.allow_packed) + 1);
                atomicAdd(a.ovf_count, 1u);
            }
        } else if (!(DBG && (a.debug_skip & 2))) {
            constexpr int MPL = T / THREADS;
            // per-type bounds of the pass's weight vectors (wave-uniform)
            uint32_t lo0 = a.coef[0][0], lo1 = a.coef[0][1], lo2 = a.coef[0][2], hi0 = lo0, hi1 = lo1, hi2 = lo2;
            for (int j = 1; j < a.nk; ++j) {
                lo0 = min(lo0, a.coef[j][0]); lo1 = min(lo1, a.coef[j][1]); lo2 = min(lo2, a.coef[j][2]);
                hi0 = max(hi0, a.coef[j][0]); hi1 = max(hi1, a.coef[j][1]); hi2 = max(hi2, a.coef[j][2]);
            }
            // (y, counters) of table slot i; y == KEY_EMPTY: empty slot
            auto slot_read = [&](int i, uint32_t& y, uint32_t& v0, uint32_t& v1, uint32_t& v2) {
                if (PACKED) {
                    const uint64_t v = s_tab[PACKED ? i : 0];
                    y = v == TAB_EMPTY ? KEY_EMPTY : (uint32_t)(v >> 36);
                    v0 = (uint32_t)v & 0xFFFu; v1 = (uint32_t)(v >> 12) & 0xFFFu; v2 = (uint32_t)(v >> 24) & 0xFFFu;
                } else {
                    y = s_key[PACKED ? 0 : i];
                    v0 = s_v[0][PACKED ? 0 : i]; v1 = s_v[1][PACKED ? 0 : i]; v2 = s_v[2][PACKED ? 0 : i];
                }
            };
            auto weight = [&](uint32_t v0, uint32_t v1, uint32_t v2, uint32_t c0, uint32_t c1, uint32_t c2) -> uint64_t {
                if (PACKED) return (uint64_t)(__umul24(v0, c0) + __umul24(v1, c1) + __umul24(v2, c2));      // 12-bit counts x 8-bit weights: v_mad_u32_u24
                return (uint64_t)v0 * c0 + (uint64_t)v1 * c1 + (uint64_t)v2 * c2;
            };
            auto slot_lohi = [&](int i, K& kl, K& kh) {
                uint32_t y, v0, v1, v2;
                slot_read(i, y, v0, v1, v2);
                const bool valid = y != KEY_EMPTY;
                kmake(kl, valid ? weight(v0, v1, v2, lo0, lo1, lo2) : 0ull, 0ull, y);
                kmake(kh, valid ? weight(v0, v1, v2, hi0, hi1, hi2) : 0ull, 0ull, y);
            };
            auto slot_kind_key = [&](int i, int j) {
                uint32_t y, v0, v1, v2;
                slot_read(i, y, v0, v1, v2);
                K key;
                kmake(key, y != KEY_EMPTY ? weight(v0, v1, v2, a.coef[j][0], a.coef[j][1], a.coef[j][2]) : 0ull, 0ull, y);
                return key;
            };
            // rank r of kind j -> output row (unpartitioned aid) or this partition's partial list; nv valid keys in the wave
            auto emit_ranked = [&](int j, K key, uint32_t rank, int nv) {
                const bool put = kvalid(key) && rank < (uint32_t)a.k;
                if (lgR == 0) {
                    const size_t o = ((size_t)(a.kind_base + j) * a.n_aids + x) * (size_t)a.k + rank;
                    if (put) { a.out_y[o] = kaid(key); a.out_w[o] = kw(key); }
                    if (lane == 0) a.out_n[(size_t)(a.kind_base + j) * a.n_aids + x] = nv < a.k ? nv : a.k;
                } else {
                    const size_t base = ((size_t)it * a.nk + j) * (size_t)a.k;
                    if (put) kstore(key, &a.part_w[base + rank], &a.part_y[base + rank]);
                    if ((int)lane < a.k && (int)lane >= nv) { K z; kclear(z); kstore(z, &a.part_w[base + lane], &a.part_y[base + lane]); }
                }
            };
            // exact top-k of kind j by ONE wave over the whole table (two scans): only when the candidate list overflowed
            auto wave_exact_topk = [&](int j) {
                K lb;
                kclear(lb);
                int bi = -1;
                for (int i = (int)lane; i < T; i += 64) {
                    const K key = slot_kind_key(i, j);
                    if (kbetter(key, lb)) { lb = key; bi = i; }
                }
                K best = lb;
                wave_bitonic_sort_desc(best);
                for (int i = (int)lane; i < T; i += 64) {
                    K key = slot_kind_key(i, j);
                    if (i == bi) kclear(key);
                    wave_topk_push(best, key, a.k);
                }
                emit(j, best);
            };
            // kind j from the candidate list s_cand[0, ncand): rank = number of better candidates (keys are distinct: one
            // per aid_y), read back from LDS with uniform addresses -- no sorting network, every iteration independent
            auto finish_kind = [&](int j, uint32_t ncand) {
                if (ncand <= 64u) {
                    K key;
                    kclear(key);
                    if (lane < ncand) key = slot_kind_key((int)s_cand[lane], j);
                    s_exw[j][lane] = key.c;
                    wave_lds_sync();
                    uint32_t rank = 0;
#pragma unroll 4
                    for (uint32_t i = 0; i < ncand; ++i) rank += s_exw[j][i] > key.c ? 1u : 0u;
                    emit_ranked(j, key, rank, __popcll(__ballot(kvalid(key))));
                } else {
                    K c4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t idx = (uint32_t)q * 64u + lane;
                        kclear(c4[q]);
                        if (idx < ncand) c4[q] = slot_kind_key((int)s_cand[idx], j);
                    }
                    K best;
                    wave_topk_select<4, K>(c4, a.k, best);
                    emit(j, best);
                }
            };
            {
                // ---- one wave owns the table: compact the occupied slots, then select on the bound keys ----
                uint32_t nvalid = 0;
#pragma unroll
                for (int q = 0; q < MPL; ++q) {
                    const int i = q * THREADS + threadIdx.x;
                    const bool v = PACKED ? (s_tab[PACKED ? i : 0] != TAB_EMPTY) : (s_key[PACKED ? 0 : i] != KEY_EMPTY);
                    const uint64_t m = __ballot(v);
                    const uint32_t pos = nvalid + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (v && pos < (uint32_t)LISTCAP) s_list[NW == 1 ? pos : 0] = (uint16_t)i;
                    nvalid += (uint32_t)__popcll(m);
                }
                wave_lds_sync();
                uint32_t ncand = 0;
                bool overflow = nvalid > (uint32_t)LISTCAP;          // cannot happen (an S aid has <= S_CAP records); exact fallback anyway
                if (nvalid <= 40u || nvalid <= (uint32_t)a.k) {
                    // every occupied slot is a candidate
                    if (lane < nvalid) s_cand[lane] = s_list[NW == 1 ? lane : 0];
                    ncand = nvalid;
                } else if (!overflow) {
                    // lane-best of KL over the lane's list entries, one 64-lane sort, threshold = k-th best lane-best
                    // (a lower bound of the k-th largest KL), then every entry with KH >= threshold is a candidate
                    K lb;
                    kclear(lb);
                    for (uint32_t li = lane; li < nvalid; li += 64u) {
                        K kl, kh;
                        slot_lohi((int)s_list[NW == 1 ? li : 0], kl, kh);
                        if (kbetter(kl, lb)) lb = kl;
                    }
                    // k-th best lane-best by counting (64 independent LDS broadcasts instead of a 21-stage network)
                    s_exw[0][lane] = lb.c;
                    wave_lds_sync();
                    uint32_t rk = 0;
#pragma unroll 8
                    for (int i = 0; i < 64; ++i) rk += s_exw[0][i] > lb.c ? 1u : 0u;
                    const uint64_t mk = __ballot(kvalid(lb) && rk == (uint32_t)(a.k - 1));
                    K thr;
                    kclear(thr);
                    if (mk) thr = kshfl(lb, __ffsll((unsigned long long)mk) - 1);
                    wave_lds_sync();
                    for (uint32_t l0 = 0; l0 < nvalid; l0 += 64u) {
                        const uint32_t li = l0 + lane;
                        bool c = false;
                        uint16_t sl = 0;
                        if (li < nvalid) {
                            K kl, kh;
                            sl = s_list[NW == 1 ? li : 0];
                            slot_lohi((int)sl, kl, kh);
                            c = !kbetter(thr, kh);
                        }
                        const uint64_t m = __ballot(c);
                        const uint32_t pos = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (c && pos < (uint32_t)CCAP) s_cand[pos] = sl;
                        ncand += (uint32_t)__popcll(m);
                    }
                    overflow = ncand > (uint32_t)CCAP;
                }
                wave_lds_sync();
                for (int j = 0; j < a.nk; ++j) {
                    if (overflow) wave_exact_topk(j);
                    else finish_kind(j, ncand);
                }
                wave_lds_sync();
            }
        }
        OTTO_PH(4);                            // one-wave bin: p4 = its whole top-k, p0 = the wait for the next item's words
        } else if constexpr (NW == 1) {
        bool fast_done = false;
        OTTO_PH(7);
        if (fast_done || (DBG && (a.debug_skip & 2))) {
        } else if (ovf) {
            // LDS table full: ask the host to redo this aid with twice the partitions
            if (threadIdx.x == 0) {
                a.flag[x] = 1;
                a.boost[x] = (uint8_t)(lgR - l_log2r(a.cnt64[x], 0, a.l_cap, a.allow_packed) + 1);
                atomicAdd(a.ovf_count, 1u);
            }
        } else {
            constexpr int MPL = T / THREADS;
            static_assert(MPL <= 32, "consumed-slot bitmask is 32 bits");
            // One-wave bins: most small aids have at most 64 distinct partners. Then the valid slots are compacted to one
            // per lane (ballot ranks), each lane builds its keys once and ONE sorting network per kind is the whole top-k:
            // no lane-bests, no threshold rounds over the table.
            bool compact_done = false;
            if (NW == 1) {
                uint32_t nvalid = 0;
#pragma unroll
                for (int q = 0; q < MPL; ++q) {
                    const int i = q * THREADS + threadIdx.x;
                    const bool v = PACKED ? (s_tab[PACKED ? i : 0] != TAB_EMPTY) : (s_key[PACKED ? 0 : i] != KEY_EMPTY);
                    const uint64_t m = __ballot(v);
                    const uint32_t pos = nvalid + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (v && pos < (uint32_t)LISTCAP) s_list[pos] = (uint16_t)i;
                    nvalid += (uint32_t)__popcll(m);
                }
                // up to 256 keys: 2 or 4 per lane, built once and kept in registers; the selection (lane-best, interleaved
                // sorts, threshold rounds) then runs on registers instead of re-reading and re-deriving the table slots
                auto select_from_registers = [&](auto mtag) {
                    constexpr int MC = decltype(mtag)::value;
                    __syncthreads();
                    K c[PKD][MC];
#pragma unroll
                    for (int q = 0; q < MC; ++q) {
                        K kk[PKD];
#pragma unroll
                        for (int j = 0; j < PKD; ++j) kclear(kk[j]);
                        const uint32_t li = (uint32_t)q * 64u + lane;
                        if (li < nvalid) slot_keys((int)s_list[li], kk);
#pragma unroll
                        for (int j = 0; j < PKD; ++j) c[j][q] = kk[j];
                    }
                    K bests[PKD];
                    wave_topk_select_multi<MC, PKD, K>(c, a.nk, a.k, bests);
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (j < a.nk) emit(j, bests[j]);
                    compact_done = true;
                    __syncthreads();
                };
                if (nvalid > 64u && nvalid <= 128u) select_from_registers(std::integral_constant<int, 2>{});
                else if (nvalid > 128u && nvalid <= 256u) select_from_registers(std::integral_constant<int, 4>{});
                if (nvalid <= 64u) {
                    __syncthreads();
                    K bests[PKD];
#pragma unroll
                    for (int j = 0; j < PKD; ++j) kclear(bests[j]);
                    if (lane < nvalid) slot_keys((int)s_list[lane], bests);
                    wave_bitonic_sort_multi<PKD, K>(bests);
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (j < a.nk) emit(j, bests[j]);
                    compact_done = true;
                    __syncthreads();
                }
            }
            if (!compact_done) {
            // Candidate keys are NOT cached in registers (3 kinds x MPL keys would cost ~100 VGPRs and halve the
            // resident workgroups): they are recomputed from the LDS table when needed (one read + a few 32-bit
            // ops), with a per-kind bitmask of the slots of this lane that are already in a list.
            K lb[PKD];
            uint32_t done[PKD];
            uint32_t nonempty = 0;                   // slots of this lane that hold a key (the later passes skip the rest)
            {
                int bi[PKD];
#pragma unroll
                for (int j = 0; j < PKD; ++j) { kclear(lb[j]); bi[j] = 0; }
#pragma unroll 2
                for (int q = 0; q < MPL; ++q) {      // NOT fully unrolled: keeps a couple of slots in flight, not all MPL
                    K kk[PKD];
                    slot_keys(q * THREADS + threadIdx.x, kk);
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < PKD; ++j) {
                        any = any || kvalid(kk[j]);
                        if (kbetter(kk[j], lb[j])) { lb[j] = kk[j]; bi[j] = q; }
                    }
                    if (any) nonempty |= 1u << q;
                }
#pragma unroll
                for (int j = 0; j < PKD; ++j) done[j] = 1u << bi[j];
            }
            if (NW == 1) {
                // ---- one wave owns the whole table: sort the lane-bests of all kinds (interleaved networks),
                //      then insert the few remaining candidates that still beat the k-th entry -------------
                K bests[PKD];
#pragma unroll
                for (int j = 0; j < PKD; ++j) bests[j] = lb[j];
                wave_bitonic_sort_multi<PKD, K>(bests);
                // one rescan of the lane's slots serves ALL kinds: per kind the lane's best remaining key above the
                // kind's k-th entry; repeat while any kind still found one (rounds = max over kinds, not the sum)
                for (;;) {
                    K thr[PKD], cb[PKD];
                    int ci[PKD];
#pragma unroll
                    for (int j = 0; j < PKD; ++j) {
                        thr[j] = kshfl(bests[j], a.k - 1);
                        kclear(cb[j]);
                        ci[j] = -1;
                    }
#pragma unroll 1
                    for (int q = 0; q < MPL; ++q) {
                        K kk[PKD];
                        slot_keys(q * THREADS + threadIdx.x, kk);
#pragma unroll
                        for (int j = 0; j < PKD; ++j)
                            if (!((done[j] >> q) & 1u) && kvalid(kk[j]) && kbetter(kk[j], thr[j]) &&
                                (ci[j] < 0 || kbetter(kk[j], cb[j]))) { cb[j] = kk[j]; ci[j] = q; }
                    }
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < PKD; ++j) {
                        if (j >= a.nk) continue;
                        const bool qual = ci[j] >= 0;
                        if (__ballot(qual) == 0) continue;
                        any = true;
                        if (qual) done[j] |= 1u << ci[j];
                        wave_topk_push(bests[j], cb[j], a.k);
                    }
                    if (!any) break;
                }
#pragma unroll
                for (int j = 0; j < PKD; ++j)
                    if (j < a.nk) emit(j, bests[j]);
            }
            }
        }
        } else {
        // ================= multi-wave bins: walk the dense list of occupied slots =================
        const uint32_t nocc = s_wcnt[NW > 1 ? wid : 0];               // this wave's region: the keys IT entered
        bool dense = true;                                            // else: some wave entered more keys than its region holds
#pragma unroll
        for (int w = 0; w < NW; ++w) dense = dense && s_wcnt[NW > 1 ? w : 0] <= (uint32_t)RCAP;
        // HEAVY FIRST. A key whose counters are exactly ONE click record ("light": 76 % of the keys of an M aid, 56 % of a
        // heavy aid's) has, for every kind of the pass, a smaller weight than any other key (two records, or one cart / order
        // record: checked on the host, `hot_ok`), so with at least k heavy keys the top-k of every kind consists of heavy
        // keys only. The wave reorders ITS region of the occupied list -- heavy slots first, light slots behind, nothing
        // dropped: the table is still cleared through the whole list -- and the top-k walks visit the heavy part only
        // (a walk costs ~35 vector instructions per 64 entries and kind pass, this reordering ~10). Fewer than k heavy keys in
        // the whole table (never at OTTO shape): the walks are redone over the full regions.
        uint32_t nwalk = nocc;
        bool hot = false;
        if (HOT && dense && a.hot_ok) {
            constexpr int NITMAX = (RCAP + 63) / 64;
            uint32_t ent[NITMAX];
            uint32_t hv = 0, vv = 0;
            const int nq = (int)((nocc + 63u) / 64u);                 // uniform in the wave
#pragma unroll
            for (int q = 0; q < NITMAX; ++q) {
                const uint32_t idx = (uint32_t)q * 64u + lane;
                ent[q] = 0;
                if (q < nq && idx < nocc) {
                    const uint32_t sl = s_occ[NW > 1 ? wid * RCAP + idx : 0];
                    ent[q] = sl;
                    vv |= 1u << q;
                    bool light;
                    if (PACKED) light = (s_tab[PACKED ? sl : 0] & 0xFFFFFFFFFull) == 1ull;
                    else light = s_v[0][PACKED ? 0 : sl] == 1u && (s_v[1][PACKED ? 0 : sl] | s_v[2][PACKED ? 0 : sl]) == 0u;
                    if (!light) hv |= 1u << q;
                }
            }
            uint32_t H = 0;
#pragma unroll
            for (int q = 0; q < NITMAX; ++q)
                if (q < nq) H += (uint32_t)__popcll(__ballot((hv >> q) & 1u));
            uint32_t hpos = 0, lpos = H;
#pragma unroll
            for (int q = 0; q < NITMAX; ++q) {
                if (q < nq) {
                    const bool v = (vv >> q) & 1u, h = (hv >> q) & 1u;
                    const uint64_t mh = __ballot(h), ml = __ballot(v && !h);
                    if (v) {
                        const uint64_t m = h ? mh : ml;
                        const uint32_t pos = (h ? hpos : lpos) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        s_occ[NW > 1 ? wid * RCAP + pos : 0] = (uint16_t)ent[q];
                    }
                    hpos += (uint32_t)__popcll(mh);
                    lpos += (uint32_t)__popcll(ml);
                }
            }
            if (lane == 0) s_hcnt[NW > 1 ? wid : 0] = H;
            wave_lds_sync();
            nwalk = H;
            hot = true;
        }
        if (ovf) {
            // LDS table full: ask the host to redo this aid with twice the partitions
            if (threadIdx.x == 0) {
                a.flag[x] = 1;
                a.boost[x] = (uint8_t)(lgR - l_log2r(a.cnt64[x], 0, a.l_cap, a.allow_packed) + 1);
                atomicAdd(a.ovf_count, 1u);
            }
        } else if (!(DBG && (a.debug_skip & 2))) {
            constexpr int MPL = T / THREADS;
            int nit = dense ? (int)((nwalk + 63u) / 64u) : MPL;                       // entries per lane (uniform in the wave)
            auto slot_at = [&](int q) -> int {                                        // q-th entry of this lane, -1: none
                if (!dense) return q * THREADS + (int)threadIdx.x;
                const uint32_t idx = (uint32_t)q * 64u + lane;
                return idx < nwalk ? (int)s_occ[NW > 1 ? wid * RCAP + idx : 0] : -1;
            };
            // key of ONE kind (weight vector c0, c1, c2 of that kind, hoisted by the caller) of table slot i
            auto slot_key1 = [&](int i, uint32_t c0, uint32_t c1, uint32_t c2) {
                uint32_t y, v0, v1, v2;
                if (PACKED) {
                    const uint64_t v = s_tab[PACKED ? i : 0];
                    y = v == TAB_EMPTY ? KEY_EMPTY : (uint32_t)(v >> 36);
                    v0 = (uint32_t)v & 0xFFFu; v1 = (uint32_t)(v >> 12) & 0xFFFu; v2 = (uint32_t)(v >> 24) & 0xFFFu;
                } else {
                    y = s_key[PACKED ? 0 : i];
                    v0 = s_v[0][PACKED ? 0 : i]; v1 = s_v[1][PACKED ? 0 : i]; v2 = s_v[2][PACKED ? 0 : i];
                }
                uint64_t uw = 0, qw = 0;
                if (y != KEY_EMPTY) {
                    if (TP) uw = (uint64_t)v0 | ((uint64_t)v1 << 12) | ((uint64_t)v2 << 24);
                    else if (GROUP == OTTO_COVIS_GROUP_TIME) qw = 65536ull * v0 + (((uint64_t)v2 << 32) | v1);
                    else if (PACKED) uw = __umul24(v0, c0) + __umul24(v1, c1) + __umul24(v2, c2);
                    else uw = (uint64_t)v0 * c0 + (uint64_t)v1 * c1 + (uint64_t)v2 * c2;
                }
                K r;
                kmake(r, uw, qw, y);
                return r;
            };
            // rank r of kind j -> output row (unpartitioned aid) or this partition's partial list; nv valid keys in the wave
            auto emit_ranked = [&](int j, K key, uint32_t rank, int nv) {
                const bool put = kvalid(key) && rank < (uint32_t)a.k;
                if (lgR == 0) {
                    const size_t o = ((size_t)(a.kind_base + j) * a.n_aids + x) * (size_t)a.k + rank;
                    if (put) { a.out_y[o] = kaid(key); a.out_w[o] = kw(key); }
                    if (lane == 0) a.out_n[(size_t)(a.kind_base + j) * a.n_aids + x] = nv < a.k ? nv : a.k;
                } else {
                    const size_t base = ((size_t)it * a.nk + j) * (size_t)a.k;
                    if (put) kstore(key, &a.part_w[base + rank], &a.part_y[base + rank]);
                    if ((int)lane < a.k && (int)lane >= nv) { K z; kclear(z); kstore(z, &a.part_w[base + lane], &a.part_y[base + lane]); }
                }
            };
            // Kind j from its candidate list s_exw[j][0, n), n <= 64 (every key at or above the threshold): rank = number
            // of better candidates, read back with uniform addresses -- no sorting network, every iteration independent.
            // Keys are distinct (one per aid_y), so the ranks are a permutation.
            auto finish_list = [&](int j, uint32_t n, bool store_tau) {
                K key;
                kclear(key);
                if (lane < n) kload(key, s_exw[j][lane], s_exy[0][WIDE ? lane : 0]);
                uint32_t rank = 0;
#pragma unroll 4
                for (uint32_t i = 0; i < n; ++i) {
                    K o;
                    kload(o, s_exw[j][i], s_exy[0][WIDE ? i : 0]);
                    kcount_better(rank, o, key);
                }
                emit_ranked(j, key, rank, (int)n);
                if (store_tau && n >= 32u && kvalid(key) && rank == 31u) ktau_store(key, a.tau_w, a.tau_y, (size_t)j * a.n_aids + x);
            };
            // exact top-k of kind j by ONE wave (two walks over the list / table): only when a candidate list overflowed
            auto wave_exact_topk = [&](int j) {
                const uint32_t c0 = a.coef[j][0], c1 = a.coef[j][1], c2 = a.coef[j][2];
                // entry e of the walk: region by region (dense) or slot by slot
                const int nreg = dense ? NW : 1;
                K lb;
                kclear(lb);
                int bi = -1;
                for (int w = 0; w < nreg; ++w) {
                    const int total = dense ? (int)(hot ? s_hcnt[NW > 1 ? w : 0] : s_wcnt[NW > 1 ? w : 0]) : T;
                    for (int i = (int)lane; i < total; i += 64) {
                        const int sl = dense ? (int)s_occ[NW > 1 ? w * RCAP + i : 0] : i;
                        const K key = slot_key1(sl, c0, c1, c2);
                        if (kbetter(key, lb)) { lb = key; bi = sl; }
                    }
                }
                K best = lb;
                wave_bitonic_sort_desc(best);
                for (int w = 0; w < nreg; ++w) {
                    const int total = dense ? (int)(hot ? s_hcnt[NW > 1 ? w : 0] : s_wcnt[NW > 1 ? w : 0]) : T;
                    for (int i0 = 0; i0 < total; i0 += 64) {
                        const int i = i0 + (int)lane;
                        K key;
                        kclear(key);
                        if (i < total) {
                            const int sl = dense ? (int)s_occ[NW > 1 ? w * RCAP + i : 0] : i;
                            if (sl != bi) key = slot_key1(sl, c0, c1, c2);
                        }
                        wave_topk_push(best, key, a.k);
                    }
                }
                emit(j, best);
            };
            auto append = [&](int j, K key) {
                const uint32_t pos = atomicAdd(&s_nex[j], 1u);
                if (pos < (uint32_t)EXCAP) kstore(key, &s_exw[j][pos], &s_exy[0][WIDE ? pos : 0]);
                else s_more = 1;
            };

            // ---- partitions of a heavy aid whose sibling left a threshold guess: ONE walk collects every key above the
            //      guess; at least k per kind and no overflow: the ranked lists are the exact top-k ----
            bool fast_done = false;
            bool pre_issued = false;
            // ---- few heavy keys (the M bin: ~170 of an aid's ~600 keys): the wave of kind j selects ALONE from all of them --
            //      its keys in registers (up to SHR per lane), lane-bests, k-th lane-best by counting = threshold, ballot-
            //      compacted candidates, ranks by counting. One barrier instead of the four of P1 .. P4, no LDS atomics,
            //      a third of their instructions.
            constexpr bool SH = HOT && NW <= 4;
            constexpr int SHR = 4;
            if (SH && hot && a.hot_ok >= 2) {
                __syncthreads();                             // s_hcnt of every wave
                uint32_t pre[NW + 1];
                pre[0] = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) pre[w + 1] = pre[w] + s_hcnt[NW > 1 ? w : 0];
                const uint32_t th = pre[NW];
                if (th < (uint32_t)a.k) {                    // fewer than k heavy keys: every key is walked
                    hot = false;
                    nwalk = nocc;
                    nit = (int)((nwalk + 63u) / 64u);
                } else if (th <= 64u * SHR) {
#ifdef OTTO_PHASE_PROF
                    if (threadIdx.x == 0) ph[14]++;
#endif
                    if (wid < a.nk && wid < PKD) {
                        const int j = wid;
                        const uint32_t c0 = a.coef[j][0], c1 = a.coef[j][1], c2 = a.coef[j][2];
                        K keys[SHR], lbk;
                        kclear(lbk);
#pragma unroll
                        for (int q = 0; q < SHR; ++q) {
                            const uint32_t e = (uint32_t)q * 64u + lane;
                            kclear(keys[q]);
                            if (e < th) {
                                int w = 0;
#pragma unroll
                                for (int ww = 1; ww < NW; ++ww)
                                    if (e >= pre[ww]) w = ww;
                                const uint32_t sl = s_occ[NW > 1 ? w * RCAP + (int)(e - pre[w]) : 0];
                                keys[q] = slot_key1((int)sl, c0, c1, c2);
                                if (kbetter(keys[q], lbk)) lbk = keys[q];
                            }
                        }
                        kstore(lbk, &s_exw[j][lane], &s_exy[0][WIDE ? lane : 0]);
                        wave_lds_sync();
                        uint32_t rank = 0;
#pragma unroll 8
                        for (int i = 0; i < 64; ++i) {
                            K o;
                            kload(o, s_exw[j][i], s_exy[0][WIDE ? i : 0]);
                            kcount_better(rank, o, lbk);
                        }
                        const uint64_t mk = __ballot(kvalid(lbk) && rank == (uint32_t)(a.k - 1));
                        K thr;
                        kclear(thr);
                        if (mk) thr = kshfl(lbk, __ffsll((unsigned long long)mk) - 1);
                        wave_lds_sync();
                        uint32_t n = 0;
#pragma unroll
                        for (int q = 0; q < SHR; ++q) {
                            const bool c = kvalid(keys[q]) && !kbetter(thr, keys[q]);
                            const uint64_t m = __ballot(c);
                            const uint32_t pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            if (c && pos < (uint32_t)EXCAP) kstore(keys[q], &s_exw[j][pos], &s_exy[0][WIDE ? pos : 0]);
                            n += (uint32_t)__popcll(m);
                        }
                        wave_lds_sync();
                        if (n > (uint32_t)EXCAP) wave_exact_topk(j);
                        else finish_list(j, n, false);
                    }
                    fast_done = true;
                }
            }
            bool gv = use_guess && !fast_done;
#pragma unroll
            for (int j = 0; j < PKD; ++j)
                if (j < a.nk && !kvalid(guess[j])) gv = false;
            if (gv) {                                    // uniform: every thread loaded the same words
                if (threadIdx.x < PK) s_nex[threadIdx.x] = 0;
                if (threadIdx.x == 0) { s_more = 0; if (WARM) s_nxt = nx; }
                __syncthreads();
                // the next item's bucket is touched now, one 64-byte line per thread, so that the prefetch below (which must wait
                // for the last walk over the table: it holds BU registers per lane) finds it in L2 instead of HBM -- the records
                // of a partition arrived 16 % of the kernel's time after they were asked for
                uint32_t warm = 0;
                if (WARM && s_nxt.it != 0xFFFFFFFFu && (s_nxt.item >> 50) != 0 && a.pstart != nullptr) {
                    const uint64_t wi = s_nxt.ps + (uint64_t)threadIdx.x * 16u;
                    if (wi < s_nxt.pe) warm = a.prec[wi];
                }
                for (int q = 0; q < nit; ++q) {
                    const int sl = slot_at(q);
                    if (sl < 0) continue;
                    K kk[PKD];
                    slot_keys(sl, kk);
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (j < a.nk && kvalid(kk[j]) && kbetter(kk[j], guess[j])) append(j, kk[j]);
                }
                asm volatile("" :: "v"(warm));            // the touch is a real load: its value is dropped here
                __syncthreads();
                prefetch_next();                         // last walk over the table done: request the next item's records
                pre_issued = true;
                bool ok = s_more == 0;
#pragma unroll
                for (int j = 0; j < PKD; ++j)
                    if (j < a.nk && s_nex[j] < (uint32_t)a.k) ok = false;
#ifdef OTTO_PHASE_PROF
                if (threadIdx.x == 0) { ph[8]++; if (ok) ph[9]++; else if (s_more) ph[11]++; else ph[10]++; }
#endif
                if (ok) {
                    if (wid < a.nk && wid < PKD) finish_list(wid, s_nex[wid], true);
                    fast_done = true;
                }
                __syncthreads();                         // lists are reused by the two-pass path / the next item
            }
            OTTO_PH(7);
            if (!fast_done) {
                // ---- P1 every lane: its best key per kind -> the key's SLOT to LDS. P2 wave j: best of each group of NW
                //      lanes, one 64-lane sort, k-th = threshold of kind j (a lower bound of the k-th best key: the group
                //      bests are a subset). P3 every lane: keys at or above the threshold -> candidate list of the kind
                //      (about k of them). P4 wave j: rank the list. ----
                K lb[PKD];
                int bi[PKD];
                // the keys of the lane's first NCK entries stay in registers for P3 (the dense list gives a lane 2 - 4 entries
                // in the common case: P3 then recomputes nothing)
                constexpr int NCK = (PACKED && THREADS <= 512) ? 3 : 0;
                K ck[NCK > 0 ? NCK : 1][PKD];
                for (;;) {
#pragma unroll
                for (int j = 0; j < PKD; ++j) { kclear(lb[j]); bi[j] = 0xFFFF; }
#pragma unroll
                for (int q = 0; q < NCK; ++q) {
                    const int sl = q < nit ? slot_at(q) : -1;
#pragma unroll
                    for (int j = 0; j < PKD; ++j) kclear(ck[q][j]);
                    if (sl >= 0) {
                        slot_keys(sl, ck[q]);
#pragma unroll
                        for (int j = 0; j < PKD; ++j)
                            if (kbetter(ck[q][j], lb[j])) { lb[j] = ck[q][j]; bi[j] = sl; }
                    }
                }
                for (int q = NCK; q < nit; ++q) {
                    const int sl = slot_at(q);
                    if (sl < 0) continue;
                    K kk[PKD];
                    slot_keys(sl, kk);
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (kbetter(kk[j], lb[j])) { lb[j] = kk[j]; bi[j] = sl; }
                }
#pragma unroll
                for (int j = 0; j < PKD; ++j)
                    if (j < a.nk) s_lbi[j][NW > 1 ? threadIdx.x : 0] = (uint16_t)bi[j];
                if (threadIdx.x < PK) s_nex[threadIdx.x] = 0;
                if (threadIdx.x == 0) s_more = 0;
                __syncthreads();
                if (hot) {                               // block-uniform: fewer than k heavy keys in the table -> walk every key
                    uint32_t th = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) th += s_hcnt[NW > 1 ? w : 0];
                    if (th < (uint32_t)a.k) {
                        hot = false;
                        nwalk = nocc;
                        nit = (int)((nwalk + 63u) / 64u);
                        continue;
                    }
                }
                break;
                }
                OTTO_PH(4);
                if (wid < a.nk && wid < PKD) {
                    K gb;
                    kclear(gb);
                    const uint32_t c0 = a.coef[wid][0], c1 = a.coef[wid][1], c2 = a.coef[wid][2];
#pragma unroll
                    for (int q = 0; q < NW; ++q) {
                        const uint32_t sl = s_lbi[wid][NW > 1 ? q * 64 + lane : 0];
                        if (sl != 0xFFFFu) {
                            const K o = slot_key1((int)sl, c0, c1, c2);
                            if (kbetter(o, gb)) gb = o;
                        }
                    }
                    // k-th best of the 64 group bests by counting (rank = number of better keys, read back with uniform
                    // addresses: 64 independent iterations instead of the 21 dependent stages of a sorting network)
                    kstore(gb, &s_exw[wid][lane], &s_exy[0][WIDE ? lane : 0]);
                    wave_lds_sync();
                    uint32_t rank = 0;
#pragma unroll 8
                    for (int i = 0; i < 64; ++i) {
                        K o;
                        kload(o, s_exw[wid][i], s_exy[0][WIDE ? i : 0]);
                        kcount_better(rank, o, gb);
                    }
                    const uint64_t mk = __ballot(kvalid(gb) && rank == (uint32_t)(a.k - 1));
                    K thr;
                    kclear(thr);
                    if (mk) thr = kshfl(gb, __ffsll((unsigned long long)mk) - 1);
                    if (lane == 0) kstore(thr, &s_thrw[wid], &s_thry[wid]);
                    // rank 31: a lower bound of the partition's 32nd best key = the guess for the aid's other partitions
                    if (use_guess && kvalid(gb) && rank == 31u) ktau_store(gb, a.tau_w, a.tau_y, (size_t)wid * a.n_aids + x);
                    wave_lds_sync();
                }
                OTTO_PH(5);
                __syncthreads();
                K thr[PKD];
#pragma unroll
                for (int j = 0; j < PKD; ++j) kload(thr[j], s_thrw[j], s_thry[j]);
#pragma unroll
                for (int q = 0; q < NCK; ++q) {
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (j < a.nk && kvalid(ck[q][j]) && !kbetter(thr[j], ck[q][j])) append(j, ck[q][j]);
                }
                for (int q = NCK; q < nit; ++q) {
                    const int sl = slot_at(q);
                    if (sl < 0) continue;
                    K kk[PKD];
                    slot_keys(sl, kk);
#pragma unroll
                    for (int j = 0; j < PKD; ++j)
                        if (j < a.nk && kvalid(kk[j]) && !kbetter(thr[j], kk[j])) append(j, kk[j]);
                }
                if (PREF && !pre_issued && threadIdx.x == 0) s_nxt = nx;
                __syncthreads();
                if (!pre_issued) prefetch_next();
                OTTO_PH(6);
                if (wid < a.nk && wid < PKD) {
                    if (s_more) wave_exact_topk(wid);
                    else finish_list(wid, s_nex[wid] < (uint32_t)EXCAP ? s_nex[wid] : (uint32_t)EXCAP, false);
                }
                OTTO_PH(7);
            }
        }
        // ---- clear the table for the next item: through the list when it is complete ----
        __syncthreads();
        if (!(DBG && (a.debug_skip & 4))) {
            if (dense) {
                for (uint32_t idx = lane; idx < nocc; idx += 64u) clear_slot((int)s_occ[NW > 1 ? wid * RCAP + idx : 0]);
            } else {
                clear_table();
            }
        }
        if (threadIdx.x == 0) s_ovf = 0;
        }   // multi-wave bins

        // ---- hand the prefetched next item over ----
        if (DYNAMIC) {
            __syncthreads();                       // every thread is done with s_cur's consumers and the table
            if (threadIdx.x == 0) {
                s_cur = nx;
                idx_next = idx_far;                                   // dequeued one iteration ago
                idx_far = take();                                     // consumed one iteration from now
            }
        } else {
            cur = nx;
            sidx += gridDim.x;
        }
    }
#ifdef OTTO_PHASE_PROF
    if (threadIdx.x == 0 && a.prof)
        for (int i = 0; i < 16; ++i) atomicAdd(&a.prof[i], ph[i]);
#endif
}

// merge the R partial top-k lists of one heavy aid (item with part == 0 and R > 1): one workgroup per item, ONE WAVE PER KIND
// (the largest aid's 5,000 - 10,000 candidates per kind are one wave's serial work: the kernel lasts as long as that aid),
// the next 64 candidates requested before the current ones are merged
template <typename K, bool RAW>
__device__ __forceinline__ void merge_lists(const ReduceArgs& a, uint32_t it, uint32_t x, int lgR, int j, unsigned lane) {
    const uint64_t ncand = (uint64_t)a.k << lgR;
    auto fetch = [&](uint64_t c0) {
        const uint64_t c = c0 + lane;
        K cand;
        kclear(cand);
        if (c < ncand) {
            const size_t o = ((size_t)(it + c / a.k) * a.nk + j) * (size_t)a.k + (c % a.k);
            kload(cand, a.part_w[o], a.part_y[o]);
        }
        return cand;
    };
    K best;
    kclear(best);
    K cand = fetch(0);
    for (uint64_t c0 = 0; c0 < ncand; c0 += 64) {
        const K nxt = c0 + 64 < ncand ? fetch(c0 + 64) : cand;
        // the first 64 candidates are sorted at once (into an empty list every one of them would be pushed one by one: 64
        // serial insertions per kind and aid); later batches push the few that beat the k-th
        if (c0 == 0) { best = cand; wave_bitonic_sort_desc(best); }
        else wave_topk_push(best, cand, a.k);
        cand = nxt;
    }
    const bool valid = (int)lane < a.k && kvalid(best);
    const size_t o = ((size_t)(a.kind_base + j) * a.n_aids + x) * (size_t)a.k + lane;
    if (valid) {
        a.out_y[o] = kaid(best);
        if constexpr (RAW) a.out_w[o] = best.c >> REC_AID_BITS;      // packed time-weighted keys carry the Q16 weight itself
        else a.out_w[o] = kweight(best);
    }
    const int nvalid = __popcll(__ballot(valid));
    if (lane == 0) a.out_n[(size_t)(a.kind_base + j) * a.n_aids + x] = nvalid;
}

template <int GROUP>
__global__ __launch_bounds__(64 * PK) void k_merge(ReduceArgs a) {
    const uint32_t it = blockIdx.x;
    if (it >= a.n_items) return;
    const uint64_t item = a.items[it];
    const uint32_t part = (uint32_t)((item >> 26) & 0xFFFFFFu);
    const int lgR = (int)(item >> 50);
    if (part != 0 || lgR == 0) return;
    const uint32_t x = (uint32_t)(item & REC_AID_MASK);
    if (a.flag[x]) return;
    const unsigned lane = lane_id();
    const int j = (int)(threadIdx.x >> 6);
    if (j >= a.nk) return;
    if constexpr (GROUP == OTTO_COVIS_GROUP_TIME) {
        // the aid's partitions wrote 64-bit packed keys (packed layouts) or wide (weight, aid) keys (wide layout)
        if (heavy_mode(a.cnt64[x], a.allow_packed, a.l_cap) != 0) merge_lists<KeyN, true>(a, it, x, lgR, j, lane);
        else merge_lists<KeyW, false>(a, it, x, lgR, j, lane);
    } else {
        merge_lists<KeyN, false>(a, it, x, lgR, j, lane);
    }
}

// ---------------------------------------------------------------------------
// multi-GPU exchange helpers
// ---------------------------------------------------------------------------
struct ExportRuns {   // 1 if run slot i leaves for [lo, hi)
    const uint32_t* run_x;
    const uint64_t* run_desc;
    uint32_t lo, hi;
    __device__ uint64_t operator()(int64_t i) const {
        const uint32_t x = run_x[i];
        return (desc_len(run_desc[i]) && x >= lo && x < hi) ? 1ull : 0ull;
    }
};
struct ExportRecs {
    const uint32_t* run_x;
    const uint64_t* run_desc;
    uint32_t lo, hi;
    __device__ uint64_t operator()(int64_t i) const {
        const uint32_t x = run_x[i];
        const uint64_t len = desc_pairs(run_desc[i]);               // exported runs are private rows: pairs only
        return (len && x >= lo && x < hi) ? len : 0ull;
    }
};

__global__ void k_export(const uint32_t* run_x, const uint64_t* run_desc, int64_t n_slots, uint32_t lo, uint32_t hi,
                         const uint64_t* run_pos, const uint64_t* rec_pos, const uint32_t* rec, const uint32_t* tw,
                         uint32_t* o_hdr, uint32_t* o_rec, uint32_t* o_tw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t d = run_desc[i];
        const uint32_t len = desc_pairs(d), sp = desc_sp(d);
        const uint32_t x = run_x[i];
        if (len && x >= lo && x < hi) {
            const uint64_t rp = run_pos[i], cp = rec_pos[i], off = desc_slot(d);
            o_hdr[2 * rp] = x;
            o_hdr[2 * rp + 1] = len;
            for (uint32_t t = 0; t < len; ++t) {
                const uint32_t pos = t >= sp ? t + 1 : t;              // shared list: skip the run's own entry
                o_rec[cp + t] = rec[off + pos];
                if (o_tw) o_tw[cp + t] = tw[off + (sp != DESC_SP_NONE ? sp : pos)];
            }
        }
    }
}

// ---- one-pass export of every owner's piece (owner-major in ONE buffer, ready to be the all-to-all send buffer) ----
constexpr int MAX_OWNERS = 64;
struct OwnerArgs {
    const uint32_t* run_x;
    const uint64_t* run_desc;
    int64_t slot0;                       // first run slot of the range this call covers (chunked export)
    int64_t n_slots;                     // run slots of the range
    int n_owners;
    uint32_t bounds[MAX_OWNERS + 1];     // owner o holds aid_x in [bounds[o], bounds[o+1])
    uint64_t run_base[MAX_OWNERS];       // fill: first run / record of owner o inside the output buffers
    uint64_t rec_base[MAX_OWNERS];
    unsigned long long* totals;          // [MAX_OWNERS] runs << 36 | records per owner (plan: totals; fill: running cursor).
                                         // ONE word so a block's run range and record range are reserved by one atomic
    const uint32_t* rec;
    const uint32_t* tw;
    uint32_t* o_hdr;
    uint32_t* o_rec;
    uint32_t* o_tw;
};

__device__ __forceinline__ int owner_of(const OwnerArgs& a, uint32_t x) {
    int o = 0;
    while (o + 1 < a.n_owners && x >= a.bounds[o + 1]) ++o;
    return o;
}

constexpr int EXP_REC_BITS = 34;      // export totals / cursors per owner: runs << 34 | records (2^30 runs, 2^34 records)

// plan: runs and records per owner. A streaming reduction: LDS accumulators live across the block's whole grid-stride
// loop, one global atomic per owner and block at the end.
__global__ __launch_bounds__(256) void k_export_plan(OwnerArgs a) {
    __shared__ unsigned long long s_tot[MAX_OWNERS];
    if (threadIdx.x < MAX_OWNERS) s_tot[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_slots; i += (int64_t)gridDim.x * 256) {
        const uint64_t d = a.run_desc[a.slot0 + i];
        const uint32_t len = desc_pairs(d);
        if (len) atomicAdd(&s_tot[owner_of(a, a.run_x[a.slot0 + i])], (1ull << EXP_REC_BITS) | len);
    }
    __syncthreads();
    if ((int)threadIdx.x < a.n_owners && s_tot[threadIdx.x]) atomicAdd(&a.totals[threadIdx.x], s_tot[threadIdx.x]);
}

// fill: chunks of EXP_CHUNK run slots per workgroup round. Block-local LDS ranks (one packed word hands a run its rank
// AND its record offset, so both orders agree), one global cursor bump per chunk and owner, headers written by the
// run's thread, records copied by one 32-lane half-wave per run (contiguous reads and writes).
constexpr int EXP_CHUNK = 512;
__global__ __launch_bounds__(256) void k_export_fill(OwnerArgs a) {
    constexpr int PER = EXP_CHUNK / 256;
    __shared__ unsigned long long s_rr[MAX_OWNERS];      // runs << 32 | records of this chunk per owner
    __shared__ unsigned long long s_rbase[MAX_OWNERS], s_cbase[MAX_OWNERS];
    __shared__ uint64_t s_src[EXP_CHUNK], s_dst[EXP_CHUNK];
    __shared__ uint8_t s_len[EXP_CHUNK], s_sp[EXP_CHUNK];
    const int64_t n_chunks = (a.n_slots + EXP_CHUNK - 1) / EXP_CHUNK;
    for (int64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        if (threadIdx.x < MAX_OWNERS) s_rr[threadIdx.x] = 0;
        __syncthreads();
        uint64_t d[PER];
        uint32_t x[PER], my_run[PER], my_rec[PER];
        int o[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = ch * EXP_CHUNK + (int64_t)u * 256 + threadIdx.x;
            d[u] = i < a.n_slots ? a.run_desc[a.slot0 + i] : 0ull;
            x[u] = 0; o[u] = 0; my_run[u] = my_rec[u] = 0;
            if (desc_len(d[u])) x[u] = a.run_x[a.slot0 + i];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t len = desc_pairs(d[u]);
            if (len) {
                o[u] = owner_of(a, x[u]);
                const unsigned long long old = atomicAdd(&s_rr[o[u]], (1ull << 32) | len);
                my_run[u] = (uint32_t)(old >> 32);
                my_rec[u] = (uint32_t)old;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < a.n_owners) {
            const unsigned long long r = s_rr[threadIdx.x] >> 32, c = s_rr[threadIdx.x] & 0xFFFFFFFFull;
            unsigned long long rb = 0, cb = 0;
            if (r) {
                const unsigned long long old = atomicAdd(&a.totals[threadIdx.x], (r << EXP_REC_BITS) | c);
                rb = old >> EXP_REC_BITS;
                cb = old & ((1ull << EXP_REC_BITS) - 1ull);
            }
            s_rbase[threadIdx.x] = rb;
            s_cbase[threadIdx.x] = cb;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int r = u * 256 + threadIdx.x;
            const uint32_t len = desc_pairs(d[u]);                   // exported runs are private rows: pairs only
            s_len[r] = (uint8_t)len;
            s_sp[r] = (uint8_t)desc_sp(d[u]);
            if (len) {
                const uint64_t rp = a.run_base[o[u]] + s_rbase[o[u]] + my_run[u];
                const uint64_t cp = a.rec_base[o[u]] + s_cbase[o[u]] + my_rec[u];
                *reinterpret_cast<uint2*>(&a.o_hdr[2 * rp]) = make_uint2(x[u], len);
                s_src[r] = desc_slot(d[u]);
                s_dst[r] = cp;
            }
        }
        __syncthreads();
        const int hw = threadIdx.x >> 5, l = threadIdx.x & 31;
        constexpr int CU = 8;                                  // runs in flight per half-wave
        for (int r0 = hw * CU; r0 < EXP_CHUNK; r0 += 8 * CU) {
            uint32_t v[CU], w[CU];
            bool ok[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const uint32_t sp = s_sp[r0 + u];
                const uint32_t pos = (uint32_t)l >= sp ? l + 1 : l;     // shared list: skip the run's own entry
                ok[u] = (uint32_t)l < (uint32_t)s_len[r0 + u];
                v[u] = ok[u] ? a.rec[s_src[r0 + u] + pos] : 0u;
                w[u] = (ok[u] && a.o_tw) ? a.tw[s_src[r0 + u] + (sp != DESC_SP_NONE ? sp : pos)] : 0u;
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                if (ok[u]) {
                    a.o_rec[s_dst[r0 + u] + l] = v[u];
                    if (a.o_tw) a.o_tw[s_dst[r0 + u] + l] = w[u];
                }
            }
        }
        __syncthreads();
    }
}

// statistics only (otto_covis_stats): runs that read a shared component list, records kept in private rows
__global__ __launch_bounds__(256) void k_desc_totals(const uint64_t* run_desc, int64_t n_slots, unsigned long long* out) {
    unsigned long long shared = 0, rows = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * 256) {
        const uint64_t d = run_desc[i];
        const uint32_t len = desc_len(d);
        if (len) {
            if (desc_sp(d) != DESC_SP_NONE) shared += 1;
            else rows += len;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        shared += (unsigned long long)__shfl_xor((long long)shared, o, 64);
        rows += (unsigned long long)__shfl_xor((long long)rows, o, 64);
    }
    if (lane_id() == 0) {
        if (shared) atomicAdd(&out[0], shared);
        if (rows) atomicAdd(&out[1], rows);
    }
}

struct HdrLen {
    const uint32_t* hdr;
    __device__ uint64_t operator()(int64_t i) const { return hdr[2 * i + 1]; }
};

__global__ void k_import(const uint32_t* hdr, int64_t n_runs, const uint64_t* rec_pos, uint64_t rec_base,
                         uint64_t run_base, uint32_t* run_x, uint64_t* run_desc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (int64_t)gridDim.x * blockDim.x) {
        run_x[run_base + i] = hdr[2 * i + 1] ? hdr[2 * i] : RUN_X_EMPTY;
        run_desc[run_base + i] = make_desc(rec_base + rec_pos[i], hdr[2 * i + 1]);
    }
}

}  // namespace otto

// ===========================================================================
// host side
// ===========================================================================
using namespace otto;

struct otto_covis_ctx {
    otto_covis_params p;
    // K1 products
    DevBuf rec, tw, run_x, run_desc;
    uint64_t rec_used = 0;    // record slots
    unsigned long long desc_totals[2] = {0, 0};   // statistics: shared-list runs, private-row records (otto_covis_stats)
    bool desc_totals_valid = false;
    uint64_t run_used = 0;    // run slots
    int64_t sessions = 0;
    // chunk scratch
    DevBuf pair_base, ev_base, partial, cls_pos[N_WIN_CLASSES], sess_list, cls_byte;
    int fast_path = 1;
    int fused = 2;                 // no filter kind configured: 2 = k_expand_lists (component lists), 1 = k_expand_fused, 0 = class-sorted kernels
    // index
    bool index_valid = false;
    DevBuf cnt64, run_start, run_rank, sorted_desc, item_start, boost, flag, counters;
    DevBuf items[3];
    uint64_t n_items[3] = {0, 0, 0};
    uint64_t bin_pairs[3] = {0, 0, 0};
    uint64_t bin_runs[3] = {0, 0, 0};
    uint64_t n_pairs = 0, n_runs = 0;
    // partition pass of heavy aids
    DevBuf litem_start, chunks, pcount, pcursor, pstart, prec, ptw, item_part;
    uint64_t n_chunks = 0;
    int partition = 1;
    bool exact_round = false;      // set by finalize for the rounds that redo flagged aids
    int part_sized = 1;            // option "part_sized": capacity-sized buckets on the first attempt (no count pass); 0 = always counted
    int debug_skip = 0;
    // reduce scratch
    DevBuf part_y, part_w;
    DevBuf bcount, bstart, tmp_runs;          // bucketed index
    int bucket_index = 1;          // option "bucket_index": LDS-atomic index build (0 = global-atomic histogram)
    int s_wgs = 20;                // option "s_wgs": one-wave workgroups of the S bin per CU (A/B)
    int p_wgs = 4;                 // option "p_wgs": workgroups of the partition scatter per CU (4 are resident: 38 KB of LDS each; chunks are dequeued
                                   // dynamically, so the count no longer matters: 4 / 8 / 12 measure 2.92 ms; the static deal needed 12 for 2.99)
    int bkt_sh = 0;                // option "bkt_sh": log2 aids per index bucket (0 = from the aid space)
    DevBuf lorder[3][3], lrank, lmode_start;   // [bin][mode] processing order of the tiers / layouts (heavy: pilots first)
    uint64_t n_order[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int packed_heavy = 2;          // option "packed_heavy": packed layout for heavy aids with < 4096 runs (1: 2^14 tables for every aid
                                   // above l_cap records, 2: 2^14 only for the aids that fit one table, 2^13 partitions of l_cap beyond)
    int items_allow_packed = -1;   // layout rule the L item list was built with (-1: not built)
    DevBuf tau_w, tau_y;           // threshold guesses of partitioned heavy aids (per reduce pass)
    int guess = 1;                 // option "guess": single-pass top-k from a sibling partition's threshold
    hipStream_t side = nullptr;    // the partition pass of the heavy aids runs here, beside the S / M bins
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool part_early = false;       // the partition of this pass is in flight on `side`
    uint64_t items_gen = 1;        // bumped whenever the heavy item list / partition chunks are rebuilt
    uint64_t part_gen = 0;         // items_gen the partition buckets in prec / ptw were filled for (0: none)
    bool part_has_tw = false;      // ... with the time channel alongside
    int overlap_partition = 0;     // option "overlap_partition": measured at full OTTO shape, both kernels take twice as long side by
                                   // side (partition 3.8 -> 6.2 ms, reduce S 2.9 -> 6.3 ms: the S bin's 20 workgroups per CU leave the
                                   // partition workgroups no LDS), the step gains 0.4 ms of 31.5 and the per-kernel times stop adding up:
                                   // off by default
    int hot = 2;                   // option "hot": 1 = top-k walks of the multi-wave bins over the heavy keys only, 2 = + single-wave
                                   // selection when the heavy keys are few (M bin), 0 = off (A/B)
    DevBuf exp_run_pos, exp_rec_pos, exp_totals;
    uint64_t exp_n_runs[64] = {0}, exp_n_recs[64] = {0};
    int exp_planned = 0;
    int64_t retries = 0;
    uint32_t l_cap = L_CAP;
    std::string knames[OTTO_COVIS_T_COUNT];   // kernels launched under each timing slot since the last reset (otto_covis_kernel_names)
    hipEvent_t ev[2 * OTTO_COVIS_T_COUNT];
    bool ev_set[OTTO_COVIS_T_COUNT];
    bool ev_ok = false;
};

// remember the kernel (template instantiation) launched under timing slot i, once per distinct name
static void kname(otto_covis_ctx* c, int i, const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::string& d = c->knames[i];
    if (d.find(buf) != std::string::npos) return;
    if (!d.empty()) d += " + ";
    d += buf;
}
static void tbegin(otto_covis_ctx* c, int i, hipStream_t s) {
    if (c->ev_ok) { (void)hipEventRecord(c->ev[2 * i], s); }
}
static void tend(otto_covis_ctx* c, int i, hipStream_t s) {
    if (c->ev_ok) { (void)hipEventRecord(c->ev[2 * i + 1], s); c->ev_set[i] = true; }
}

extern "C" const char* otto_last_error(void) { return g_err.c_str(); }

extern "C" int otto_covis_create(otto_covis_ctx** out, const otto_covis_params* p) {
    OTTO_REQUIRE(out && p, "otto_covis_create: null argument");
    OTTO_REQUIRE(p->window >= 2 && p->window <= OTTO_COVIS_MAX_WINDOW, "window must be in [2, 32], got %d", p->window);
    OTTO_REQUIRE(p->max_gap >= 0, "max_gap must be >= 0");
    OTTO_REQUIRE(p->n_aids > 0 && p->n_aids <= OTTO_COVIS_MAX_AIDS, "n_aids must be in [1, 2^26], got %u", p->n_aids);
    OTTO_REQUIRE(p->n_filters >= 0 && p->n_filters <= OTTO_COVIS_MAX_FILTERS, "n_filters must be in [0, 4]");
    OTTO_REQUIRE(p->n_type_weights >= 0 && p->n_type_weights <= OTTO_COVIS_MAX_TYPE_WEIGHTS, "n_type_weights must be in [0, 4]");
    OTTO_REQUIRE(p->ts_max >= p->ts_min, "ts_max < ts_min");
    for (int j = 0; j < p->n_type_weights; ++j)
        for (int t = 0; t < 3; ++t)
            OTTO_REQUIRE(p->type_weight[j][t] > 0 && p->type_weight[j][t] < 256, "type_weight[%d][%d] must be in [1, 255]", j, t);
    for (int f = 0; f < p->n_filters; ++f) OTTO_REQUIRE(p->filter_mask[f] < 512, "filter_mask[%d] has bits above 8", f);
    otto_covis_ctx* c = new (std::nothrow) otto_covis_ctx();
    OTTO_REQUIRE(c, "out of host memory");
    c->p = *p;
    c->ev_ok = true;
    for (int i = 0; i < 2 * OTTO_COVIS_T_COUNT; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) c->ev_ok = false;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        c->side = nullptr;                                  // no side stream: the partition stays on the caller's stream
    }
    memset(c->ev_set, 0, sizeof c->ev_set);
    *out = c;
    return 0;
}

extern "C" void otto_covis_destroy(otto_covis_ctx* c) {
    if (!c) return;
    DevBuf* all[] = {&c->rec, &c->tw, &c->run_x, &c->run_desc, &c->pair_base, &c->ev_base, &c->partial, &c->cnt64,
                     &c->cls_pos[0], &c->cls_pos[1], &c->cls_pos[2], &c->cls_pos[3], &c->cls_pos[4], &c->cls_pos[5],
                     &c->sess_list, &c->cls_byte,
                     &c->run_start, &c->run_rank, &c->sorted_desc, &c->item_start, &c->boost, &c->flag, &c->counters,
                     &c->items[0], &c->items[1], &c->items[2], &c->part_y, &c->part_w, &c->tau_w, &c->tau_y, &c->lorder[1][0], &c->lorder[1][1], &c->lorder[1][2], &c->lorder[2][0], &c->lorder[2][1], &c->lorder[2][2], &c->lrank, &c->lmode_start, &c->bcount, &c->bstart, &c->tmp_runs, &c->exp_run_pos, &c->exp_rec_pos, &c->exp_totals,
                     &c->litem_start, &c->chunks, &c->pcount, &c->pcursor, &c->pstart, &c->prec, &c->ptw, &c->item_part};
    for (DevBuf* b : all) b->release();
    if (c->ev_ok)
        for (int i = 0; i < 2 * OTTO_COVIS_T_COUNT; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c;
}

extern "C" int otto_covis_reset(otto_covis_ctx* c) {
    OTTO_REQUIRE(c, "null ctx");
    c->rec_used = c->run_used = 0;
    c->sessions = 0;
    c->index_valid = false;
    c->desc_totals_valid = false;
    c->n_pairs = c->n_runs = 0;
    c->n_items[0] = c->n_items[1] = c->n_items[2] = 0;
    memset(c->ev_set, 0, sizeof c->ev_set);
    for (auto& n : c->knames) n.clear();
    return 0;
}

extern "C" int otto_covis_feed(otto_covis_ctx* c, const uint32_t* d_aid, const int32_t* d_ts, const uint8_t* d_type,
                               const int64_t* d_sess_off, int64_t n_sess, void* stream) {
    OTTO_REQUIRE(c && d_sess_off, "otto_covis_feed: null argument");
    OTTO_REQUIRE(n_sess >= 0, "n_sess < 0");
    if (n_sess == 0) return 0;
    OTTO_REQUIRE(d_aid && d_ts && d_type, "otto_covis_feed: null event arrays");
    hipStream_t s = (hipStream_t)stream;
    const otto_covis_params& p = c->p;

    tbegin(c, OTTO_COVIS_T_WINSCAN, s);
    OTTO_TRY(c->pair_base.ensure((size_t)(n_sess + 1) * 8, 0, s));
    OTTO_TRY(c->ev_base.ensure((size_t)(n_sess + 1) * 8, 0, s));
    OTTO_TRY(c->partial.ensure(scan_partial_bytes(n_sess > (int64_t)p.n_aids ? n_sess : (int64_t)p.n_aids), 0, s));
    OTTO_TRY(device_scan(WinPairs{d_sess_off, p.window}, n_sess, c->pair_base.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    OTTO_TRY(device_scan(WinEvents{d_sess_off, p.window}, n_sess, c->ev_base.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    OTTO_REQUIRE(n_sess < (1ll << 32), "more than 2^32 sessions in one chunk");
    const bool fused = p.n_filters == 0 && c->fused;
    if (fused) {
        // no filter kinds: one fused launch over the sessions in memory order (k_expand_fused), no class lists
        tend(c, OTTO_COVIS_T_WINSCAN, s);
        uint64_t tot[2];
        OTTO_HIP(hipMemcpyAsync(&tot[0], c->pair_base.as<uint64_t>() + n_sess, 8, hipMemcpyDeviceToHost, s));
        OTTO_HIP(hipMemcpyAsync(&tot[1], c->ev_base.as<uint64_t>() + n_sess, 8, hipMemcpyDeviceToHost, s));
        OTTO_HIP(hipStreamSynchronize(s));
        const uint64_t n_ev = tot[1];
        const bool lists = c->fused == 2;                  // component lists: n list slots per window behind the pair slots
        const uint64_t n_slots = tot[0] + (lists ? n_ev : 0ull);
        OTTO_REQUIRE(c->rec_used + n_slots < (1ull << DESC_SLOT_BITS), "record slot space exhausted");
        OTTO_TRY(c->rec.ensure((size_t)(c->rec_used + n_slots + REC_PAD) * 4, (size_t)c->rec_used * 4, s));
        if (p.want_time) OTTO_TRY(c->tw.ensure((size_t)(c->rec_used + n_slots) * 4, (size_t)c->rec_used * 4, s));
        OTTO_TRY(c->run_x.ensure((size_t)(c->run_used + n_ev) * 4, (size_t)c->run_used * 4, s));
        OTTO_TRY(c->run_desc.ensure((size_t)(c->run_used + n_ev) * 8, (size_t)c->run_used * 8, s));
        tbegin(c, OTTO_COVIS_T_EXPAND, s);
        ExpandArgs a;
        memset(&a, 0, sizeof(a));
        a.aid = d_aid; a.ts = d_ts; a.type = d_type; a.sess_off = d_sess_off;
        a.pair_base = c->pair_base.as<uint64_t>(); a.ev_base = c->ev_base.as<uint64_t>();
        a.rec = c->rec.as<uint32_t>(); a.tw = c->tw.as<uint32_t>();
        a.run_x = c->run_x.as<uint32_t>(); a.run_desc = c->run_desc.as<uint64_t>();
        a.rec_base = c->rec_used; a.run_base = c->run_used;
        a.list_base = c->rec_used + tot[0];
        a.window = p.window; a.max_gap = p.max_gap;
        a.t0 = p.ts_min; a.tspan = (int64_t)p.ts_max - (int64_t)p.ts_min;
        a.debug = c->debug_skip >> 4;
        const int64_t tiles = (n_sess + 63) / 64;
        const int64_t blocks = (tiles + 3) / 4;
        int per_cu = 0, n_cu = 0, dev = 0;
        OTTO_HIP(hipGetDevice(&dev));
        OTTO_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (lists) {
            if (p.want_time) OTTO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_expand_lists<true, false>, 256, 0));
            else OTTO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_expand_lists<false, false>, 256, 0));
        } else if (p.want_time) OTTO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_expand_fused<true, false>, 256, 0));
        else OTTO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_expand_fused<false, false>, 256, 0));
        const int64_t resident = (int64_t)(per_cu > 0 ? per_cu : 4) * (n_cu > 0 ? n_cu : 256);   // one round of resident workgroups
        const int grid = (int)(blocks < resident ? blocks : resident);
        kname(c, OTTO_COVIS_T_EXPAND, lists ? "k_expand_lists<%s, %s>" : "k_expand_fused<%s, %s>", p.want_time ? "true" : "false", a.debug ? "true" : "false");
        if (lists) {
            if (a.debug) {
                if (p.want_time) k_expand_lists<true, true><<<grid, 256, 0, s>>>(a, n_sess);
                else k_expand_lists<false, true><<<grid, 256, 0, s>>>(a, n_sess);
            } else if (p.want_time) k_expand_lists<true, false><<<grid, 256, 0, s>>>(a, n_sess);
            else k_expand_lists<false, false><<<grid, 256, 0, s>>>(a, n_sess);
        } else if (a.debug) {
            if (p.want_time) k_expand_fused<true, true><<<grid, 256, 0, s>>>(a, n_sess, c->fast_path);
            else k_expand_fused<false, true><<<grid, 256, 0, s>>>(a, n_sess, c->fast_path);
        } else if (p.want_time) k_expand_fused<true, false><<<grid, 256, 0, s>>>(a, n_sess, c->fast_path);
        else k_expand_fused<false, false><<<grid, 256, 0, s>>>(a, n_sess, c->fast_path);
        OTTO_HIP(hipGetLastError());
        tend(c, OTTO_COVIS_T_EXPAND, s);
        c->rec_used += n_slots;
        c->run_used += n_ev;
        c->sessions += n_sess;
        c->index_valid = false;
    c->desc_totals_valid = false;
        return 0;
    }
    // filter kinds: window classes (3 sizes x gap-free or not): one session list, six segments
    const int use_fast = p.n_filters == 0 && c->fast_path;
    OTTO_TRY(c->cls_byte.ensure((size_t)n_sess, 0, s));
    k_classify<<<(unsigned)((n_sess + 255) / 256), 256, 0, s>>>(d_sess_off, d_ts, p.window, p.max_gap, use_fast, n_sess,
                                                                 c->cls_byte.as<uint8_t>());
    OTTO_HIP(hipGetLastError());
    for (int cl = 0; cl < N_WIN_CLASSES; ++cl) {
        OTTO_TRY(c->cls_pos[cl].ensure((size_t)(n_sess + 1) * 8, 0, s));
        OTTO_TRY(device_scan(WinClass{c->cls_byte.as<uint8_t>(), cl}, n_sess, c->cls_pos[cl].as<uint64_t>(), c->partial.as<uint64_t>(), s));
    }
    tend(c, OTTO_COVIS_T_WINSCAN, s);
    uint64_t totals[2 + N_WIN_CLASSES];
    OTTO_HIP(hipMemcpyAsync(&totals[0], c->pair_base.as<uint64_t>() + n_sess, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipMemcpyAsync(&totals[1], c->ev_base.as<uint64_t>() + n_sess, 8, hipMemcpyDeviceToHost, s));
    for (int cl = 0; cl < N_WIN_CLASSES; ++cl)
        OTTO_HIP(hipMemcpyAsync(&totals[2 + cl], c->cls_pos[cl].as<uint64_t>() + n_sess, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    const uint64_t n_slots = totals[0], n_ev = totals[1];
    OTTO_REQUIRE(c->rec_used + n_slots < (1ull << DESC_SLOT_BITS), "record slot space exhausted");

    OTTO_TRY(c->rec.ensure((size_t)(c->rec_used + n_slots + REC_PAD) * 4, (size_t)c->rec_used * 4, s));
    if (p.want_time) OTTO_TRY(c->tw.ensure((size_t)(c->rec_used + n_slots) * 4, (size_t)c->rec_used * 4, s));
    OTTO_TRY(c->run_x.ensure((size_t)(c->run_used + n_ev) * 4, (size_t)c->run_used * 4, s));
    OTTO_TRY(c->run_desc.ensure((size_t)(c->run_used + n_ev) * 8, (size_t)c->run_used * 8, s));
    OTTO_TRY(c->sess_list.ensure((size_t)(n_sess + 1) * 4, 0, s));

    tbegin(c, OTTO_COVIS_T_EXPAND, s);
    ClassFill cf;
    uint64_t lb = 0;
    for (int cl = 0; cl < N_WIN_CLASSES; ++cl) {
        cf.pos[cl] = c->cls_pos[cl].as<uint64_t>();
        cf.base[cl] = lb;
        lb += totals[2 + cl];
    }
    k_fill_classes<<<(unsigned)((n_sess + 255) / 256), 256, 0, s>>>(c->cls_byte.as<uint8_t>(), n_sess, cf, c->sess_list.as<uint32_t>());
    OTTO_HIP(hipGetLastError());
    ExpandArgs a;
    a.aid = d_aid; a.ts = d_ts; a.type = d_type; a.sess_off = d_sess_off;
    a.pair_base = c->pair_base.as<uint64_t>(); a.ev_base = c->ev_base.as<uint64_t>();
    a.rec = c->rec.as<uint32_t>(); a.tw = c->tw.as<uint32_t>();
    a.run_x = c->run_x.as<uint32_t>(); a.run_desc = c->run_desc.as<uint64_t>();
    a.rec_base = c->rec_used; a.run_base = c->run_used;
    a.window = p.window; a.max_gap = p.max_gap;
    a.t0 = p.ts_min; a.tspan = (int64_t)p.ts_max - (int64_t)p.ts_min;
    for (int f = 0; f < 4; ++f) a.fmask[f] = f < p.n_filters ? p.filter_mask[f] : 0u;
    for (int cl = 0; cl < N_WIN_CLASSES; ++cl) {
        a.sess_list = c->sess_list.as<uint32_t>() + cf.base[cl];
        a.n_list = (int64_t)totals[2 + cl];
        if (a.n_list == 0) continue;
        const int size_cl = cl % 3;
        const bool fast = cl >= 3;
        const int wpw = size_cl == 0 ? 8 : (size_cl == 1 ? 4 : 2);
        const int64_t waves = (a.n_list + wpw - 1) / wpw;
        const int64_t blocks = (waves + 3) / 4;
        const int grid = (int)(blocks < 256 * 16 ? blocks : 256 * 16);
        const int variant = (p.want_time ? 2 : 0) | (p.n_filters > 0 ? 1 : 0);
#define OTTO_EXPAND(G)                                                                   \
        kname(c, OTTO_COVIS_T_EXPAND, fast ? "k_expand_fast<%d, %s>" : "k_expand<%d, %s, %s>", G, p.want_time ? "true" : "false", p.n_filters > 0 ? "true" : "false"); \
        if (fast) {                                                                      \
            if (p.want_time) k_expand_fast<G, true><<<grid, 256, 0, s>>>(a);             \
            else k_expand_fast<G, false><<<grid, 256, 0, s>>>(a);                        \
        } else switch (variant) {                                                        \
            case 0: k_expand<G, false, false><<<grid, 256, 0, s>>>(a); break;            \
            case 1: k_expand<G, false, true><<<grid, 256, 0, s>>>(a); break;             \
            case 2: k_expand<G, true, false><<<grid, 256, 0, s>>>(a); break;             \
            default: k_expand<G, true, true><<<grid, 256, 0, s>>>(a); break;             \
        }
        if (size_cl == 0) { OTTO_EXPAND(8) } else if (size_cl == 1) { OTTO_EXPAND(16) } else { OTTO_EXPAND(32) }
#undef OTTO_EXPAND
        OTTO_HIP(hipGetLastError());
    }
    tend(c, OTTO_COVIS_T_EXPAND, s);

    c->rec_used += n_slots;
    c->run_used += n_ev;
    c->sessions += n_sess;
    c->index_valid = false;
    c->desc_totals_valid = false;
    return 0;
}

// pre (nullable): the totals of k_aid_totals for this configuration -- no host synchronisation then
static int build_items(otto_covis_ctx* c, int bin, int only_flagged, hipStream_t s, int allow_packed = 0, const uint64_t* pre = nullptr) {
    if (bin == 2) c->items_gen++;                        // the partition buckets of the previous list are stale
    const uint32_t n_aids = c->p.n_aids;
    ItemCount f{c->cnt64.as<uint64_t>(), c->boost.as<uint8_t>(), c->flag.as<uint32_t>(), bin, only_flagged, c->l_cap, allow_packed, -1};
    if (bin == 2) c->items_allow_packed = allow_packed;
    OTTO_TRY(device_scan(f, (int64_t)n_aids, c->item_start.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    uint64_t total = pre ? pre[8 + bin] : 0;
    if (!pre) {
        OTTO_HIP(hipMemcpyAsync(&total, c->item_start.as<uint64_t>() + n_aids, 8, hipMemcpyDeviceToHost, s));
        OTTO_HIP(hipStreamSynchronize(s));
    }
    OTTO_REQUIRE(total < (1ull << 32), "too many work items (%llu)", (unsigned long long)total);
    c->n_items[bin] = total;
    if (total) {
        OTTO_TRY(c->items[bin].ensure((size_t)total * 8, 0, s));
        k_fill_items<<<(n_aids + 255) / 256, 256, 0, s>>>(f, n_aids, c->item_start.as<uint64_t>(), c->items[bin].as<uint64_t>());
        OTTO_HIP(hipGetLastError());
    }
    if (bin == 2) {
        // processing order per table layout (pilots first); item_start holds the item index of every heavy aid
        for (int mode = 0; mode < 3; ++mode) {
            c->n_order[bin][mode] = 0;
            if (!total) continue;
            ItemCount fm = f;
            fm.mode = mode;
            OTTO_TRY(c->lrank.ensure((size_t)(n_aids + 1) * 8, 0, s));
            OTTO_TRY(c->lmode_start.ensure((size_t)(n_aids + 1) * 8, 0, s));
            OTTO_TRY(device_scan(ItemAny{fm}, (int64_t)n_aids, c->lrank.as<uint64_t>(), c->partial.as<uint64_t>(), s));
            OTTO_TRY(device_scan(fm, (int64_t)n_aids, c->lmode_start.as<uint64_t>(), c->partial.as<uint64_t>(), s));
            uint64_t n_pilots = pre ? pre[14 + mode] : 0, n_mode = pre ? pre[11 + mode] : 0;
            if (!pre) {
                OTTO_HIP(hipMemcpyAsync(&n_pilots, c->lrank.as<uint64_t>() + n_aids, 8, hipMemcpyDeviceToHost, s));
                OTTO_HIP(hipMemcpyAsync(&n_mode, c->lmode_start.as<uint64_t>() + n_aids, 8, hipMemcpyDeviceToHost, s));
                OTTO_HIP(hipStreamSynchronize(s));
            }
            c->n_order[bin][mode] = n_mode;
            if (!n_mode) continue;
            OTTO_TRY(c->lorder[bin][mode].ensure((size_t)n_mode * 4, 0, s));
            k_fill_order<<<(n_aids + 255) / 256, 256, 0, s>>>(fm, n_aids, c->item_start.as<uint64_t>(), c->lmode_start.as<uint64_t>(),
                                                              c->lrank.as<uint64_t>(), n_pilots, c->lorder[bin][mode].as<uint32_t>());
            OTTO_HIP(hipGetLastError());
        }
    }
    if (bin == 2) {
        // heavy aids: first L item of every aid + the work items of the partition pass
        OTTO_TRY(c->litem_start.ensure((size_t)(n_aids + 1) * 8, 0, s));
        OTTO_HIP(hipMemcpyAsync(c->litem_start.p, c->item_start.p, (size_t)(n_aids + 1) * 8, hipMemcpyDeviceToDevice, s));
        ChunkCount cf{c->cnt64.as<uint64_t>(), c->boost.as<uint8_t>(), c->flag.as<uint32_t>(), only_flagged, c->l_cap, allow_packed};
        OTTO_TRY(device_scan(cf, (int64_t)n_aids, c->item_start.as<uint64_t>(), c->partial.as<uint64_t>(), s));
        uint64_t nch = pre ? pre[17] : 0;
        if (!pre) {
            OTTO_HIP(hipMemcpyAsync(&nch, c->item_start.as<uint64_t>() + n_aids, 8, hipMemcpyDeviceToHost, s));
            OTTO_HIP(hipStreamSynchronize(s));
        }
        OTTO_REQUIRE(nch < (1ull << 32), "too many partition chunks");
        c->n_chunks = nch;
        if (nch) {
            OTTO_TRY(c->chunks.ensure((size_t)nch * 8, 0, s));
            k_fill_chunks<<<(n_aids + 255) / 256, 256, 0, s>>>(cf, n_aids, c->item_start.as<uint64_t>(), c->chunks.as<uint64_t>());
            OTTO_HIP(hipGetLastError());
        }
    }
    return 0;
}

static int build_index(otto_covis_ctx* c, hipStream_t s) {
    c->items_gen++;                                      // new runs / a new heavy item list: the partition buckets are stale
    const uint32_t n_aids = c->p.n_aids;
    tbegin(c, OTTO_COVIS_T_INDEX, s);
    OTTO_TRY(c->cnt64.ensure((size_t)n_aids * 8, 0, s));
    OTTO_TRY(c->run_start.ensure((size_t)(n_aids + 1) * 8, 0, s));
    OTTO_TRY(c->item_start.ensure((size_t)(n_aids + 1) * 8, 0, s));
    OTTO_TRY(c->run_rank.ensure((size_t)(c->run_used ? c->run_used : 1) * 4, 0, s));
    OTTO_TRY(c->boost.ensure((size_t)n_aids, 0, s));
    OTTO_TRY(c->flag.ensure((size_t)n_aids * 4, 0, s));
    OTTO_TRY(c->counters.ensure(512, 0, s));
    OTTO_TRY(c->partial.ensure(scan_partial_bytes((int64_t)n_aids), 0, s));
    OTTO_HIP(hipMemsetAsync(c->cnt64.p, 0, (size_t)n_aids * 8, s));
    OTTO_HIP(hipMemsetAsync(c->boost.p, 0, (size_t)n_aids, s));
    OTTO_HIP(hipMemsetAsync(c->flag.p, 0, (size_t)n_aids * 4, s));
    const int64_t n_slots = (int64_t)c->run_used;
    int aid_bits = 1;
    while ((1ull << aid_bits) < (uint64_t)n_aids) ++aid_bits;
    BktArgs ba;
    memset(&ba, 0, sizeof ba);
    // ~1000 buckets where the aid space allows it (measured at OTTO shape, 2^21 aids: buckets of 2^10 / 2^11 / 2^12 / 2^13 aids ->
    // index 2.95 / 3.05 / 3.1 / 3.3 ms: a bucket's runs (1 - 2 MB) should stay inside one XCD's L2 between the count walk, the
    // place walk and its scattered 8-byte stores; the pieces a 16 k-run chunk sends to a bucket are still 16 - 18 runs long);
    // buckets of at most 2^13 aids (the record's 13 bits, 96 KB of LDS in k_bkt_fused)
    ba.sh = aid_bits - 10 < 10 ? 10 : (aid_bits - 10 > BKT_MAX_SH ? BKT_MAX_SH : aid_bits - 10);
    if (c->bkt_sh >= 8 && c->bkt_sh <= BKT_MAX_SH) ba.sh = c->bkt_sh;                 // option "bkt_sh" (A/B)
    ba.nb = (uint32_t)(((uint64_t)n_aids + (1ull << ba.sh) - 1) >> ba.sh);
    const bool bucketed = c->bucket_index && n_slots > 0 && n_slots < (1ll << 32) && ba.nb <= (uint32_t)BKT_MAX_NB;
    if (bucketed) {
        ba.run_x = c->run_x.as<uint32_t>(); ba.run_desc = c->run_desc.as<uint64_t>();
        ba.n_slots = n_slots; ba.n_aids = n_aids;
        const int64_t n_chunks = (n_slots + BKT_CHUNK - 1) / BKT_CHUNK;
        const int sgrid = (int)(n_chunks < 256 * 2 ? n_chunks : 256 * 2);
        const int64_t n_cells = (int64_t)ba.nb * sgrid;
        OTTO_TRY(c->bcount.ensure((size_t)n_cells * 4, 0, s));
        OTTO_TRY(c->bstart.ensure((size_t)(n_cells + 1) * 8, 0, s));
        OTTO_TRY(c->partial.ensure(scan_partial_bytes(n_cells > (int64_t)n_aids ? n_cells : (int64_t)n_aids), 0, s));
        ba.bcnt_blk = c->bcount.as<uint32_t>();
        ba.bscan = c->bstart.as<uint64_t>();
        ba.bstride = (uint32_t)sgrid;
        kname(c, OTTO_COVIS_T_INDEX, "k_bkt_split<false/true> + k_bkt_fused + k_aid_totals + k_items_fill");
        k_bkt_split<false><<<sgrid, BKT_THREADS, 0, s>>>(ba);
        OTTO_HIP(hipGetLastError());
        OTTO_TRY(device_scan(BktCount{ba.bcnt_blk}, n_cells, c->bstart.as<uint64_t>(), c->partial.as<uint64_t>(), s));
        // non-empty runs <= run slots: the buffers are sized by the bound, so the split needs no host round trip
        OTTO_TRY(c->tmp_runs.ensure((size_t)n_slots * 8, 0, s));
        ba.tmp = c->tmp_runs.as<uint64_t>();
        k_bkt_split<true><<<sgrid, BKT_THREADS, 0, s>>>(ba);
        OTTO_HIP(hipGetLastError());
        ba.cnt64 = c->cnt64.as<uint64_t>();
        const int lgrid = (int)(ba.nb < 256u * 2u ? ba.nb : 256u * 2u);
        // the split left every bucket's runs contiguous: counting / scanning / placing is one launch
        OTTO_TRY(c->sorted_desc.ensure((size_t)n_slots * 8, 0, s));
        ba.sorted_desc = c->sorted_desc.as<uint64_t>();
        OTTO_HIP(hipFuncSetAttribute((const void*)k_bkt_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 12 << BKT_MAX_SH));
        k_bkt_fused<<<lgrid, BKT_THREADS, (size_t)12 << ba.sh, s>>>(ba, c->run_start.as<uint64_t>());
        OTTO_HIP(hipGetLastError());
    } else if (n_slots) {
        int grid = (int)((n_slots + 255) / 256 < 256 * 32 ? (n_slots + 255) / 256 : 256 * 32);
        kname(c, OTTO_COVIS_T_INDEX, "k_hist_runs + k_scatter_runs + k_scan_* + k_fill_*");
        k_hist_runs<<<grid, 256, 0, s>>>(c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>(), n_slots, c->cnt64.as<uint64_t>(),
                                         c->run_rank.as<uint32_t>(), n_aids);
        OTTO_HIP(hipGetLastError());
    }
    // every total the build needs on the host in one reduction + ONE synchronisation (boost / flag are zero here)
    uint64_t tot[N_TOTALS];
    const int item_blocks = (int)((n_aids + ITEM_BLOCK_AIDS - 1) / ITEM_BLOCK_AIDS);
    {
        unsigned long long* d_tot = c->counters.as<unsigned long long>() + 4;        // counters: 64 u64 words, [4, 4 + N_TOTALS) here
        OTTO_HIP(hipMemsetAsync(d_tot, 0, N_TOTALS * 8, s));
        OTTO_TRY(c->item_part.ensure((size_t)item_blocks * N_ITEM_SCANS * 8, 0, s));
        k_aid_totals<<<item_blocks, 256, 0, s>>>(c->cnt64.as<uint64_t>(), c->boost.as<uint8_t>(), c->flag.as<uint32_t>(), n_aids, c->l_cap,
                                                 c->packed_heavy, d_tot, c->item_part.as<unsigned long long>());
        OTTO_HIP(hipGetLastError());
        OTTO_HIP(hipMemcpyAsync(tot, d_tot, N_TOTALS * 8, hipMemcpyDeviceToHost, s));
    }
    if (!bucketed)
        OTTO_TRY(device_scan(RunCount{c->cnt64.as<uint64_t>()}, (int64_t)n_aids, c->run_start.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    OTTO_HIP(hipStreamSynchronize(s));
    c->n_pairs = tot[0];
    c->n_runs = tot[1];
    for (int bin = 0; bin < 3; ++bin) { c->bin_pairs[bin] = tot[2 + bin]; c->bin_runs[bin] = tot[5 + bin]; }
    if (!bucketed && n_slots) {
        OTTO_TRY(c->sorted_desc.ensure((size_t)(c->n_runs ? c->n_runs : 1) * 8, 0, s));
        int grid = (int)((n_slots + 255) / 256 < 256 * 32 ? (n_slots + 255) / 256 : 256 * 32);
        k_scatter_runs<<<grid, 256, 0, s>>>(c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>(), c->run_rank.as<uint32_t>(),
                                            n_slots, c->run_start.as<uint64_t>(), c->sorted_desc.as<uint64_t>(), n_aids);
        OTTO_HIP(hipGetLastError());
    }
    {   // all work lists in one launch (build_items is the per-bin form the overflow retries use)
        ItemFillArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.cnt64 = c->cnt64.as<uint64_t>(); fa.boost = c->boost.as<uint8_t>(); fa.n_aids = n_aids; fa.l_cap = c->l_cap;
        fa.allow_packed = c->packed_heavy; fa.block_part = c->item_part.as<unsigned long long>();
        for (int bin = 0; bin < 3; ++bin) {
            OTTO_REQUIRE(tot[8 + bin] < (1ull << 32), "too many work items (%llu)", (unsigned long long)tot[8 + bin]);
            c->n_items[bin] = tot[8 + bin];
            OTTO_TRY(c->items[bin].ensure((size_t)(tot[8 + bin] ? tot[8 + bin] : 1) * 8, 0, s));
            fa.items[bin] = c->items[bin].as<uint64_t>();
        }
        c->items_allow_packed = c->packed_heavy;
        for (int mode = 0; mode < 3; ++mode) {
            c->n_order[2][mode] = tot[11 + mode];
            OTTO_TRY(c->lorder[2][mode].ensure((size_t)(tot[11 + mode] ? tot[11 + mode] : 1) * 4, 0, s));
            fa.order[mode] = c->lorder[2][mode].as<uint32_t>();
            fa.n_pilots[mode] = tot[14 + mode];
        }
        OTTO_REQUIRE(tot[17] < (1ull << 32), "too many partition chunks");
        c->n_chunks = tot[17];
        OTTO_TRY(c->chunks.ensure((size_t)(tot[17] ? tot[17] : 1) * 8, 0, s));
        fa.chunks = c->chunks.as<uint64_t>();
        OTTO_TRY(c->litem_start.ensure((size_t)(n_aids + 1) * 8, 0, s));
        fa.litem_start = c->litem_start.as<uint64_t>();
        k_item_part_scan<<<1, 64 * N_ITEM_SCANS, 0, s>>>(c->item_part.as<unsigned long long>(), (uint32_t)item_blocks);
        k_items_fill<<<item_blocks, 256, 0, s>>>(fa);
        OTTO_HIP(hipGetLastError());
    }
    tend(c, OTTO_COVIS_T_INDEX, s);
    c->retries = 0;
    c->index_valid = true;
    return 0;
}

static void reduce_name(otto_covis_ctx* c, int slot, int log2t, int threads, int group, bool packed, int minw, int gu) {
    kname(c, slot, "k_reduce<%d, %d, %d, %s, %d, %d, 1, false>", log2t, threads, group, packed ? "true" : "false", minw, gu);
}
// the diagnostics instantiation (DBG) exists for the TYPE group only (the bench path); debug_skip is ignored elsewhere
#define OTTO_LAUNCH_REDUCE(slot, grid, threads, args, ...)                                             \
    do {                                                                                               \
        reduce_name(c, slot, __VA_ARGS__);                                                             \
        if (GROUP == OTTO_COVIS_GROUP_TYPE && (args).debug_skip)                                       \
            k_reduce<__VA_ARGS__, 1, GROUP == OTTO_COVIS_GROUP_TYPE><<<grid, threads, 0, s>>>(args);   \
        else                                                                                           \
            k_reduce<__VA_ARGS__, 1, false><<<grid, threads, 0, s>>>(args);                            \
    } while (0)

// bucket the records of the partitioned heavy aids by hash partition (a: the L item list of this pass). Leaves pstart / pcursor /
// prec (/ ptw) in the context. Synchronises `s` once (bucket total -> allocation).
static int run_partition(otto_covis_ctx* c, const ReduceArgs& a, bool time, hipStream_t s) {
            // bucket the records of heavy aids by hash partition once (count -> scan -> scatter)
    tbegin(c, OTTO_COVIS_T_PARTITION, s);
    OTTO_TRY(c->pcount.ensure((size_t)a.n_items * 4, 0, s));
    OTTO_TRY(c->pcursor.ensure((size_t)a.n_items * 4, 0, s));
    OTTO_TRY(c->pstart.ensure((size_t)(a.n_items + 1) * 8, 0, s));
    OTTO_TRY(c->partial.ensure(scan_partial_bytes((int64_t)a.n_items), 0, s));
    OTTO_HIP(hipMemsetAsync(c->pcursor.p, 0, (size_t)a.n_items * 4, s));
    PartArgs pa{c->chunks.as<uint64_t>(), (uint32_t)c->n_chunks, c->cnt64.as<uint64_t>(), c->boost.as<uint8_t>(),
                c->run_start.as<uint64_t>(), c->sorted_desc.as<uint64_t>(), c->rec.as<uint32_t>(), c->tw.as<uint32_t>(),
                c->litem_start.as<uint64_t>(), c->pcount.as<uint32_t>(), c->pcursor.as<uint32_t>(),
                c->pstart.as<uint64_t>(), nullptr, nullptr, c->l_cap, c->p.window, a.allow_packed, c->flag.as<uint32_t>(),
                c->counters.as<uint32_t>()};
    const uint32_t pres = 256u * (uint32_t)c->p_wgs;
    const uint32_t pgrid = (uint32_t)(c->n_chunks < pres ? c->n_chunks : pres);
    const uint32_t cgrid = (uint32_t)(c->n_chunks < 256u * 8u ? c->n_chunks : 256u * 8u);
    // First attempt: buckets sized from the record counts the index already holds (2 x mean + margin), no count pass.
    // Retry rounds (aids whose bucket or LDS table overflowed): counted buckets, exact.
    const bool sized = c->part_sized && !c->exact_round;
    if (sized) {
        kname(c, OTTO_COVIS_T_PARTITION, "k_partition<true>");
        OTTO_TRY(device_scan(ItemCap{a.items, c->cnt64.as<uint64_t>()}, (int64_t)a.n_items, c->pstart.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    } else {
        kname(c, OTTO_COVIS_T_PARTITION, "k_partition<false> + k_partition<true>");
        OTTO_HIP(hipMemsetAsync(c->pcount.p, 0, (size_t)a.n_items * 4, s));
        k_partition<false, false><<<cgrid, 256, 0, s>>>(pa);
        OTTO_HIP(hipGetLastError());
        OTTO_TRY(device_scan(PCount{c->pcount.as<uint32_t>()}, (int64_t)a.n_items, c->pstart.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    }
    uint64_t bucket_total = 0;
    OTTO_HIP(hipMemcpyAsync(&bucket_total, c->pstart.as<uint64_t>() + a.n_items, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    OTTO_TRY(c->prec.ensure((size_t)(bucket_total ? bucket_total : 1) * 4, 0, s));
    if (time) OTTO_TRY(c->ptw.ensure((size_t)(bucket_total ? bucket_total : 1) * 4, 0, s));
    pa.prec = c->prec.as<uint32_t>();
    pa.ptw = time ? c->ptw.as<uint32_t>() : nullptr;
    pa.work = c->counters.as<uint32_t>() + 8;             // (slots 0 .. 3: overflow count + the reduce bins' work counters)
    OTTO_HIP(hipMemsetAsync(pa.work, 0, 4, s));
    if (time) k_partition<true, true><<<pgrid, 256, 0, s>>>(pa);
    else k_partition<true, false><<<pgrid, 256, 0, s>>>(pa);
    OTTO_HIP(hipGetLastError());
    tend(c, OTTO_COVIS_T_PARTITION, s);
    return 0;
}

template <int GROUP>
static int launch_reduce(otto_covis_ctx* c, ReduceArgs a, int bin, hipStream_t s) {
    a.items = c->items[bin].as<uint64_t>();
    a.n_items = (uint32_t)c->n_items[bin];
    a.n_work = a.n_items;
    if (a.n_items == 0) return 0;
    a.order = nullptr;
    a.allow_packed = c->items_allow_packed > 0 ? c->items_allow_packed : 0;
    uint32_t* wc = c->counters.as<uint32_t>() + 1 + bin;
    OTTO_HIP(hipMemsetAsync(wc, 0, 4, s));
    a.work_counter = wc;
#ifdef OTTO_PHASE_PROF
    // diagnostic build only: per-launch phase shares (shader-clock ticks of thread 0 of every workgroup) + wall time
    static unsigned long long* d_prof = nullptr;
    static hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (!d_prof) { OTTO_HIP(hipMalloc(&d_prof, 128)); OTTO_HIP(hipEventCreate(&pe0)); OTTO_HIP(hipEventCreate(&pe1)); }
    a.prof = d_prof;
    auto prof_begin = [&]() { (void)hipMemsetAsync(d_prof, 0, 128, s); (void)hipEventRecord(pe0, s); };
    auto prof_end = [&](const char* tag, uint32_t n_work) {
        unsigned long long h[16];
        (void)hipEventRecord(pe1, s); (void)hipStreamSynchronize(s); (void)hipMemcpy(h, d_prof, 128, hipMemcpyDeviceToHost);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, pe0, pe1);
        unsigned long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[i];
        fprintf(stderr, "[phase-prof] %s items %u  %.3f ms:", tag, n_work, ms);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " p%d %.1f%%", i, tot ? 100.0 * h[i] / tot : 0.0);
        fprintf(stderr, "  (ticks %llu) guess: tried %llu ok %llu toofew %llu overflow %llu | record wait (not in p2): %.1f%%, next-item fetch (in p1): %.1f%% of the ticks | single-wave selection of few heavy keys: %llu items\n", tot, h[8], h[9], h[10], h[11], tot ? 100.0 * (double)h[12] / tot : 0.0, tot ? 100.0 * (double)h[13] / tot : 0.0, h[14]);
    };
#else
    auto prof_begin = [&]() {};
    auto prof_end = [&](const char*, uint32_t) {};
#endif
    if (bin == 0) {
        const uint32_t s_res = 256u * (uint32_t)c->s_wgs;
        uint32_t grid = a.n_items < s_res ? a.n_items : s_res;
        tbegin(c, OTTO_COVIS_T_REDUCE_S, s);
        prof_begin();
        OTTO_LAUNCH_REDUCE(OTTO_COVIS_T_REDUCE_S, grid, S_THREADS, a, S_LOG2T, S_THREADS, GROUP, true, 5, 4);
        prof_end("S", a.n_work);
        tend(c, OTTO_COVIS_T_REDUCE_S, s);
    } else if (bin == 1) {
        uint32_t grid = a.n_items < 256u * 4u ? a.n_items : 256u * 4u;
        tbegin(c, OTTO_COVIS_T_REDUCE_M, s);
        prof_begin();
        OTTO_LAUNCH_REDUCE(OTTO_COVIS_T_REDUCE_M, grid, M_THREADS, a, M_LOG2T, M_THREADS, GROUP, true, 4, 4);
        prof_end("M", a.n_work);
        tend(c, OTTO_COVIS_T_REDUCE_M, s);
    } else {
        a.pstart = nullptr;
        if (c->partition && c->n_chunks) {
            if (c->part_early) {
                // launched on the side stream before the S / M bins (otto_covis_finalize): the heavy bin waits for it here
                OTTO_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
                c->part_early = false;
                c->part_gen = 0;
            } else if (c->part_gen == c->items_gen && !c->exact_round && (GROUP != OTTO_COVIS_GROUP_TIME || c->part_has_tw)) {
                // the buckets of an earlier pass / group are still valid: every group partitions the same records of the same
                // heavy item list the same way (a 7-kind build ran this pass three times)
            } else {
                const bool with_tw = c->p.want_time;          // once for every group: the time channel travels along if any group needs it
                c->part_gen = 0;
                OTTO_TRY(run_partition(c, a, with_tw, s));
                if (!c->exact_round) { c->part_gen = c->items_gen; c->part_has_tw = with_tw; }
            }
            a.pstart = c->pstart.as<uint64_t>();
            a.pcursor = c->pcursor.as<uint32_t>();
            a.prec = c->prec.as<uint32_t>();
            a.ptw = c->ptw.as<uint32_t>();
        }
        tbegin(c, OTTO_COVIS_T_REDUCE_L, s);
        // the two table layouts of heavy aids: each kernel walks its own pilot-first order over the shared item list
        for (int mode = 2; mode >= 0; --mode) {
            if (!c->n_order[2][mode]) continue;
            ReduceArgs am = a;
            am.order = c->lorder[2][mode].as<uint32_t>();
            am.n_work = (uint32_t)c->n_order[2][mode];
            OTTO_HIP(hipMemsetAsync(wc, 0, 4, s));
            prof_begin();
            if (mode == 2) {
                const uint32_t grid = am.n_work < 256u * 2u ? am.n_work : 256u * 2u;
                OTTO_LAUNCH_REDUCE(OTTO_COVIS_T_REDUCE_L, grid, 512, am, L_LOG2T, 512, GROUP, true, 4, 4);
            } else if (mode == 1) {
                const uint32_t grid = am.n_work < 256u ? am.n_work : 256u;
                OTTO_LAUNCH_REDUCE(OTTO_COVIS_T_REDUCE_L, grid, L_THREADS, am, LP_LOG2T, L_THREADS, GROUP, true, 4, 2);
            } else {
                const uint32_t grid = am.n_work < 256u ? am.n_work : 256u;
                OTTO_LAUNCH_REDUCE(OTTO_COVIS_T_REDUCE_L, grid, L_THREADS, am, L_LOG2T, L_THREADS, GROUP, false, 4, 2);
            }
            prof_end(mode == 2 ? "L packed 2^13 x512" : (mode == 1 ? "L packed 2^14 x1024" : "L wide 2^13 x1024"), am.n_work);
            OTTO_HIP(hipGetLastError());
        }
        tend(c, OTTO_COVIS_T_REDUCE_L, s);
        OTTO_HIP(hipGetLastError());
        tbegin(c, OTTO_COVIS_T_MERGE, s);
        kname(c, OTTO_COVIS_T_MERGE, "k_merge<%d>", GROUP);
        k_merge<GROUP><<<a.n_items, 64 * PK, 0, s>>>(a);
        tend(c, OTTO_COVIS_T_MERGE, s);
    }
    OTTO_HIP(hipGetLastError());
    return 0;
}

static int launch_reduce_group(otto_covis_ctx* c, const ReduceArgs& a, int bin, hipStream_t s) {
    switch (a.group) {
        case OTTO_COVIS_GROUP_TYPE: return launch_reduce<OTTO_COVIS_GROUP_TYPE>(c, a, bin, s);
        case OTTO_COVIS_GROUP_FILTER: return launch_reduce<OTTO_COVIS_GROUP_FILTER>(c, a, bin, s);
        default: return launch_reduce<OTTO_COVIS_GROUP_TIME>(c, a, bin, s);
    }
}

extern "C" int otto_covis_finalize(otto_covis_ctx* c, int group, int k, uint32_t* d_out_y, uint64_t* d_out_w,
                                   int32_t* d_out_n, void* stream) {
    OTTO_REQUIRE(c && d_out_y && d_out_w && d_out_n, "otto_covis_finalize: null argument");
    OTTO_REQUIRE(k >= 1 && k <= MAX_K, "k must be in [1, %d], got %d", MAX_K, k);
    const otto_covis_params& p = c->p;
    int n_kinds = 0;
    switch (group) {
        case OTTO_COVIS_GROUP_TYPE: n_kinds = p.n_type_weights; break;
        case OTTO_COVIS_GROUP_FILTER: n_kinds = p.n_filters; break;
        case OTTO_COVIS_GROUP_TIME:
            OTTO_REQUIRE(p.want_time, "GROUP_TIME needs params.want_time");
            n_kinds = 1;
            break;
        default: OTTO_REQUIRE(false, "unknown group %d", group);
    }
    OTTO_REQUIRE(n_kinds > 0, "group %d has no kinds configured", group);
    hipStream_t s = (hipStream_t)stream;
    if (!c->index_valid) OTTO_TRY(build_index(c, s));
    // every group uses the same layout rule (the time-weighted kind has its packed form too: one sum per key, fewer than 4096 runs
    // keep it under 2^30), so the heavy item list is built once for all groups
    const int want_packed = c->packed_heavy ? c->packed_heavy : 0;
    if (c->items_allow_packed != want_packed) OTTO_TRY(build_items(c, 2, 0, s, want_packed));
    const uint32_t n_aids = p.n_aids;
    OTTO_HIP(hipMemsetAsync(d_out_n, 0, (size_t)n_kinds * n_aids * 4, s));

    // FILTER passes carry 3 channels each; TYPE passes select up to 4 weight vectors from 3 counters.
    const int per_pass = PK;
    for (int kb = 0; kb < n_kinds; kb += per_pass) {
        ReduceArgs a;
        memset(&a, 0, sizeof a);
        a.cnt64 = c->cnt64.as<uint64_t>();
        a.run_start = c->run_start.as<uint64_t>();
        a.sorted_desc = c->sorted_desc.as<uint64_t>();
        a.rec = c->rec.as<uint32_t>();
        a.tw = c->tw.as<uint32_t>();
        a.group = group;
        a.nk = n_kinds - kb < per_pass ? n_kinds - kb : per_pass;
        a.k = k;
        a.chan_shift = group == OTTO_COVIS_GROUP_FILTER ? kb : 0;
        for (int j = 0; j < a.nk; ++j)
            for (int t = 0; t < 3; ++t)
                a.coef[j][t] = group == OTTO_COVIS_GROUP_TYPE ? (uint32_t)p.type_weight[kb + j][t] : (uint32_t)(t == j);
        a.n_aids = n_aids;
        a.kind_base = kb;
        a.out_y = d_out_y; a.out_w = d_out_w; a.out_n = d_out_n;
        a.flag = c->flag.as<uint32_t>();
        a.boost = c->boost.as<uint8_t>();
        a.ovf_count = c->counters.as<uint32_t>();
        a.l_cap = c->l_cap;
        a.debug_skip = c->debug_skip;
        // heavy-first top-k walks (k_reduce): every kind of the pass must rank a key made of ONE click record strictly below
        // a key of two records and below a key of one cart / order record
        a.hot_ok = 0;
        if (group == OTTO_COVIS_GROUP_TYPE && c->hot) {
            a.hot_ok = c->hot;
            for (int j = 0; j < a.nk; ++j) {
                const uint32_t c0 = a.coef[j][0], c1 = a.coef[j][1], c2 = a.coef[j][2];
                const uint32_t mn = c0 < c1 ? (c0 < c2 ? c0 : c2) : (c1 < c2 ? c1 : c2);
                if (!(2 * mn > c0 && c1 > c0 && c2 > c0)) a.hot_ok = 0;
            }
        }

        if (c->guess && c->partition && c->n_items[2]) {
            OTTO_TRY(c->tau_w.ensure((size_t)PK * n_aids * 8, 0, s));
            OTTO_HIP(hipMemsetAsync(c->tau_w.p, 0, (size_t)PK * n_aids * 8, s));
            a.tau_w = c->tau_w.as<uint64_t>();
            if (group == OTTO_COVIS_GROUP_TIME) {
                OTTO_TRY(c->tau_y.ensure((size_t)PK * n_aids * 4, 0, s));
                OTTO_HIP(hipMemsetAsync(c->tau_y.p, 0, (size_t)PK * n_aids * 4, s));
                a.tau_y = c->tau_y.as<uint32_t>();
            }
        }
        bool first = true;
        for (;;) {
            OTTO_HIP(hipMemsetAsync(c->counters.p, 0, 4, s));
            if (c->n_items[2]) {
                const size_t need = (size_t)c->n_items[2] * a.nk * k;
                OTTO_TRY(c->part_y.ensure(need * 4, 0, s));
                OTTO_TRY(c->part_w.ensure(need * 8, 0, s));
                a.part_y = c->part_y.as<uint32_t>();
                a.part_w = c->part_w.as<uint64_t>();
            }
            if (first) {
                // the partition pass of the heavy aids (a gather that waits on memory) beside the S and M bins (LDS atomics):
                // forked onto the side stream behind everything queued so far, joined where the heavy bin starts
                if (c->overlap_partition && c->side && c->partition && c->n_chunks && c->n_items[2]) {
                    ReduceArgs ap = a;
                    ap.items = c->items[2].as<uint64_t>();
                    ap.n_items = (uint32_t)c->n_items[2];
                    ap.allow_packed = c->items_allow_packed > 0 ? c->items_allow_packed : 0;
                    OTTO_HIP(hipEventRecord(c->ev_fork, s));
                    OTTO_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                    OTTO_TRY(run_partition(c, ap, group == OTTO_COVIS_GROUP_TIME, c->side));
                    OTTO_HIP(hipEventRecord(c->ev_join, c->side));
                    c->part_early = true;
                }
                OTTO_TRY(launch_reduce_group(c, a, 0, s));
                OTTO_TRY(launch_reduce_group(c, a, 1, s));
            }
            c->exact_round = !first;
            OTTO_TRY(launch_reduce_group(c, a, 2, s));
            uint32_t ovf = 0;
            OTTO_HIP(hipMemcpyAsync(&ovf, c->counters.p, 4, hipMemcpyDeviceToHost, s));
            OTTO_HIP(hipStreamSynchronize(s));
            if (ovf == 0) break;
            // some heavy aids overflowed their LDS table: re-partition only those (boost[x] was raised)
            c->retries++;
            OTTO_REQUIRE(c->retries < 64, "overflow re-partitioning did not converge");
            OTTO_TRY(build_items(c, 2, 1, s, c->items_allow_packed));
            OTTO_HIP(hipMemsetAsync(c->flag.p, 0, (size_t)n_aids * 4, s));
            first = false;
        }
        if (!first) OTTO_TRY(build_items(c, 2, 0, s, c->items_allow_packed));   // restore the full L list (boost kept) for later passes
    }
    return 0;
}

extern "C" int otto_covis_set_option(otto_covis_ctx* c, const char* name, int64_t value) {
    OTTO_REQUIRE(c && name, "null argument");
    c->part_gen = 0;                                     // (any switch may change how the heavy aids are partitioned)
    if (strcmp(name, "l_cap") == 0) {
        // records per hash partition of a heavy aid (L bin). Larger = fewer passes over the aid's
        // records but more LDS-table overflows (each costs a re-partition round); any value is exact.
        OTTO_REQUIRE(value >= 64 && value <= (1ll << 30), "l_cap out of range");
        c->l_cap = (uint32_t)value;
        c->index_valid = false;
    c->desc_totals_valid = false;
        return 0;
    }
    if (strcmp(name, "debug_skip") == 0) { c->debug_skip = (int)value; return 0; }   // timing diagnostics, results invalid
    if (strcmp(name, "packed_heavy") == 0) { c->packed_heavy = value == 2 ? 2 : (value != 0); c->index_valid = false; return 0; }
    if (strcmp(name, "bucket_index") == 0) { c->bucket_index = value != 0; return 0; }
    if (strcmp(name, "guess") == 0) { c->guess = value != 0; return 0; }           // threshold guessing on/off (A/B)
    if (strcmp(name, "fused") == 0) { c->fused = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return 0; }   // 2 component lists, 1 fused register rows, 0 class-sorted kernels (A/B)
    if (strcmp(name, "p_wgs") == 0) { c->p_wgs = value < 1 ? 1 : (value > 128 ? 128 : (int)value); return 0; }
    if (strcmp(name, "s_wgs") == 0) { c->s_wgs = value < 1 ? 1 : (value > 32 ? 32 : (int)value); return 0; }
    if (strcmp(name, "bkt_sh") == 0) { c->bkt_sh = (int)value; c->index_valid = false; return 0; }
    if (strcmp(name, "overlap_partition") == 0) { c->overlap_partition = value != 0; return 0; }
    if (strcmp(name, "hot") == 0) { c->hot = value < 0 ? 0 : (value > 2 ? 2 : (int)value); return 0; }
    if (strcmp(name, "fast_path") == 0) { c->fast_path = value != 0; return 0; }   // gap-free window kernel on/off (A/B)
    if (strcmp(name, "part_sized") == 0) { c->part_sized = value != 0; return 0; }   // A/B: counted buckets only
    if (strcmp(name, "partition") == 0) {
        // 1 (default): bucket heavy aids' records by hash partition once; 0: every partition re-reads
        // all of its aid's records and filters (round-1 baseline, kept for A/B measurements)
        c->partition = value != 0;
        return 0;
    }
    OTTO_REQUIRE(false, "unknown option '%s'", name);
}

// ---- PMC calibration (MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are only calibrated for 16-byte-per-lane
// streams; these two kernels move a KNOWN byte count with the 4-byte-per-lane pattern of the covisitation kernels) ----
__global__ void otto_calib_read_u32(const uint32_t* p, int64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x9E3779B9u) *sink = acc;
}
__global__ void otto_calib_write_u32(uint32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
extern "C" int otto_debug_calibrate(uint32_t* d_buf, int64_t n_u32, int32_t write, void* stream) {
    OTTO_REQUIRE(d_buf && n_u32 > 0, "bad calibration buffer");
    if (write) otto_calib_write_u32<<<256 * 8, 256, 0, (hipStream_t)stream>>>(d_buf, n_u32);
    else otto_calib_read_u32<<<256 * 8, 256, 0, (hipStream_t)stream>>>(d_buf, n_u32, d_buf);
    OTTO_HIP(hipGetLastError());
    return 0;
}

extern "C" int otto_covis_stats(otto_covis_ctx* c, int64_t* out) {
    OTTO_REQUIRE(c && out, "null argument");
    out[OTTO_COVIS_STAT_SESSIONS] = c->sessions;
    out[OTTO_COVIS_STAT_TAIL_EVENTS] = (int64_t)c->run_used;
    out[OTTO_COVIS_STAT_PAIR_SLOTS] = (int64_t)c->rec_used;
    out[OTTO_COVIS_STAT_PAIRS] = (int64_t)c->n_pairs;
    out[OTTO_COVIS_STAT_RUNS] = (int64_t)c->n_runs;
    out[OTTO_COVIS_STAT_ITEMS_S] = (int64_t)c->n_items[0];
    out[OTTO_COVIS_STAT_ITEMS_M] = (int64_t)c->n_items[1];
    out[OTTO_COVIS_STAT_ITEMS_L] = (int64_t)c->n_items[2];
    out[OTTO_COVIS_STAT_RETRIES] = c->retries;
    out[OTTO_COVIS_STAT_PAIRS_S] = (int64_t)c->bin_pairs[0];
    out[OTTO_COVIS_STAT_PAIRS_M] = (int64_t)c->bin_pairs[1];
    out[OTTO_COVIS_STAT_PAIRS_L] = (int64_t)c->bin_pairs[2];
    out[OTTO_COVIS_STAT_RUNS_S] = (int64_t)c->bin_runs[0];
    out[OTTO_COVIS_STAT_RUNS_M] = (int64_t)c->bin_runs[1];
    out[OTTO_COVIS_STAT_RUNS_L] = (int64_t)c->bin_runs[2];
    // what the pair-expand kernel wrote: one list word per run that reads a shared list, the records of the private rows
    if (!c->desc_totals_valid) {
        c->desc_totals[0] = c->desc_totals[1] = 0;
        if (c->run_used) {
            OTTO_HIP(hipDeviceSynchronize());
            OTTO_TRY(c->exp_totals.ensure(2 * MAX_OWNERS * 8, 0, nullptr));
            OTTO_HIP(hipMemset(c->exp_totals.p, 0, 16));
            const int64_t nb = ((int64_t)c->run_used + 255) / 256;
            k_desc_totals<<<(unsigned)(nb < 4096 ? nb : 4096), 256>>>(c->run_desc.as<uint64_t>(), (int64_t)c->run_used, c->exp_totals.as<unsigned long long>());
            OTTO_HIP(hipGetLastError());
            OTTO_HIP(hipMemcpy(c->desc_totals, c->exp_totals.p, 16, hipMemcpyDeviceToHost));
            c->exp_planned = 0;                       // the export cursors shared that buffer
        }
        c->desc_totals_valid = true;
    }
    out[OTTO_COVIS_STAT_SHARED_RUNS] = (int64_t)c->desc_totals[0];
    out[OTTO_COVIS_STAT_ROW_RECORDS] = (int64_t)c->desc_totals[1];
    return 0;
}

extern "C" int otto_covis_timings(otto_covis_ctx* c, float* out_ms) {
    OTTO_REQUIRE(c && out_ms, "null argument");
    for (int i = 0; i < OTTO_COVIS_T_COUNT; ++i) {
        out_ms[i] = 0.f;
        if (c->ev_ok && c->ev_set[i]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]) == hipSuccess) out_ms[i] = ms;
        }
    }
    return OTTO_COVIS_T_COUNT;
}

extern "C" int otto_covis_kernel_names(otto_covis_ctx* c, int32_t slot, char* buf, int32_t n) {
    OTTO_REQUIRE(c && buf && n > 0, "null argument");
    OTTO_REQUIRE(slot >= 0 && slot < OTTO_COVIS_T_COUNT, "timing slot %d out of range", slot);
    snprintf(buf, (size_t)n, "%s", c->knames[slot].c_str());
    return 0;
}

extern "C" int otto_covis_copy_records(otto_covis_ctx* c, uint32_t* h_rec, uint32_t* h_tw, uint32_t* h_run_x,
                                       uint64_t* h_run_desc) {
    OTTO_REQUIRE(c, "null ctx");
    OTTO_HIP(hipDeviceSynchronize());
    if (h_rec && c->rec_used) OTTO_HIP(hipMemcpy(h_rec, c->rec.p, (size_t)c->rec_used * 4, hipMemcpyDeviceToHost));
    if (h_tw && c->rec_used) {
        OTTO_REQUIRE(c->p.want_time, "no time channel (want_time == 0)");
        OTTO_HIP(hipMemcpy(h_tw, c->tw.p, (size_t)c->rec_used * 4, hipMemcpyDeviceToHost));
    }
    if (h_run_x && c->run_used) OTTO_HIP(hipMemcpy(h_run_x, c->run_x.p, (size_t)c->run_used * 4, hipMemcpyDeviceToHost));
    if (h_run_desc && c->run_used) OTTO_HIP(hipMemcpy(h_run_desc, c->run_desc.p, (size_t)c->run_used * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int otto_covis_export_count(otto_covis_ctx* c, uint32_t x_lo, uint32_t x_hi, int64_t* n_runs, int64_t* n_recs,
                                       void* stream) {
    OTTO_REQUIRE(c && n_runs && n_recs, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_slots = (int64_t)c->run_used;
    OTTO_TRY(c->exp_run_pos.ensure((size_t)(n_slots + 1) * 8, 0, s));
    OTTO_TRY(c->exp_rec_pos.ensure((size_t)(n_slots + 1) * 8, 0, s));
    OTTO_TRY(c->partial.ensure(scan_partial_bytes(n_slots), 0, s));
    OTTO_TRY(device_scan(ExportRuns{c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>(), x_lo, x_hi}, n_slots,
                         c->exp_run_pos.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    OTTO_TRY(device_scan(ExportRecs{c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>(), x_lo, x_hi}, n_slots,
                         c->exp_rec_pos.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    uint64_t t[2];
    OTTO_HIP(hipMemcpyAsync(&t[0], c->exp_run_pos.as<uint64_t>() + n_slots, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipMemcpyAsync(&t[1], c->exp_rec_pos.as<uint64_t>() + n_slots, 8, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    *n_runs = (int64_t)t[0];
    *n_recs = (int64_t)t[1];
    return 0;
}

extern "C" int otto_covis_export_runs(otto_covis_ctx* c, uint32_t x_lo, uint32_t x_hi, uint32_t* d_hdr, uint32_t* d_rec,
                                      uint32_t* d_tw, void* stream) {
    OTTO_REQUIRE(c, "null ctx");
    hipStream_t s = (hipStream_t)stream;
    int64_t nr = 0, nc = 0;
    OTTO_TRY(otto_covis_export_count(c, x_lo, x_hi, &nr, &nc, stream));
    if (nr == 0) return 0;
    OTTO_REQUIRE(d_hdr && d_rec, "null export buffers");
    OTTO_REQUIRE(!d_tw || c->p.want_time, "no time channel to export");
    const int64_t n_slots = (int64_t)c->run_used;
    int grid = (int)((n_slots + 255) / 256 < 256 * 32 ? (n_slots + 255) / 256 : 256 * 32);
    k_export<<<grid, 256, 0, s>>>(c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>(), n_slots, x_lo, x_hi,
                                  c->exp_run_pos.as<uint64_t>(), c->exp_rec_pos.as<uint64_t>(), c->rec.as<uint32_t>(),
                                  c->tw.as<uint32_t>(), d_hdr, d_rec, d_tw);
    OTTO_HIP(hipGetLastError());
    return 0;
}

static int owner_args(otto_covis_ctx* c, int n_owners, const uint32_t* h_bounds, OwnerArgs& a, int64_t slot_lo = 0, int64_t slot_hi = -1) {
    OTTO_REQUIRE(c && h_bounds, "null argument");
    OTTO_REQUIRE(n_owners >= 1 && n_owners <= MAX_OWNERS, "n_owners must be in [1, %d]", MAX_OWNERS);
    for (int o = 0; o < n_owners; ++o) OTTO_REQUIRE(h_bounds[o] <= h_bounds[o + 1], "owner bounds must be non-decreasing");
    if (slot_hi < 0) slot_hi = (int64_t)c->run_used;
    OTTO_REQUIRE(slot_lo >= 0 && slot_lo <= slot_hi && slot_hi <= (int64_t)c->run_used, "run slot range [%lld, %lld) outside [0, %llu]",
                 (long long)slot_lo, (long long)slot_hi, (unsigned long long)c->run_used);
    memset(&a, 0, sizeof a);
    a.run_x = c->run_x.as<uint32_t>();
    a.run_desc = c->run_desc.as<uint64_t>();
    a.slot0 = slot_lo;
    a.n_slots = slot_hi - slot_lo;
    a.n_owners = n_owners;
    for (int o = 0; o <= n_owners; ++o) a.bounds[o] = h_bounds[o];
    a.bounds[0] = 0;
    a.bounds[n_owners] = 0xFFFFFFFFu;
    return 0;
}

extern "C" int otto_covis_export_plan(otto_covis_ctx* c, int n_owners, const uint32_t* h_bounds, int64_t* h_n_runs,
                                      int64_t* h_n_recs, void* stream) {
    OTTO_REQUIRE(h_n_runs && h_n_recs, "null argument");
    hipStream_t s = (hipStream_t)stream;
    OwnerArgs a;
    OTTO_TRY(owner_args(c, n_owners, h_bounds, a));
    OTTO_REQUIRE(c->run_used < (1ull << (64 - EXP_REC_BITS)) && c->rec_used < (1ull << EXP_REC_BITS), "too many runs for the packed export cursor");
    OTTO_TRY(c->exp_totals.ensure(2 * MAX_OWNERS * 8, 0, s));
    OTTO_HIP(hipMemsetAsync(c->exp_totals.p, 0, 2 * MAX_OWNERS * 8, s));
    a.totals = c->exp_totals.as<unsigned long long>();
    if (a.n_slots) {
        const int64_t nb = (a.n_slots + 255) / 256;
        k_export_plan<<<(unsigned)(nb < 256 * 8 ? nb : 256 * 8), 256, 0, s>>>(a);
        OTTO_HIP(hipGetLastError());
    }
    unsigned long long t[MAX_OWNERS];
    OTTO_HIP(hipMemcpyAsync(t, c->exp_totals.p, sizeof t, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    for (int o = 0; o < n_owners; ++o) {
        c->exp_n_runs[o] = t[o] >> EXP_REC_BITS;
        c->exp_n_recs[o] = t[o] & ((1ull << EXP_REC_BITS) - 1ull);
        h_n_runs[o] = (int64_t)c->exp_n_runs[o];
        h_n_recs[o] = (int64_t)c->exp_n_recs[o];
    }
    c->exp_planned = n_owners;
    return 0;
}

extern "C" int otto_covis_export_fill(otto_covis_ctx* c, int n_owners, const uint32_t* h_bounds, uint32_t* d_hdr,
                                      uint32_t* d_rec, uint32_t* d_tw, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    OwnerArgs a;
    OTTO_TRY(owner_args(c, n_owners, h_bounds, a));
    OTTO_REQUIRE(c->exp_planned == n_owners, "call otto_covis_export_plan with the same owners first");
    OTTO_REQUIRE(!d_tw || c->p.want_time, "no time channel to export");
    uint64_t rb = 0, cb = 0;
    for (int o = 0; o < n_owners; ++o) {
        a.run_base[o] = rb;
        a.rec_base[o] = cb;
        rb += c->exp_n_runs[o];
        cb += c->exp_n_recs[o];
    }
    if (rb == 0) return 0;
    OTTO_REQUIRE(d_hdr && d_rec, "null export buffers");
    OTTO_HIP(hipMemsetAsync(c->exp_totals.p, 0, 2 * MAX_OWNERS * 8, s));
    a.totals = c->exp_totals.as<unsigned long long>();
    a.rec = c->rec.as<uint32_t>();
    a.tw = c->tw.as<uint32_t>();
    a.o_hdr = d_hdr; a.o_rec = d_rec; a.o_tw = d_tw;
    const int64_t nb = (a.n_slots + EXP_CHUNK - 1) / EXP_CHUNK;
    k_export_fill<<<(unsigned)(nb < 256 * 8 ? nb : 256 * 8), 256, 0, s>>>(a);
    OTTO_HIP(hipGetLastError());
    return 0;
}

// The same two passes over a RANGE of run slots (chunked exchange: the fill of chunk c + 1 runs while chunk c is on the
// links). The counts travel through the caller (plan_range -> fill_range), not through the context.
extern "C" int otto_covis_export_plan_range(otto_covis_ctx* c, int n_owners, const uint32_t* h_bounds, int64_t slot_lo, int64_t slot_hi,
                                            int64_t* h_n_runs, int64_t* h_n_recs, void* stream) {
    OTTO_REQUIRE(h_n_runs && h_n_recs, "null argument");
    hipStream_t s = (hipStream_t)stream;
    OwnerArgs a;
    OTTO_TRY(owner_args(c, n_owners, h_bounds, a, slot_lo, slot_hi));
    OTTO_REQUIRE(c->run_used < (1ull << (64 - EXP_REC_BITS)) && c->rec_used < (1ull << EXP_REC_BITS), "too many runs for the packed export cursor");
    OTTO_TRY(c->exp_totals.ensure(2 * MAX_OWNERS * 8, 0, s));
    OTTO_HIP(hipMemsetAsync(c->exp_totals.p, 0, 2 * MAX_OWNERS * 8, s));
    a.totals = c->exp_totals.as<unsigned long long>();
    if (a.n_slots) {
        const int64_t nb = (a.n_slots + 255) / 256;
        k_export_plan<<<(unsigned)(nb < 256 * 8 ? nb : 256 * 8), 256, 0, s>>>(a);
        OTTO_HIP(hipGetLastError());
    }
    unsigned long long t[MAX_OWNERS];
    OTTO_HIP(hipMemcpyAsync(t, c->exp_totals.p, sizeof t, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    for (int o = 0; o < n_owners; ++o) {
        h_n_runs[o] = (int64_t)(t[o] >> EXP_REC_BITS);
        h_n_recs[o] = (int64_t)(t[o] & ((1ull << EXP_REC_BITS) - 1ull));
    }
    c->exp_planned = 0;
    return 0;
}

extern "C" int otto_covis_export_fill_range(otto_covis_ctx* c, int n_owners, const uint32_t* h_bounds, int64_t slot_lo, int64_t slot_hi,
                                            const int64_t* h_n_runs, const int64_t* h_n_recs, uint32_t* d_hdr, uint32_t* d_rec,
                                            uint32_t* d_tw, void* stream) {
    OTTO_REQUIRE(h_n_runs && h_n_recs, "null argument");
    hipStream_t s = (hipStream_t)stream;
    OwnerArgs a;
    OTTO_TRY(owner_args(c, n_owners, h_bounds, a, slot_lo, slot_hi));
    OTTO_REQUIRE(!d_tw || c->p.want_time, "no time channel to export");
    uint64_t rb = 0, cb = 0;
    for (int o = 0; o < n_owners; ++o) {
        OTTO_REQUIRE(h_n_runs[o] >= 0 && h_n_recs[o] >= 0, "negative planned count");
        a.run_base[o] = rb;
        a.rec_base[o] = cb;
        rb += (uint64_t)h_n_runs[o];
        cb += (uint64_t)h_n_recs[o];
    }
    if (rb == 0) return 0;
    OTTO_REQUIRE(d_hdr && d_rec, "null export buffers");
    OTTO_TRY(c->exp_totals.ensure(2 * MAX_OWNERS * 8, 0, s));
    OTTO_HIP(hipMemsetAsync(c->exp_totals.p, 0, 2 * MAX_OWNERS * 8, s));
    a.totals = c->exp_totals.as<unsigned long long>();
    a.rec = c->rec.as<uint32_t>();
    a.tw = c->tw.as<uint32_t>();
    a.o_hdr = d_hdr; a.o_rec = d_rec; a.o_tw = d_tw;
    const int64_t nb = (a.n_slots + EXP_CHUNK - 1) / EXP_CHUNK;
    k_export_fill<<<(unsigned)(nb < 256 * 8 ? nb : 256 * 8), 256, 0, s>>>(a);
    OTTO_HIP(hipGetLastError());
    c->exp_planned = 0;
    return 0;
}

extern "C" int otto_covis_import_reserve(otto_covis_ctx* c, int64_t n_recs, uint32_t** d_rec, uint32_t** d_tw, void* stream) {
    OTTO_REQUIRE(c && d_rec && d_tw && n_recs >= 0, "otto_covis_import_reserve: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)(n_recs > 0 ? n_recs : 1);
    OTTO_TRY(c->rec.ensure((size_t)(c->rec_used + n + REC_PAD) * 4, (size_t)c->rec_used * 4, s));
    if (c->p.want_time) OTTO_TRY(c->tw.ensure((size_t)(c->rec_used + n) * 4, (size_t)c->rec_used * 4, s));
    *d_rec = c->rec.as<uint32_t>() + c->rec_used;
    *d_tw = c->p.want_time ? c->tw.as<uint32_t>() + c->rec_used : nullptr;
    return 0;
}

extern "C" int otto_covis_import_runs(otto_covis_ctx* c, const uint32_t* d_hdr, int64_t n_runs, const uint32_t* d_rec,
                                      const uint32_t* d_tw, int64_t n_recs, void* stream) {
    OTTO_REQUIRE(c, "null ctx");
    if (n_runs == 0) return 0;
    OTTO_REQUIRE(d_hdr && d_rec && n_runs > 0 && n_recs > 0, "bad import buffers");
    OTTO_REQUIRE(!c->p.want_time || d_tw, "context keeps the time channel: d_tw required");
    hipStream_t s = (hipStream_t)stream;
    OTTO_TRY(c->exp_rec_pos.ensure((size_t)(n_runs + 1) * 8, 0, s));
    OTTO_TRY(c->partial.ensure(scan_partial_bytes(n_runs), 0, s));
    OTTO_TRY(device_scan(HdrLen{d_hdr}, n_runs, c->exp_rec_pos.as<uint64_t>(), c->partial.as<uint64_t>(), s));
    OTTO_TRY(c->rec.ensure((size_t)(c->rec_used + n_recs + REC_PAD) * 4, (size_t)c->rec_used * 4, s));
    if (c->p.want_time) OTTO_TRY(c->tw.ensure((size_t)(c->rec_used + n_recs) * 4, (size_t)c->rec_used * 4, s));
    OTTO_TRY(c->run_x.ensure((size_t)(c->run_used + n_runs) * 4, (size_t)c->run_used * 4, s));
    OTTO_TRY(c->run_desc.ensure((size_t)(c->run_used + n_runs) * 8, (size_t)c->run_used * 8, s));
    // records received in place (otto_covis_import_reserve): nothing to copy
    if (d_rec != c->rec.as<uint32_t>() + c->rec_used)
        OTTO_HIP(hipMemcpyAsync(c->rec.as<uint32_t>() + c->rec_used, d_rec, (size_t)n_recs * 4, hipMemcpyDeviceToDevice, s));
    if (c->p.want_time && d_tw != c->tw.as<uint32_t>() + c->rec_used)
        OTTO_HIP(hipMemcpyAsync(c->tw.as<uint32_t>() + c->rec_used, d_tw, (size_t)n_recs * 4, hipMemcpyDeviceToDevice, s));
    int grid = (int)((n_runs + 255) / 256 < 256 * 32 ? (n_runs + 255) / 256 : 256 * 32);
    k_import<<<grid, 256, 0, s>>>(d_hdr, n_runs, c->exp_rec_pos.as<uint64_t>(), c->rec_used, c->run_used,
                                  c->run_x.as<uint32_t>(), c->run_desc.as<uint64_t>());
    OTTO_HIP(hipGetLastError());
    c->rec_used += (uint64_t)n_recs;
    c->run_used += (uint64_t)n_runs;
    c->index_valid = false;
    c->desc_totals_valid = false;
    return 0;
}

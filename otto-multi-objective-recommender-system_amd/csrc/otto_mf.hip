// Matrix-factorization kernels for MI355X (gfx950). C-ABI in include/otto_mf.h.
//
//  * forward / eval        : gather two embedding rows, fp32 dot (torch_modules.py:13-19, 32-38),
//                            MSE / BCE-with-logits loss (torch_trainer.py:126-145)
//  * step_sparse_adam      : train() body (torch_trainer.py:59-78) with torch.optim.SparseAdam semantics.
//                            Duplicate rows inside a batch are coalesced WITHOUT a sort: each row is claimed
//                            by its smallest occurrence id (atomicMin on a per-row owner word), every
//                            occurrence adds its gradient row into the owner's slot of a [2B, d] buffer
//                            (256-B contiguous float atomics), the owner applies the non-linear Adam update.
//  * bpr_step              : counter-based negative sampling + fused gather/dot/sigmoid/SGD scatter
//                            (hogwild: racing plain stores; batch: the same owner scheme, deterministic up to
//                            fp32 atomic order)
//  * score_topk            : U[B,d] x V[N,d]^T on v_mfma_f32_32x32x2_f32 (exact f32) fused with a running
//                            per-row top-k; the B x N score matrix is never materialised
//                            (recbole/inference.py:76-80 materialises and copies it to the host)
#include "common.h"
#include "../../include/otto_mf.h"

#include <math.h>
#include <string.h>

namespace otto {

constexpr int32_t OWNER_FREE = 0x7F7F7F7F;   // memset(0x7F) pattern; larger than any occurrence id

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// loss and d(loss)/d(out) (without the 1/B of reduction='mean')
__device__ __forceinline__ void loss_grad(int kind, float out, float t, float* loss, float* g) {
    if (kind == OTTO_MF_LOSS_MSE) {
        const float e = out - t;
        *loss = e * e;
        *g = 2.0f * e;
    } else {
        // BCEWithLogits: max(x,0) - x*t + log(1 + exp(-|x|)); grad = sigmoid(x) - t
        const float ax = fabsf(out);
        *loss = fmaxf(out, 0.0f) - out * t + log1pf(expf(-ax));
        const float s = out >= 0.0f ? 1.0f / (1.0f + expf(-out)) : expf(out) / (1.0f + expf(out));
        *g = s - t;
    }
}

// block partial sums -> partial[blockIdx]; finalised by k_loss_final (deterministic order)
template <int THREADS>
__device__ __forceinline__ void block_loss_partial(float v, float* partial) {
    __shared__ float s_w[THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < THREADS / 64; ++i) t += s_w[i];
        partial[blockIdx.x] = t;
    }
}

// out = sum of partial[0..n) * scale; sums (nullable): sums[q] += sum of partial[(q + 1) * stride ..) in double -- the
// running score sums of validate() (one launch per batch on one stream: the additions are ordered, no atomics)
__global__ __launch_bounds__(256) void k_loss_final(const float* partial, int n, float scale, float* out, double* sums,
                                                    int n_sums, int stride, double count) {
    __shared__ float s_w[4];
    for (int q = 0; q <= n_sums; ++q) {
        float v = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) v += partial[q * stride + i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if (lane_id() == 0) s_w[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (q == 0) *out = t * scale;
            else sums[q - 1] += (double)t;
        }
    }
    if (threadIdx.x == 0 && sums) sums[n_sums] += count;
}

// ---------------------------------------------------------------------------
// forward / eval / step phase 1
//   MODE 0: pred only; 1: pred (optional) + loss partials; 3: + score partials of validate() (|e|, e^2, hits)
// Samples whose row ids fall outside the tables are skipped (pred = NaN) and counted in *err: the caller sees
// OTTO_EINVAL from otto_mf_check instead of a memory fault.
// ---------------------------------------------------------------------------
constexpr int MF_GRID_MAX = 256 * 8;

struct FwdArgs {
    const float* E1;
    const float* E2;
    const int64_t* i1;
    const int64_t* i2;
    const int64_t* target;
    int64_t B;
    int64_t n1, n2;
    int d;
    int G;          // lanes per sample = d / 4
    int loss_kind;
    float* pred;    // nullable
    float* partial; // [4][MF_GRID_MAX]
    uint32_t* err;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_mf_fwd(FwdArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t groups_per_block = 256 / G;
    const int64_t g0 = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / G;
    const int64_t gstride = (int64_t)gridDim.x * groups_per_block;
    float lsum = 0.f, s_abs = 0.f, s_sq = 0.f, s_hit = 0.f;
    for (int64_t b = g0; b < a.B; b += gstride) {
        const int64_t r1 = a.i1[b], r2 = a.i2[b];
        if ((uint64_t)r1 >= (uint64_t)a.n1 || (uint64_t)r2 >= (uint64_t)a.n2) {
            if (gl == 0) {
                atomicAdd(a.err, 1u);
                if (a.pred) a.pred[b] = __builtin_nanf("");
            }
            continue;
        }
        const float4 e1 = ld4(a.E1 + r1 * a.d + 4 * gl);
        const float4 e2 = ld4(a.E2 + r2 * a.d + 4 * gl);
        const float out = group_sum(dot4(e1, e2), G);
        if (gl == 0) {
            if (a.pred) a.pred[b] = out;
            if (MODE >= 1) {
                float l, g;
                const float t = (float)a.target[b];
                loss_grad(a.loss_kind, out, t, &l, &g);
                lsum += l;
                if (MODE == 3) {
                    // what validate() scores: the raw output (MSE models) or sigmoid(output) (BCE models) against the target;
                    // hit = (probability >= 0.5) == label, i.e. accuracy at the reference's threshold 0.5
                    const float p = a.loss_kind == OTTO_MF_LOSS_MSE ? out : 1.0f / (1.0f + expf(-out));
                    const float e = p - t;
                    s_abs += fabsf(e);
                    s_sq += e * e;
                    s_hit += ((p >= 0.5f) == (t >= 0.5f)) ? 1.0f : 0.0f;
                }
            }
        }
    }
    if (MODE >= 1) block_loss_partial<256>(lsum, a.partial);
    if (MODE == 3) {
        __syncthreads();
        block_loss_partial<256>(s_abs, a.partial + MF_GRID_MAX);
        __syncthreads();
        block_loss_partial<256>(s_sq, a.partial + 2 * MF_GRID_MAX);
        __syncthreads();
        block_loss_partial<256>(s_hit, a.partial + 3 * MF_GRID_MAX);
    }
}

// ---------------------------------------------------------------------------
// SparseAdam step (train() body). torch coalesces duplicate rows of the sparse gradient before the non-linear
// update; here, without a sort:
//   phase 1  k_rmf_fwd   : forward + loss + dL/dout per sample; every occurrence bumps its row's counter with ONE
//                          returning atomic. The SECOND arriver of a row learns that the row is duplicated: it zeroes
//                          its own slot of the [2B, d] gradient buffer and publishes the slot id in slot[row].
//   phase 2  k_rmf_acc   : rows that occur once in the batch (almost every session row, most aid rows of the long tail):
//                          gradient row in registers -> Adam on (p, m, v) in place, no gradient buffer traffic at all.
//                          Duplicated rows: every occurrence adds its gradient row into the published slot
//                          (global_atomic_add_f32, 16 bytes per lane, contiguous per row).
//   phase 3  k_rmf_apply : the second arriver of each duplicated row applies Adam from its slot and clears the counter.
// Counters are zero between steps (phase 2 / 3 clear exactly the rows they finish).
// ---------------------------------------------------------------------------
struct StepArgs {
    float* E1; float* m1; float* v1;
    float* E2; float* m2; float* v2;
    const int64_t* i1;
    const int64_t* i2;
    const int64_t* target;
    int64_t B;
    int64_t n1, n2;
    int d;
    int G;
    int loss_kind;
    float* partial;
    float* coef;            // [B] dL/dout / B
    uint32_t* cnt1;         // [n1] occurrences of the row in this batch
    uint32_t* cnt2;         // [n2] (== cnt1 for a shared table)
    int32_t* slot1;         // [n1] duplicated rows: occurrence id whose gradient slot accumulates the row
    int32_t* slot2;
    uint8_t* role;          // [2B] 1: this occurrence applies its slot in phase 3
    float* grad;            // [2B, d]
    uint32_t* err;
    float omb1, omb2, eps, step_size;   // 1-beta1, 1-beta2
};

__device__ __forceinline__ float adam1(float g, float& m, float& v, float omb1, float omb2, float eps, float step_size) {
    // torch/optim/_functional.py sparse_adam: m += (1-b1)(g-m); v += (1-b2)(g^2-v); p -= step_size*m/(sqrt(v)+eps)
    // (omb = float(1 - beta) with the subtraction done in double on the host, as torch does)
    m = m + (g - m) * omb1;
    v = v + (g * g - v) * omb2;
    return -step_size * (m / (sqrtf(v) + eps));
}

__device__ __forceinline__ void adam_row(float* E, float* M, float* V, float4 p, float4 g, const StepArgs& a) {
    float4 m = ld4(M), v = ld4(V);
    p.x += adam1(g.x, m.x, v.x, a.omb1, a.omb2, a.eps, a.step_size);
    p.y += adam1(g.y, m.y, v.y, a.omb1, a.omb2, a.eps, a.step_size);
    p.z += adam1(g.z, m.z, v.z, a.omb1, a.omb2, a.eps, a.step_size);
    p.w += adam1(g.w, m.w, v.w, a.omb1, a.omb2, a.eps, a.step_size);
    st4(E, p); st4(M, m); st4(V, v);
}

__device__ __forceinline__ bool rows_ok(const StepArgs& a, int64_t r1, int64_t r2) {
    return (uint64_t)r1 < (uint64_t)a.n1 && (uint64_t)r2 < (uint64_t)a.n2;
}

__global__ __launch_bounds__(256) void k_rmf_fwd(StepArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    float lsum = 0.f;
    const float invB = 1.0f / (float)a.B;
    for (int64_t b = (int64_t)blockIdx.x * gpb + threadIdx.x / G; b < a.B; b += (int64_t)gridDim.x * gpb) {
        const int64_t r1 = a.i1[b], r2 = a.i2[b];
        if (!rows_ok(a, r1, r2)) {
            if (gl == 0) { atomicAdd(a.err, 1u); a.role[b] = 0; a.role[a.B + b] = 0; }
            continue;
        }
        const float4 e1 = ld4(a.E1 + r1 * a.d + 4 * gl);
        const float4 e2 = ld4(a.E2 + r2 * a.d + 4 * gl);
        const float out = group_sum(dot4(e1, e2), G);
        uint32_t old1 = 0, old2 = 0;
        if (gl == 0) {
            float l, g;
            loss_grad(a.loss_kind, out, (float)a.target[b], &l, &g);
            lsum += l;
            a.coef[b] = g * invB;
            old1 = atomicAdd(&a.cnt1[r1], 1u);
            old2 = atomicAdd(&a.cnt2[r2], 1u);
            a.role[b] = old1 == 1u;
            a.role[a.B + b] = old2 == 1u;
            if (old1 == 1u) a.slot1[r1] = (int32_t)b;
            if (old2 == 1u) a.slot2[r2] = (int32_t)(a.B + b);
        }
        // the group leader's ranks, for every lane of the group (G <= 64 lanes inside one wave)
        old1 = (uint32_t)__shfl((int)old1, (int)((threadIdx.x & 63) - gl), 64);
        old2 = (uint32_t)__shfl((int)old2, (int)((threadIdx.x & 63) - gl), 64);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (old1 == 1u) st4(a.grad + b * a.d + 4 * gl, z);
        if (old2 == 1u) st4(a.grad + (a.B + b) * a.d + 4 * gl, z);
    }
    block_loss_partial<256>(lsum, a.partial);
}

__device__ __forceinline__ void atomic_add4(float* p, float4 v) {
    atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

__global__ __launch_bounds__(256) void k_rmf_acc(StepArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    for (int64_t b = (int64_t)blockIdx.x * gpb + threadIdx.x / G; b < a.B; b += (int64_t)gridDim.x * gpb) {
        const int64_t r1 = a.i1[b], r2 = a.i2[b];
        if (!rows_ok(a, r1, r2)) continue;
        const uint32_t c1 = a.cnt1[r1], c2 = a.cnt2[r2];
        const float c = a.coef[b];
        float* p1 = a.E1 + r1 * a.d + 4 * gl;
        float* p2 = a.E2 + r2 * a.d + 4 * gl;
        const float4 e1 = ld4(p1), e2 = ld4(p2);
        const float4 g1 = make_float4(c * e2.x, c * e2.y, c * e2.z, c * e2.w);
        const float4 g2 = make_float4(c * e1.x, c * e1.y, c * e1.z, c * e1.w);
        // every lane of the group has read both counters and both rows before anything is written
        __builtin_amdgcn_wave_barrier();
        if (c1 == 1u) {
            adam_row(p1, a.m1 + r1 * a.d + 4 * gl, a.v1 + r1 * a.d + 4 * gl, e1, g1, a);
            if (gl == 0) a.cnt1[r1] = 0;
        } else {
            atomic_add4(a.grad + (int64_t)a.slot1[r1] * a.d + 4 * gl, g1);
        }
        if (c2 == 1u) {
            adam_row(p2, a.m2 + r2 * a.d + 4 * gl, a.v2 + r2 * a.d + 4 * gl, e2, g2, a);
            if (gl == 0) a.cnt2[r2] = 0;
        } else {
            atomic_add4(a.grad + (int64_t)a.slot2[r2] * a.d + 4 * gl, g2);
        }
    }
}

__global__ __launch_bounds__(256) void k_rmf_apply(StepArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    for (int64_t o = (int64_t)blockIdx.x * gpb + threadIdx.x / G; o < 2 * a.B; o += (int64_t)gridDim.x * gpb) {
        if (!a.role[o]) continue;
        const bool first = o < a.B;
        const int64_t r = first ? a.i1[o] : a.i2[o - a.B];
        float* E = (first ? a.E1 : a.E2) + r * a.d + 4 * gl;
        adam_row(E, (first ? a.m1 : a.m2) + r * a.d + 4 * gl, (first ? a.v1 : a.v2) + r * a.d + 4 * gl, ld4(E),
                 ld4(a.grad + o * a.d + 4 * gl), a);
        if (gl == 0) (first ? a.cnt1 : a.cnt2)[r] = 0;
    }
}

// ---------------------------------------------------------------------------
// BPR
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// negative item for global row `row`: uniform over [0, n_items) (multiply-high of 32 random bits),
// redrawn (attempt 0..15) while equal to the positive; falls back to (pos + 1) % n_items
__host__ __device__ __forceinline__ int64_t bpr_negative(uint64_t seed, uint64_t epoch, uint64_t row, int64_t pos,
                                                         int64_t n_items) {
    const uint64_t base = mix64(seed ^ (epoch * 0xD1342543DE82EF95ull)) ^ (row * 0xA0761D6478BD642Full);
    for (uint64_t att = 0; att < 16; ++att) {
        const uint64_t r = mix64(base ^ (att * 0xE7037ED1A0B428DBull));
        const int64_t j = (int64_t)(((r >> 32) * (uint64_t)n_items) >> 32);
        if (j != pos) return j;
    }
    return (pos + 1) % n_items;
}

struct BprArgs {
    float* U;
    float* V;
    const int64_t* u;
    const int64_t* i;
    int64_t B;
    int64_t n_users;
    uint32_t* err;
    int64_t n_items;
    int d;
    int G;
    uint64_t seed, epoch;
    int64_t row0;
    float lr, l2;
    float* partial;
    int64_t* neg_out;
    // batch mode
    float* coef;      // [B] sigmoid(-x)
    int64_t* neg;     // [B]
    int32_t* ownerU;
    int32_t* ownerV;
    float* grad;      // [3B, d]
};

__device__ __forceinline__ float softplus_neg(float x) {   // -log(sigmoid(x)) = log(1 + exp(-x))
    return fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_neg(float x) {    // sigmoid(-x)
    return x >= 0.0f ? expf(-x) / (1.0f + expf(-x)) : 1.0f / (1.0f + expf(x));
}

// hogwild: one lane group per triplet, rows read, updated and written back in place (racing by design)
__global__ __launch_bounds__(256) void k_bpr_hogwild(BprArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    float lsum = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * gpb + threadIdx.x / G; b < a.B; b += (int64_t)gridDim.x * gpb) {
        const int64_t u = a.u[b], i = a.i[b];
        if ((uint64_t)u >= (uint64_t)a.n_users || (uint64_t)i >= (uint64_t)a.n_items) {   // skipped and reported (otto_mf_check)
            if (gl == 0) atomicAdd(a.err, 1u);
            continue;
        }
        const int64_t j = bpr_negative(a.seed, a.epoch, (uint64_t)(a.row0 + b), i, a.n_items);
        float* pu = a.U + u * a.d + 4 * gl;
        float* pi = a.V + i * a.d + 4 * gl;
        float* pj = a.V + j * a.d + 4 * gl;
        const float4 eu = ld4(pu), ei = ld4(pi), ej = ld4(pj);
        const float4 df = make_float4(ei.x - ej.x, ei.y - ej.y, ei.z - ej.z, ei.w - ej.w);
        const float x = group_sum(dot4(eu, df), G);
        const float s = sigmoid_neg(x);
        const float lr = a.lr, l2 = a.l2;
        st4(pu, make_float4(eu.x + lr * (s * df.x - l2 * eu.x), eu.y + lr * (s * df.y - l2 * eu.y),
                            eu.z + lr * (s * df.z - l2 * eu.z), eu.w + lr * (s * df.w - l2 * eu.w)));
        st4(pi, make_float4(ei.x + lr * (s * eu.x - l2 * ei.x), ei.y + lr * (s * eu.y - l2 * ei.y),
                            ei.z + lr * (s * eu.z - l2 * ei.z), ei.w + lr * (s * eu.w - l2 * ei.w)));
        st4(pj, make_float4(ej.x + lr * (-s * eu.x - l2 * ej.x), ej.y + lr * (-s * eu.y - l2 * ej.y),
                            ej.z + lr * (-s * eu.z - l2 * ej.z), ej.w + lr * (-s * eu.w - l2 * ej.w)));
        if (gl == 0) {
            lsum += softplus_neg(x);
            if (a.neg_out) a.neg_out[b] = j;
        }
    }
    block_loss_partial<256>(lsum, a.partial);
}

// batch mode phase 1: sample, score, claim owners
__global__ __launch_bounds__(256) void k_bpr_fwd(BprArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    float lsum = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * gpb + threadIdx.x / G; b < a.B; b += (int64_t)gridDim.x * gpb) {
        const int64_t u = a.u[b], i = a.i[b];
        if ((uint64_t)u >= (uint64_t)a.n_users || (uint64_t)i >= (uint64_t)a.n_items) {
            if (gl == 0) { atomicAdd(a.err, 1u); a.neg[b] = -1; }       // phases 2 and 3 skip the row on neg < 0
            continue;
        }
        const int64_t j = bpr_negative(a.seed, a.epoch, (uint64_t)(a.row0 + b), i, a.n_items);
        const float4 eu = ld4(a.U + u * a.d + 4 * gl), ei = ld4(a.V + i * a.d + 4 * gl), ej = ld4(a.V + j * a.d + 4 * gl);
        const float4 df = make_float4(ei.x - ej.x, ei.y - ej.y, ei.z - ej.z, ei.w - ej.w);
        const float x = group_sum(dot4(eu, df), G);
        if (gl == 0) {
            a.coef[b] = sigmoid_neg(x);
            a.neg[b] = j;
            if (a.neg_out) a.neg_out[b] = j;
            lsum += softplus_neg(x);
            atomicMin(&a.ownerU[u], (int32_t)b);
            atomicMin(&a.ownerV[i], (int32_t)(a.B + b));
            atomicMin(&a.ownerV[j], (int32_t)(2 * a.B + b));
        }
    }
    block_loss_partial<256>(lsum, a.partial);
}

// phase 2: accumulate the ascent direction of every occurrence into its row owner's slot
__global__ __launch_bounds__(256) void k_bpr_acc(BprArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    for (int64_t b = (int64_t)blockIdx.x * gpb + threadIdx.x / G; b < a.B; b += (int64_t)gridDim.x * gpb) {
        const int64_t u = a.u[b], i = a.i[b], j = a.neg[b];
        if (j < 0) continue;
        const float s = a.coef[b], l2 = a.l2;
        const float4 eu = ld4(a.U + u * a.d + 4 * gl), ei = ld4(a.V + i * a.d + 4 * gl), ej = ld4(a.V + j * a.d + 4 * gl);
        float* gu = a.grad + (int64_t)a.ownerU[u] * a.d + 4 * gl;
        float* gi = a.grad + (int64_t)a.ownerV[i] * a.d + 4 * gl;
        float* gj = a.grad + (int64_t)a.ownerV[j] * a.d + 4 * gl;
        atomicAdd(gu + 0, s * (ei.x - ej.x) - l2 * eu.x); atomicAdd(gu + 1, s * (ei.y - ej.y) - l2 * eu.y);
        atomicAdd(gu + 2, s * (ei.z - ej.z) - l2 * eu.z); atomicAdd(gu + 3, s * (ei.w - ej.w) - l2 * eu.w);
        atomicAdd(gi + 0, s * eu.x - l2 * ei.x); atomicAdd(gi + 1, s * eu.y - l2 * ei.y);
        atomicAdd(gi + 2, s * eu.z - l2 * ei.z); atomicAdd(gi + 3, s * eu.w - l2 * ei.w);
        atomicAdd(gj + 0, -s * eu.x - l2 * ej.x); atomicAdd(gj + 1, -s * eu.y - l2 * ej.y);
        atomicAdd(gj + 2, -s * eu.z - l2 * ej.z); atomicAdd(gj + 3, -s * eu.w - l2 * ej.w);
    }
}

// phase 3: owners apply p += lr * G and release the row
__global__ __launch_bounds__(256) void k_bpr_apply(BprArgs a) {
    const int G = a.G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t gpb = 256 / G;
    for (int64_t o = (int64_t)blockIdx.x * gpb + threadIdx.x / G; o < 3 * a.B; o += (int64_t)gridDim.x * gpb) {
        const int which = (int)(o / a.B);
        const int64_t b = o - (int64_t)which * a.B;
        if (a.neg[b] < 0) continue;
        const int64_t r = which == 0 ? a.u[b] : (which == 1 ? a.i[b] : a.neg[b]);
        int32_t* owner = which == 0 ? a.ownerU : a.ownerV;
        if (owner[r] != (int32_t)o) continue;
        float* P = (which == 0 ? a.U : a.V) + r * a.d + 4 * gl;
        const float4 g = ld4(a.grad + o * a.d + 4 * gl);
        float4 p = ld4(P);
        p.x += a.lr * g.x; p.y += a.lr * g.y; p.z += a.lr * g.z; p.w += a.lr * g.w;
        st4(P, p);
        __builtin_amdgcn_wave_barrier();
        if (gl == 0) owner[r] = OWNER_FREE;
    }
}

// ---------------------------------------------------------------------------
// full-sort scoring + top-k on f32 MFMA
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SC_BM = 128;      // rows per workgroup (4 waves x 32)
constexpr int SC_BN = 32;       // items per tile
constexpr int SC_MAXK = 32;

struct ScoreArgs {
    const float* U;
    const float* V;
    int64_t B, N;
    int k;
    int64_t pad_col;
    int nsplit;
    int64_t items_per_split;   // multiple of SC_BN
    float* part_s;             // [nsplit][Bpad][k]
    int32_t* part_i;
};

// insert (score, id) into the sorted (desc) k-list of one row in LDS; the whole wave cooperates.
// Returns the new k-th score (threshold).
__device__ __forceinline__ float list_insert(volatile float* ls, volatile int32_t* li, int k, float sc, int32_t id) {
    const unsigned l = lane_id();
    const float cur = l < (unsigned)k ? ls[l] : -INFINITY;
    // entries that stay ahead of the newcomer: higher score (ids arrive in ascending order, so on a
    // tie the resident entry has the smaller id and stays ahead)
    const bool ahead = l < (unsigned)k && cur >= sc;
    const int pos = __popcll(__ballot(ahead));
    const int32_t curi = l < (unsigned)k ? li[l] : 0;
    __builtin_amdgcn_wave_barrier();
    if ((int)l >= pos && (int)l + 1 < k) { ls[l + 1] = cur; li[l + 1] = curi; }
    if ((int)l == pos && pos < k) { ls[l] = sc; li[l] = id; }
    __builtin_amdgcn_wave_barrier();
    return ls[k - 1];
}

template <int D>
__global__ __launch_bounds__(256) void k_score(ScoreArgs a) {
    constexpr int HD = D / 2;               // k range of one lane half
    constexpr int LDV = D + 4;              // padded LDS row (floats): conflict-free ds_read_b128
    __shared__ float s_v[SC_BN * LDV];
    __shared__ float s_ls[4][32][SC_MAXK];
    __shared__ int32_t s_li[4][32][SC_MAXK];

    const int wid = threadIdx.x >> 6;
    const unsigned l = lane_id();
    const int r = l & 31, h = l >> 5;
    const int64_t row_tile = blockIdx.x;
    const int split = blockIdx.y;
    const int64_t row = row_tile * SC_BM + wid * 32 + r;

    // A fragments: U[row][h*HD + s], s = 0..HD-1, kept in registers for the whole item loop
    float ua[HD];
#pragma unroll
    for (int q = 0; q < HD; q += 4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.B) t = ld4(a.U + row * D + h * HD + q);
        ua[q] = t.x; ua[q + 1] = t.y; ua[q + 2] = t.z; ua[q + 3] = t.w;
    }
    for (int i = l; i < 32 * SC_MAXK; i += 64) { (&s_ls[wid][0][0])[i] = -INFINITY; (&s_li[wid][0][0])[i] = 0x7FFFFFFF; }
    float thr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) thr[q] = -INFINITY;

    const int64_t n_lo = (int64_t)split * a.items_per_split;
    int64_t n_hi = n_lo + a.items_per_split;
    if (n_hi > a.N) n_hi = a.N;

    // staging: thread t loads SC_BN*D/256 floats of the next V tile (contiguous float4s)
    constexpr int F4_PER_THREAD = (SC_BN * D / 4 + 255) / 256;
    float4 stage[F4_PER_THREAD];
    auto load_tile = [&](int64_t n0) {
#pragma unroll
        for (int q = 0; q < F4_PER_THREAD; ++q) {
            const int f = threadIdx.x + q * 256;          // float4 index inside the tile
            const int item = f / (D / 4), c4 = f % (D / 4);
            stage[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < SC_BN * D / 4 && n0 + item < n_hi) stage[q] = ld4(a.V + (n0 + item) * D + 4 * c4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < F4_PER_THREAD; ++q) {
            const int f = threadIdx.x + q * 256;
            const int item = f / (D / 4), c4 = f % (D / 4);
            if (f < SC_BN * D / 4) st4(&s_v[item * LDV + 4 * c4], stage[q]);
        }
    };

    if (n_lo < n_hi) load_tile(n_lo);
    for (int64_t n0 = n_lo; n0 < n_hi; n0 += SC_BN) {
        __syncthreads();            // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (n0 + SC_BN < n_hi) load_tile(n0 + SC_BN);   // prefetch under the MFMAs

        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const float* vb = &s_v[r * LDV + h * HD];
#pragma unroll
        for (int q = 0; q < HD; q += 4) {
            const float4 b = ld4(vb + q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[q], b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[q + 1], b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[q + 2], b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[q + 3], b.w, acc, 0, 0, 0);
        }
        // acc[q]: item column = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
        const int64_t item = n0 + r;
        const bool item_ok = item < n_hi && item != a.pad_col;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float sc = item_ok ? acc[q] : -INFINITY;
            uint64_t m = __ballot(sc > thr[q]);
            while (m) {
                const int src = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                const int srow = (q & 3) + 8 * (q >> 2) + 4 * (src >> 5);
                const float s2 = __shfl(sc, src, 64);
                const int32_t id2 = (int32_t)(n0 + (src & 31));
                const float nt = list_insert(s_ls[wid][srow], s_li[wid][srow], a.k, s2, id2);
                if (h == (src >> 5)) thr[q] = nt;
                m &= __ballot(sc > thr[q]);
            }
        }
    }
    __syncthreads();
    // partial lists of this (row tile, split)
    const int64_t Bpad = (int64_t)gridDim.x * SC_BM;
    for (int i = l; i < 32 * a.k; i += 64) {
        const int rr = i / a.k, c = i % a.k;
        const int64_t grow = row_tile * SC_BM + wid * 32 + rr;
        const int64_t o = ((int64_t)split * Bpad + grow) * a.k + c;
        a.part_s[o] = s_ls[wid][rr][c];
        a.part_i[o] = s_li[wid][rr][c];
    }
}

// merge nsplit partial lists per row: one wave per row, (score desc, id asc)
__global__ __launch_bounds__(64) void k_score_merge(const float* part_s, const int32_t* part_i, int nsplit, int64_t Bpad,
                                                    int64_t B, int k, int32_t* ids, float* scores) {
    const int64_t row = blockIdx.x;
    if (row >= B) return;
    const unsigned l = lane_id();
    float bs = -INFINITY;
    int32_t bi = 0x7FFFFFFF;
    const int total = nsplit * k;
    for (int c0 = 0; c0 < total; c0 += 64) {
        const int c = c0 + (int)l;
        float cs = -INFINITY;
        int32_t ci = 0x7FFFFFFF;
        if (c < total) {
            const int64_t o = ((int64_t)(c / k) * Bpad + row) * k + (c % k);
            cs = part_s[o];
            ci = part_i[o];
            if (ci < 0) ci = 0x7FFFFFFF;          // -1: an empty slot of a gathered partial list
        }
        auto better = [](float s1, int32_t i1, float s2, int32_t i2) { return s1 > s2 || (s1 == s2 && i1 < i2); };
        float ts = __shfl(bs, k - 1, 64);
        int32_t ti = __shfl(bi, k - 1, 64);
        uint64_t m = __ballot(ci != 0x7FFFFFFF && better(cs, ci, ts, ti));
        while (m) {
            const int src = __ffsll((unsigned long long)m) - 1;
            const float s = __shfl(cs, src, 64);
            const int32_t id = __shfl(ci, src, 64);
            const float us = __shfl_up(bs, 1, 64);
            const int32_t ui = __shfl_up(bi, 1, 64);
            if (better(s, id, bs, bi)) {
                if (l > 0 && better(s, id, us, ui)) { bs = us; bi = ui; }
                else { bs = s; bi = id; }
            }
            ts = __shfl(bs, k - 1, 64);
            ti = __shfl(bi, k - 1, 64);
            m &= m - 1;
            m &= __ballot(ci != 0x7FFFFFFF && better(cs, ci, ts, ti));
        }
    }
    if ((int)l < k) {
        ids[row * k + l] = bi == 0x7FFFFFFF ? -1 : bi;
        scores[row * k + l] = bs;
    }
}

}  // namespace otto

// ===========================================================================
// host side
// ===========================================================================
using namespace otto;

struct otto_mf_ctx {
    int64_t n1, n2, max_batch;
    int d, G, shared;
    DevBuf owner1, owner2;          // BPR batch mode: row -> smallest occurrence id (OWNER_FREE between steps)
    DevBuf cnt1, cnt2, slot1, slot2, role;   // SparseAdam step: occurrence counters (zero between steps), slots, roles
    DevBuf grad, coef, neg, partial;
    DevBuf err;                     // [0] u32: samples skipped because a row id was outside its table (sticky until read)
    DevBuf sums;                    // [4] double: running sum|e|, sum e^2, hits, count of otto_mf_eval_sums
    void release_all() {
        DevBuf* all[] = {&owner1, &owner2, &cnt1, &cnt2, &slot1, &slot2, &role, &grad, &coef, &neg, &partial, &err, &sums};
        for (DevBuf* b : all) b->release();
    }
};

static int mf_grid(int64_t B, int G) {
    const int64_t gpb = 256 / G;
    int64_t g = (B + gpb - 1) / gpb;
    const int64_t cap = MF_GRID_MAX;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static bool valid_d(int d) { return d == 4 || d == 8 || d == 16 || d == 32 || d == 64 || d == 128 || d == 256; }

extern "C" int otto_mf_create(otto_mf_ctx** out, int64_t n1, int64_t n2, int32_t d, int64_t max_batch, int32_t shared_table) {
    OTTO_REQUIRE(out, "null ctx pointer");
    OTTO_REQUIRE(valid_d(d), "d must be one of 4,8,16,32,64,128,256 (got %d)", d);
    OTTO_REQUIRE(n1 > 0 && (shared_table || n2 > 0), "table sizes must be positive");
    OTTO_REQUIRE(max_batch > 0 && 3 * max_batch < 0x7F000000ll, "max_batch out of range");
    otto_mf_ctx* c = new (std::nothrow) otto_mf_ctx();
    OTTO_REQUIRE(c, "out of host memory");
    c->n1 = n1; c->n2 = shared_table ? n1 : n2; c->d = d; c->G = d / 4; c->shared = shared_table != 0;
    c->max_batch = max_batch;
    int rc = 0;
    auto fail = [&](int code) { c->release_all(); delete c; return code; };
    // per-row words of table 1 (and of table 2 unless the table is shared)
    if ((rc = c->owner1.ensure((size_t)n1 * 4, 0, 0)) || (rc = c->cnt1.ensure((size_t)n1 * 4, 0, 0)) ||
        (rc = c->slot1.ensure((size_t)n1 * 4, 0, 0)))
        return fail(rc);
    if (!c->shared && ((rc = c->owner2.ensure((size_t)n2 * 4, 0, 0)) || (rc = c->cnt2.ensure((size_t)n2 * 4, 0, 0)) ||
                       (rc = c->slot2.ensure((size_t)n2 * 4, 0, 0))))
        return fail(rc);
    if ((rc = c->grad.ensure((size_t)3 * max_batch * d * 4, 0, 0)) || (rc = c->coef.ensure((size_t)max_batch * 4, 0, 0)) ||
        (rc = c->neg.ensure((size_t)max_batch * 8, 0, 0)) || (rc = c->partial.ensure((size_t)4 * MF_GRID_MAX * 4, 0, 0)) ||
        (rc = c->role.ensure((size_t)2 * max_batch, 0, 0)) || (rc = c->err.ensure(64, 0, 0)) || (rc = c->sums.ensure(64, 0, 0)))
        return fail(rc);
    bool ok = hipMemset(c->owner1.p, 0x7F, (size_t)n1 * 4) == hipSuccess && hipMemset(c->cnt1.p, 0, (size_t)n1 * 4) == hipSuccess &&
              hipMemset(c->err.p, 0, 64) == hipSuccess && hipMemset(c->sums.p, 0, 64) == hipSuccess;
    if (ok && !c->shared)
        ok = hipMemset(c->owner2.p, 0x7F, (size_t)n2 * 4) == hipSuccess && hipMemset(c->cnt2.p, 0, (size_t)n2 * 4) == hipSuccess;
    if (!ok) {
        set_error("hipMemset of the per-row words failed");
        return fail(-5);
    }
    *out = c;
    return 0;
}

extern "C" void otto_mf_destroy(otto_mf_ctx* c) {
    if (!c) return;
    c->release_all();
    delete c;
}

static int check_batch(otto_mf_ctx* c, int64_t B) {
    OTTO_REQUIRE(c, "null ctx");
    OTTO_REQUIRE(B > 0 && B <= c->max_batch, "batch %lld outside (0, max_batch=%lld]", (long long)B, (long long)c->max_batch);
    return 0;
}

extern "C" int otto_mf_check(otto_mf_ctx* c, int64_t* n_bad, void* stream) {
    OTTO_REQUIRE(c, "null ctx");
    hipStream_t s = (hipStream_t)stream;
    uint32_t bad = 0;
    OTTO_HIP(hipMemcpyAsync(&bad, c->err.p, 4, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    if (n_bad) *n_bad = (int64_t)bad;
    if (bad) {
        OTTO_HIP(hipMemsetAsync(c->err.p, 0, 4, s));
        set_error("%u sample(s) carried a row id outside the embedding tables (n1 = %lld, n2 = %lld) and were skipped", bad,
                  (long long)c->n1, (long long)c->n2);
        return -22;
    }
    return 0;
}

extern "C" int otto_mf_forward(otto_mf_ctx* c, const float* E1, const float* E2, const int64_t* i1, const int64_t* i2,
                               int64_t B, float* pred, void* stream) {
    OTTO_TRY(check_batch(c, B));
    OTTO_REQUIRE(E1 && E2 && i1 && i2 && pred, "null argument");
    FwdArgs a{E1, E2, i1, i2, nullptr, B, c->n1, c->n2, c->d, c->G, 0, pred, nullptr, c->err.as<uint32_t>()};
    k_mf_fwd<0><<<mf_grid(B, c->G), 256, 0, (hipStream_t)stream>>>(a);
    OTTO_HIP(hipGetLastError());
    return 0;
}

static int eval_impl(otto_mf_ctx* c, const float* E1, const float* E2, const int64_t* i1, const int64_t* i2,
                     const int64_t* target, int64_t B, int32_t loss_kind, float* pred, float* loss_out, bool sums, void* stream) {
    OTTO_TRY(check_batch(c, B));
    OTTO_REQUIRE(E1 && E2 && i1 && i2 && target && loss_out, "null argument");
    OTTO_REQUIRE(loss_kind == OTTO_MF_LOSS_MSE || loss_kind == OTTO_MF_LOSS_BCE, "unknown loss kind %d", loss_kind);
    hipStream_t s = (hipStream_t)stream;
    const int grid = mf_grid(B, c->G);
    FwdArgs a{E1, E2, i1, i2, target, B, c->n1, c->n2, c->d, c->G, loss_kind, pred, c->partial.as<float>(), c->err.as<uint32_t>()};
    if (sums) {
        k_mf_fwd<3><<<grid, 256, 0, s>>>(a);
        k_loss_final<<<1, 256, 0, s>>>(c->partial.as<float>(), grid, 1.0f / (float)B, loss_out, c->sums.as<double>(), 3, MF_GRID_MAX, (double)B);
    } else {
        k_mf_fwd<1><<<grid, 256, 0, s>>>(a);
        k_loss_final<<<1, 256, 0, s>>>(c->partial.as<float>(), grid, 1.0f / (float)B, loss_out, nullptr, 0, MF_GRID_MAX, 0.0);
    }
    OTTO_HIP(hipGetLastError());
    return 0;
}

extern "C" int otto_mf_eval(otto_mf_ctx* c, const float* E1, const float* E2, const int64_t* i1, const int64_t* i2,
                            const int64_t* target, int64_t B, int32_t loss_kind, float* pred, float* loss_out, void* stream) {
    return eval_impl(c, E1, E2, i1, i2, target, B, loss_kind, pred, loss_out, false, stream);
}

extern "C" int otto_mf_eval_sums(otto_mf_ctx* c, const float* E1, const float* E2, const int64_t* i1, const int64_t* i2,
                                 const int64_t* target, int64_t B, int32_t loss_kind, float* pred, float* loss_out, void* stream) {
    return eval_impl(c, E1, E2, i1, i2, target, B, loss_kind, pred, loss_out, true, stream);
}

extern "C" int otto_mf_read_sums(otto_mf_ctx* c, double* h_sums, int32_t reset, void* stream) {
    OTTO_REQUIRE(c && h_sums, "null argument");
    hipStream_t s = (hipStream_t)stream;
    OTTO_HIP(hipMemcpyAsync(h_sums, c->sums.p, 32, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    if (reset) OTTO_HIP(hipMemsetAsync(c->sums.p, 0, 32, s));
    return 0;
}

extern "C" int otto_mf_step_sparse_adam(otto_mf_ctx* c, float* E1, float* m1, float* v1, float* E2, float* m2, float* v2,
                                        const int64_t* i1, const int64_t* i2, const int64_t* target, int64_t B,
                                        int32_t loss_kind, double lr, double beta1, double beta2, double eps, int64_t t,
                                        float* loss_out, void* stream) {
    OTTO_TRY(check_batch(c, B));
    OTTO_REQUIRE(E1 && m1 && v1 && E2 && m2 && v2 && i1 && i2 && target && loss_out, "null argument");
    OTTO_REQUIRE(loss_kind == OTTO_MF_LOSS_MSE || loss_kind == OTTO_MF_LOSS_BCE, "unknown loss kind %d", loss_kind);
    OTTO_REQUIRE(t >= 1, "step count t must be >= 1");
    OTTO_REQUIRE(!c->shared || (E1 == E2 && m1 == m2 && v1 == v2), "shared_table context needs identical table pointers");
    hipStream_t s = (hipStream_t)stream;
    const int grid = mf_grid(B, c->G);
    // torch: step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t), evaluated in double on the host
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.E1 = E1; a.m1 = m1; a.v1 = v1; a.E2 = E2; a.m2 = m2; a.v2 = v2;
    a.i1 = i1; a.i2 = i2; a.target = target; a.B = B; a.n1 = c->n1; a.n2 = c->n2; a.d = c->d; a.G = c->G;
    a.loss_kind = loss_kind;
    a.partial = c->partial.as<float>(); a.coef = c->coef.as<float>();
    a.cnt1 = c->cnt1.as<uint32_t>(); a.cnt2 = c->shared ? a.cnt1 : c->cnt2.as<uint32_t>();
    a.slot1 = c->slot1.as<int32_t>(); a.slot2 = c->shared ? a.slot1 : c->slot2.as<int32_t>();
    a.role = c->role.as<uint8_t>(); a.grad = c->grad.as<float>(); a.err = c->err.as<uint32_t>();
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.step_size = (float)(lr * sqrt(bc2) / bc1);
    k_rmf_fwd<<<grid, 256, 0, s>>>(a);
    k_loss_final<<<1, 256, 0, s>>>(c->partial.as<float>(), grid, 1.0f / (float)B, loss_out, nullptr, 0, MF_GRID_MAX, 0.0);
    k_rmf_acc<<<grid, 256, 0, s>>>(a);
    k_rmf_apply<<<mf_grid(2 * B, c->G), 256, 0, s>>>(a);
    OTTO_HIP(hipGetLastError());
    return 0;
}

extern "C" int otto_mf_bpr_step(otto_mf_ctx* c, float* U, float* V, const int64_t* u, const int64_t* i, int64_t B,
                                uint64_t seed, uint64_t epoch, int64_t row0, float lr, float l2, int32_t mode,
                                float* loss_sum, int64_t* neg_out, void* stream) {
    OTTO_TRY(check_batch(c, B));
    OTTO_REQUIRE(U && V && u && i && loss_sum, "null argument");
    OTTO_REQUIRE(!c->shared, "BPR needs separate user and item tables");
    OTTO_REQUIRE(c->n2 >= 2, "BPR needs at least 2 items");
    OTTO_REQUIRE(mode == OTTO_MF_BPR_HOGWILD || mode == OTTO_MF_BPR_BATCH, "unknown BPR mode %d", mode);
    hipStream_t s = (hipStream_t)stream;
    const int grid = mf_grid(B, c->G);
    BprArgs a;
    memset(&a, 0, sizeof a);
    a.U = U; a.V = V; a.u = u; a.i = i; a.B = B; a.n_users = c->n1; a.err = c->err.as<uint32_t>(); a.n_items = c->n2; a.d = c->d; a.G = c->G;
    a.seed = seed; a.epoch = epoch; a.row0 = row0; a.lr = lr; a.l2 = l2;
    a.partial = c->partial.as<float>(); a.neg_out = neg_out;
    if (mode == OTTO_MF_BPR_HOGWILD) {
        k_bpr_hogwild<<<grid, 256, 0, s>>>(a);
    } else {
        a.coef = c->coef.as<float>(); a.neg = c->neg.as<int64_t>();
        a.ownerU = c->owner1.as<int32_t>(); a.ownerV = c->owner2.as<int32_t>(); a.grad = c->grad.as<float>();
        OTTO_HIP(hipMemsetAsync(c->grad.p, 0, (size_t)3 * B * c->d * 4, s));
        k_bpr_fwd<<<grid, 256, 0, s>>>(a);
        k_bpr_acc<<<grid, 256, 0, s>>>(a);
        k_bpr_apply<<<mf_grid(3 * B, c->G), 256, 0, s>>>(a);
    }
    k_loss_final<<<1, 256, 0, s>>>(c->partial.as<float>(), grid, 1.0f, loss_sum, nullptr, 0, MF_GRID_MAX, 0.0);
    OTTO_HIP(hipGetLastError());
    return 0;
}

static int score_nsplit(int64_t B, int64_t N) {
    const int64_t row_tiles = (B + SC_BM - 1) / SC_BM;
    int64_t ns = (1024 + row_tiles - 1) / row_tiles;
    const int64_t max_ns = (N + 32 * SC_BN - 1) / (32 * SC_BN);   // at least 32 tiles per split
    if (ns > max_ns) ns = max_ns;
    if (ns < 1) ns = 1;
    return (int)ns;
}

extern "C" int64_t otto_mf_score_workspace(int64_t B, int64_t N, int32_t k) {
    if (B <= 0 || N <= 0 || k <= 0) return 0;
    const int64_t Bpad = (B + SC_BM - 1) / SC_BM * SC_BM;
    return (int64_t)score_nsplit(B, N) * Bpad * k * 8;
}

extern "C" int otto_mf_score_topk(const float* U, const float* V, int64_t B, int64_t N, int32_t d, int32_t k,
                                  int64_t pad_col, int32_t* ids, float* scores, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
    OTTO_REQUIRE(U && V && ids && scores && workspace, "null argument");
    OTTO_REQUIRE(B > 0 && N > 0 && N < 0x7FFFFFFF, "bad B/N");
    OTTO_REQUIRE(k >= 1 && k <= SC_MAXK, "k must be in [1, %d]", SC_MAXK);
    OTTO_REQUIRE(d == 8 || d == 16 || d == 32 || d == 64 || d == 128, "scoring supports d in {8,16,32,64,128} (got %d)", d);
    OTTO_REQUIRE(workspace_bytes >= otto_mf_score_workspace(B, N, k), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t row_tiles = (B + SC_BM - 1) / SC_BM;
    const int64_t Bpad = row_tiles * SC_BM;
    const int ns = score_nsplit(B, N);
    int64_t per = (N + ns - 1) / ns;
    per = (per + SC_BN - 1) / SC_BN * SC_BN;
    ScoreArgs a{U, V, B, N, k, pad_col, ns, per, (float*)workspace, (int32_t*)((char*)workspace + (size_t)ns * Bpad * k * 4)};
    dim3 grid((unsigned)row_tiles, (unsigned)ns);
    switch (d) {
        case 8: k_score<8><<<grid, 256, 0, s>>>(a); break;
        case 16: k_score<16><<<grid, 256, 0, s>>>(a); break;
        case 32: k_score<32><<<grid, 256, 0, s>>>(a); break;
        case 64: k_score<64><<<grid, 256, 0, s>>>(a); break;
        default: k_score<128><<<grid, 256, 0, s>>>(a); break;
    }
    OTTO_HIP(hipGetLastError());
    k_score_merge<<<(unsigned)B, 64, 0, s>>>(a.part_s, a.part_i, ns, Bpad, B, k, ids, scores);
    OTTO_HIP(hipGetLastError());
    return 0;
}

extern "C" int otto_mf_topk_merge(const float* part_scores, const int32_t* part_ids, int32_t n_lists, int64_t B, int32_t k,
                                  int32_t* ids, float* scores, void* stream) {
    OTTO_REQUIRE(part_scores && part_ids && ids && scores, "null argument");
    OTTO_REQUIRE(n_lists >= 1 && B > 0 && k >= 1 && k <= SC_MAXK, "bad n_lists / B / k");
    k_score_merge<<<(unsigned)B, 64, 0, (hipStream_t)stream>>>(part_scores, part_ids, n_lists, B, B, k, ids, scores);
    OTTO_HIP(hipGetLastError());
    return 0;
}

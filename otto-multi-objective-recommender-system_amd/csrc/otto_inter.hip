// (session, candidate) interaction features on the device (include/otto_inter.h; reference:
// src/ranker/interaction_feature_engineering.py:56-113). One wave per session: the session's events sit in LDS, every lane
// owns up to two candidates and walks the events once; session aggregates are wave reductions, per-aid aggregates go
// through global atomics into a [n_aids] accumulator array that a second kernel turns into the nine aid features.
#include "common.h"
#include "../../include/otto_inter.h"

#include <math.h>

namespace otto {

struct AidAcc {              // per candidate aid, over all sessions
    unsigned long long occ_sum, last_sum;
    double score_sum, score_sq;
    uint32_t rows, last_rows, occ_max, last_max;
    uint32_t score_max_bits;  // order-preserving image of the float
    uint32_t pad;
};

__device__ __forceinline__ uint32_t float_order(float f) {       // monotone float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_unorder(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64); v = t < v ? t : v; }
    return v;
}

// cand_off == nullptr: dense rows, session s owns cand[s * C, +C) (-1 padded); else the CSR rows of the ranker's candidate
// table, session s owns cand[cand_off[s], cand_off[s + 1]) (any length: the session's own aids + up to 100 candidates)
__global__ __launch_bounds__(256) void k_inter_rows(const uint32_t* aid, const uint8_t* type, const int64_t* off, int64_t n_sess,
                                                    const int32_t* cand, const float* score, int C, const int64_t* cand_off,
                                                    uint32_t n_aids, uint16_t* row, float* sess_feat, AidAcc* acc, uint32_t* err) {
    __shared__ __attribute__((aligned(16))) uint32_t s_aid[4][OTTO_INTER_MAX_SESSION];
    __shared__ __attribute__((aligned(16))) uint8_t s_type[4][OTTO_INTER_MAX_SESSION];
    const int w = threadIdx.x >> 6;
    const unsigned lane = lane_id();
    const float nanf_ = __builtin_nanf("");
    for (int64_t s = (int64_t)blockIdx.x * 4 + w; s < n_sess; s += (int64_t)gridDim.x * 4) {
        const int64_t lo = off[s];
        int64_t n64 = off[s + 1] - lo;
        if (n64 > OTTO_INTER_MAX_SESSION) {
            if (lane == 0) atomicAdd(err, 1u);
            n64 = OTTO_INTER_MAX_SESSION;
        }
        const int n = (int)n64;
        for (int i = (int)lane; i < n; i += 64) { s_aid[w][i] = aid[lo + i]; s_type[w][i] = type[lo + i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double sc_sum = 0, sc_sq = 0, occ_sum = 0, last_sum = 0;
        uint32_t rows = 0, last_rows = 0, occ_max = 0, last_max = 0, sc_max = 0, sc_min = 0xFFFFFFFFu;
        const int64_t cb = cand_off ? cand_off[s] : s * (int64_t)C;
        const int Cs = cand_off ? (int)(cand_off[s + 1] - cb) : C;
        for (int c0 = 0; c0 < Cs; c0 += 64) {
            const int c = c0 + (int)lane;
            const int32_t y = c < Cs ? cand[cb + c] : -1;
            uint32_t cnt[3] = {0, 0, 0}, last = 0;
            if (y >= 0) {
                // four events per LDS read, no branch: a loop of one dependent LDS read + branch per event runs at one LDS
                // round trip per event
#pragma unroll 2
                for (int i0 = 0; i0 < n; i0 += 4) {
                    const uint4 a4 = *reinterpret_cast<const uint4*>(&s_aid[w][i0]);
                    const uint32_t t4 = *reinterpret_cast<const uint32_t*>(&s_type[w][i0]);
                    const uint32_t aj[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t t = (t4 >> (8 * e)) & 0xFFu;
                        const uint32_t eq = (aj[e] == (uint32_t)y && i0 + e < n) ? 1u : 0u;
                        cnt[0] += eq & (t == 0 ? 1u : 0u); cnt[1] += eq & (t == 1 ? 1u : 0u); cnt[2] += eq & (t == 2 ? 1u : 0u);
                        last = eq ? (uint32_t)(i0 + e) + 1u : last;        // cumcount + 1 of the last occurrence (:57-65)
                    }
                }
                const uint32_t occ = cnt[0] + cnt[1] + cnt[2];
                uint16_t* r = row + ((size_t)cb + c) * OTTO_INTER_ROW_FEATURES;
                r[0] = (uint16_t)occ; r[1] = (uint16_t)last; r[2] = (uint16_t)cnt[0]; r[3] = (uint16_t)cnt[1]; r[4] = (uint16_t)cnt[2];
                const float f = score[cb + c];
                const uint32_t fo = float_order(f);
                sc_sum += (double)f; sc_sq += (double)f * (double)f;
                sc_max = fo > sc_max ? fo : sc_max; sc_min = fo < sc_min ? fo : sc_min;
                occ_sum += occ; occ_max = occ > occ_max ? occ : occ_max;
                ++rows;
                if (last) { ++last_rows; last_sum += last; last_max = last > last_max ? last : last_max; }
                if ((uint32_t)y < n_aids) {
                    AidAcc* a = acc + y;
                    atomicAdd(&a->rows, 1u);
                    atomicAdd(&a->score_sum, (double)f);
                    atomicAdd(&a->score_sq, (double)f * (double)f);
                    atomicMax(&a->score_max_bits, fo);
                    if (occ) { atomicAdd(&a->occ_sum, (unsigned long long)occ); atomicMax(&a->occ_max, occ); }
                    if (last) { atomicAdd(&a->last_rows, 1u); atomicAdd(&a->last_sum, (unsigned long long)last); atomicMax(&a->last_max, last); }
                } else {
                    atomicAdd(err, 1u);
                }
            } else if (c < Cs) {
                uint16_t* r = row + ((size_t)cb + c) * OTTO_INTER_ROW_FEATURES;
                r[0] = r[1] = r[2] = r[3] = r[4] = 0;
            }
        }
        // session features over the candidate rows (:87-98)
        const double t_sc = wave_sum(sc_sum), t_sq = wave_sum(sc_sq), t_occ = wave_sum(occ_sum), t_last = wave_sum(last_sum);
        const double nr = wave_sum((double)rows), nl = wave_sum((double)last_rows);
        const uint32_t m_sc = wave_max_u(sc_max), n_sc = wave_min_u(sc_min), m_occ = wave_max_u(occ_max), m_last = wave_max_u(last_max);
        if (lane == 0) {
            float* o = sess_feat + (size_t)s * OTTO_INTER_SESSION_FEATURES;
            if (nr > 0) {
                const double mean = t_sc / nr;
                o[0] = (float)mean;
                o[1] = nr > 1 ? (float)sqrt(fmax((t_sq - nr * mean * mean) / (nr - 1), 0.0)) : nanf_;
                o[2] = float_unorder(n_sc);
                o[3] = float_unorder(m_sc);
                o[4] = (float)(t_occ / nr);
                o[5] = (float)t_occ;
                o[6] = (float)m_occ;
                o[7] = nl > 0 ? (float)(t_last / nl) : nanf_;
                o[8] = (float)t_last;
                o[9] = nl > 0 ? (float)m_last : nanf_;
            } else {
                for (int q = 0; q < OTTO_INTER_SESSION_FEATURES; ++q) o[q] = nanf_;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_inter_aids(const AidAcc* acc, uint32_t n_aids, float* out) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_aids) return;
    const AidAcc a = acc[x];
    float* o = out + (size_t)x * OTTO_INTER_AID_FEATURES;
    const float nanf_ = __builtin_nanf("");
    if (a.rows == 0) {
        for (int q = 0; q < OTTO_INTER_AID_FEATURES; ++q) o[q] = nanf_;
        return;
    }
    const double nr = (double)a.rows, mean = a.score_sum / nr;
    o[0] = (float)mean;
    o[1] = a.rows > 1 ? (float)sqrt(fmax((a.score_sq - nr * mean * mean) / (nr - 1), 0.0)) : nanf_;
    o[2] = float_unorder(a.score_max_bits);
    o[3] = (float)((double)a.occ_sum / nr);
    o[4] = (float)a.occ_sum;
    o[5] = (float)a.occ_max;
    o[6] = a.last_rows ? (float)((double)a.last_sum / (double)a.last_rows) : nanf_;
    o[7] = (float)a.last_sum;
    o[8] = a.last_rows ? (float)a.last_max : nanf_;
}

}  // namespace otto

using namespace otto;

extern "C" int64_t otto_inter_workspace(uint32_t n_aids) { return (int64_t)n_aids * (int64_t)sizeof(AidAcc) + 256; }

extern "C" int otto_inter_features(const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off, int64_t n_sess,
                                   const int32_t* d_cand, const float* d_score, int32_t C, uint32_t n_aids, uint16_t* d_row,
                                   float* d_sess_feat, float* d_aid_feat, void* d_ws, int64_t ws_bytes, void* stream) {
    OTTO_REQUIRE(n_sess >= 0 && C >= 1 && C <= OTTO_INTER_MAX_CAND, "C must be in [1, %d]", OTTO_INTER_MAX_CAND);
    OTTO_REQUIRE(n_aids > 0 && d_aid_feat && d_ws, "otto_inter_features: null argument");
    OTTO_REQUIRE(ws_bytes >= otto_inter_workspace(n_aids), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    AidAcc* acc = (AidAcc*)((char*)d_ws + 256);
    uint32_t* err = (uint32_t*)d_ws;
    OTTO_HIP(hipMemsetAsync(d_ws, 0, (size_t)otto_inter_workspace(n_aids), s));
    if (n_sess > 0) {
        OTTO_REQUIRE(d_aid && d_type && d_sess_off && d_cand && d_score && d_row && d_sess_feat, "otto_inter_features: null argument");
        const int64_t blocks = (n_sess + 3) / 4;
        k_inter_rows<<<(unsigned)(blocks < 256 * 16 ? blocks : 256 * 16), 256, 0, s>>>(d_aid, d_type, d_sess_off, n_sess, d_cand, d_score, C,
                                                                                      nullptr, n_aids, d_row, d_sess_feat, acc, err);
        OTTO_HIP(hipGetLastError());
    }
    k_inter_aids<<<(n_aids + 255) / 256, 256, 0, s>>>(acc, n_aids, d_aid_feat);
    OTTO_HIP(hipGetLastError());
    uint32_t herr = 0;
    OTTO_HIP(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    OTTO_REQUIRE(herr == 0, "%u sessions longer than %d events or candidates outside [0, n_aids)", herr, OTTO_INTER_MAX_SESSION);
    return 0;
}

extern "C" int otto_inter_features_rows(const uint32_t* d_aid, const uint8_t* d_type, const int64_t* d_sess_off, int64_t n_sess,
                                        const int64_t* d_cand_off, const int32_t* d_cand, const float* d_score, uint32_t n_aids,
                                        uint16_t* d_row, float* d_sess_feat, float* d_aid_feat, void* d_ws, int64_t ws_bytes,
                                        void* stream) {
    OTTO_REQUIRE(n_sess >= 0 && n_aids > 0 && d_aid_feat && d_ws, "otto_inter_features_rows: bad argument");
    OTTO_REQUIRE(ws_bytes >= otto_inter_workspace(n_aids), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    AidAcc* acc = (AidAcc*)((char*)d_ws + 256);
    uint32_t* err = (uint32_t*)d_ws;
    OTTO_HIP(hipMemsetAsync(d_ws, 0, (size_t)otto_inter_workspace(n_aids), s));
    if (n_sess > 0) {
        OTTO_REQUIRE(d_aid && d_type && d_sess_off && d_cand_off && d_cand && d_score && d_row && d_sess_feat, "otto_inter_features_rows: null argument");
        const int64_t blocks = (n_sess + 3) / 4;
        k_inter_rows<<<(unsigned)(blocks < 256 * 16 ? blocks : 256 * 16), 256, 0, s>>>(d_aid, d_type, d_sess_off, n_sess, d_cand, d_score, 0,
                                                                                      d_cand_off, n_aids, d_row, d_sess_feat, acc, err);
        OTTO_HIP(hipGetLastError());
    }
    k_inter_aids<<<(n_aids + 255) / 256, 256, 0, s>>>(acc, n_aids, d_aid_feat);
    OTTO_HIP(hipGetLastError());
    uint32_t herr = 0;
    OTTO_HIP(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
    OTTO_HIP(hipStreamSynchronize(s));
    OTTO_REQUIRE(herr == 0, "%u sessions longer than %d events or candidates outside [0, n_aids)", herr, OTTO_INTER_MAX_SESSION);
    return 0;
}

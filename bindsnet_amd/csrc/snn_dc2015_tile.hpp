// snn_dc2015_tile.hpp -- device helpers shared by the resident forms of the DiehlAndCook2015 plan (snn_dc2015_resident.hip:
// general / lean / second-generation lean kernels; snn_dc2015_async.hip: third-generation lean kernel): the tagged-granule
// accessors of the spike exchange and PostPre on the LDS-resident [Nin][CW] weight slice.  Moved here verbatim from
// snn_dc2015_resident.hip in round 4 (tools/isa_diff.py: the existing kernels' device code is unchanged).
#pragma once
#include "snn_dc2015.hpp"

namespace {

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// PostPre on the LDS-resident slice: (row, column) items, rows listed in `arows` (all rows when FULL).
template <class SUM, bool FULL, int CWL, int NTL>
__device__ __forceinline__ void stdp_rows_lds(const DcCtx &c, int nact, const uint16_t *arows, const uint32_t *rowmask,
                                              const uint32_t *colmask, const uint8_t *__restrict__ sbytes,
                                              const float *xnu0, const float *__restrict__ xsrc, float *wtile, int c0,
                                              int tid, int Emain) {
    constexpr int CW = CWL, NT = NTL;                             // (tile width / workgroup size of the calling kernel)
    const int B = c.B, Nin = c.Nin, N = c.N;
    const int nitems = nact * CW;
    const int q = tid % CW, jq = c0 + q;                          // NT % CW == 0: a thread keeps its column
    if (jq >= N) return;
    const uint32_t cm = (c.nu1 != 0.f) ? colmask[q] : 0u;
    auto update = [&](int i, uint32_t m, float w) -> float {
        const int e = i * N + jq;
        if (c.nu0 != 0.f) {                                      // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
            float uu = 0.f;
            if (m) {
                SUM acc; acc.init(e >= Emain);
                while (m) {
                    const int b = __ffs(m) - 1; m &= m - 1;
                    const float sv = sbytes ? (float)sbytes[b * Nin + i] : 1.0f;
                    acc.add(b, sv * xnu0[b * CW + q], B);
                }
                uu = acc.finish(B);
            }
            if (c.use_dt) uu = uu * c.dt;
            w = w - uu;
        }
        if (c.nu1 != 0.f) {                                      // w += dt * sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
            uint32_t mm = cm;
            float uu = 0.f;
            if (mm) {
                SUM acc; acc.init(e >= Emain);
                while (mm) {
                    const int b = __ffs(mm) - 1; mm &= mm - 1;
                    acc.add(b, xsrc[b * Nin + i] * (1.0f * c.nu1), B);
                }
                uu = acc.finish(B);
            }
            if (c.use_dt) uu = uu * c.dt;
            w = w + uu;
        }
        if (c.has_min && w < c.wmin) w = c.wmin;
        if (c.has_max && w > c.wmax) w = c.wmax;
        return w;
    };
    // two items per round: their (independent) row index / mask / weight reads share the LDS latencies
    for (int it0 = tid; it0 < nitems; it0 += 2 * NT) {
        const int it1 = it0 + NT;
        const bool h1 = it1 < nitems;
        const int i0 = FULL ? (it0 / CW) : (int)arows[it0 / CW];
        const int i1 = h1 ? (FULL ? (it1 / CW) : (int)arows[it1 / CW]) : i0;
        const uint32_t m0 = rowmask[i0], m1 = rowmask[i1];
        const float w0 = wtile[i0 * CW + q], w1 = wtile[i1 * CW + q];
        wtile[i0 * CW + q] = update(i0, m0, w0);
        if (h1) wtile[i1 * CW + q] = update(i1, m1, w1);
    }
}

// The same update for 4-column tiles without `row_sum` tail elements and 0/1 spikes (the lean form), one thread per
// listed ROW: the row's four weights in registers (one 16-byte LDS access each way), x_tgt*nu0 of a sample read as one
// float4 for all columns.  The batch sum of a column (<= 32 terms) in ATen's order = one partial per block of 16 samples,
// the closed partials added in order, ((open + closed) + 0) + 0 at the end; a block change is applied through 0/1
// factors inside fmas whose products are exact (x*1, x*0), i.e. the same single rounding as the plain adds.  Columns
// with a post-synaptic spike (rare with one_spike) take the plain cascade.
template <int NTL>
__device__ __forceinline__ void stdp_rows4(const DcCtx &c, int nact, const uint16_t *arows, const uint32_t *rowmask,
                                           const uint32_t *colmask, const float *xnu0, const float *__restrict__ xsrc,
                                           float *wtile, int c0, int tid) {
    const int B = c.B, Nin = c.Nin, N = c.N;
    uint32_t cm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cm[q] = (c.nu1 != 0.f && c0 + q < N) ? (uint32_t)__builtin_amdgcn_readfirstlane(colmask[q]) : 0u;
    const bool whole = c0 + 4 <= N;
    for (int k = tid; k < nact; k += NTL) {
        const int i = (int)arows[k];
        uint32_t m = rowmask[i];
        const float4 w4 = *(const float4 *)(wtile + i * 4);
        float w[4] = {w4.x, w4.y, w4.z, w4.w};
        if (c.nu0 != 0.f) {                                      // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            int cblk = 0;
            while (m) {
                const int b = __ffs(m) - 1; m &= m - 1;
                const float4 xn = *(const float4 *)(xnu0 + b * 4);
                const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
                const float same = (b >> 4) == cblk ? 1.f : 0.f, diff = 1.f - same;
                cblk = b >> 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a1[q] = __builtin_fmaf(a0[q], diff, a1[q]);
                    a0[q] = __builtin_fmaf(a0[q], same, 1.0f * xv[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float uu = ((a0[q] + a1[q]) + 0.f) + 0.0f;
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] - uu;
            }
        }
        if (c.nu1 != 0.f) {                                      // w += dt * sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float uu = 0.f;
                if (cm[q]) {
                    CascadeT acc; acc.init(false);
                    uint32_t mm = cm[q];
                    while (mm) {
                        const int b = __ffs(mm) - 1; mm &= mm - 1;
                        acc.add(b, xsrc[b * Nin + i] * (1.0f * c.nu1), B);
                    }
                    uu = acc.finish(B);
                }
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] + uu;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c.has_min && w[q] < c.wmin) w[q] = c.wmin;
            if (c.has_max && w[q] > c.wmax) w[q] = c.wmax;
        }
        if (whole) *(float4 *)(wtile + i * 4) = make_float4(w[0], w[1], w[2], w[3]);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c0 + q < N) wtile[i * 4 + q] = w[q];
        }
    }
}

// Speculative PostPre of the second-generation lean kernel (k_dc2015_spec): stdp_rows4 under the assumption that no own
// column has a post-synaptic spike, by the `nthreads` threads qt = 0.., from the committed weights `wsrc` (left as they
// are) into `wdst`.  full: every row (the first update of a run clamps every element).
__device__ __forceinline__ void spec_rows4(const DcCtx &c, bool full, int nact, const uint16_t *arows, const uint32_t *rowmask,
                                           const float *xnu0, const float *wsrc, float *wdst, int c0, int qt, int nthreads) {
    for (int k = qt; k < nact; k += nthreads) {
        const int i = full ? k : (int)arows[k];
        uint32_t m = rowmask[i];
        const float4 w4 = *(const float4 *)(wsrc + i * 4);
        float w[4] = {w4.x, w4.y, w4.z, w4.w};
        if (c.nu0 != 0.f) {                                      // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            int cblk = 0;
            while (m) {
                const int b = __ffs(m) - 1; m &= m - 1;
                const float4 xn = *(const float4 *)(xnu0 + b * 4);
                const float xv[4] = {xn.x, xn.y, xn.z, xn.w};
                const float same = (b >> 4) == cblk ? 1.f : 0.f, diff = 1.f - same;
                cblk = b >> 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a1[q] = __builtin_fmaf(a0[q], diff, a1[q]);
                    a0[q] = __builtin_fmaf(a0[q], same, 1.0f * xv[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float uu = ((a0[q] + a1[q]) + 0.f) + 0.0f;
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] - uu;
            }
        }
        if (c.nu1 != 0.f) {                                      // + dt * (empty sum): what the update adds without a post-synaptic spike
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float uu = 0.f;
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] + uu;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c.has_min && w[q] < c.wmin) w[q] = c.wmin;
            if (c.has_max && w[q] > c.wmax) w[q] = c.wmax;
        }
        *(float4 *)(wdst + i * 4) = make_float4(w[0], w[1], w[2], w[3]);   // (columns >= N of the last tile: unused values)
    }
}

// Columns with a post-synaptic spike x rows WITHOUT a pre-synaptic spike, on the LDS-resident slice.
template <class SUM, int CWL, int NTL>
__device__ __forceinline__ void stdp_cols_lds(const DcCtx &c, uint32_t active_cols, const uint32_t *rowmask,
                                              const uint32_t *colmask, const float *__restrict__ xsrc, float *wtile,
                                              int c0, int tid, int Emain) {
    constexpr int CW = CWL, NT = NTL;
    const int B = c.B, Nin = c.Nin, N = c.N;
    while (active_cols) {
        const int q = __ffs(active_cols) - 1; active_cols &= active_cols - 1;
        const uint32_t cm = colmask[q];
        const int jq = c0 + q;
        for (int i = tid; i < Nin; i += NT) {
            if (rowmask[i]) continue;
            const int e = i * N + jq;
            float w = wtile[i * CW + q];
            SUM acc; acc.init(e >= Emain);
            uint32_t m = cm;
            while (m) {
                const int b = __ffs(m) - 1; m &= m - 1;
                acc.add(b, xsrc[b * Nin + i] * (1.0f * c.nu1), B);
            }
            float uu = acc.finish(B);
            if (c.use_dt) uu = uu * c.dt;
            w = w + uu;
            if (c.has_min && w < c.wmin) w = c.wmin;
            if (c.has_max && w > c.wmax) w = c.wmax;
            wtile[i * CW + q] = w;
        }
    }
}

}  // namespace

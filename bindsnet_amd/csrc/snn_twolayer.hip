// snn_twolayer.hip -- fused plan "twolayer-fused": Input -> {Connection | MulticompartmentConnection+Weight}
// (PostPre or no rule) -> LIFNodes, i.e. bindsnet/models/models.py:21-91 (TwoLayerNetwork) and every graph
// of that shape, for B <= 32 samples per GPU.
//
// Nothing couples two target neurons of such a network: no lateral connection, no shared threshold, no
// arbitration.  The only state a column slice of the weights needs from outside is SOURCE-side -- the
// input spikes and the input layer's trace -- and both are functions of the inputs alone, which are known
// for the whole run when run() starts.  So:
//   k_two_xtrace  precomputes the input trace after every step           (parallel over (sample, source))
//   k_two_prep    digests every step's spikes: per-sample ascending event lists (CSR), the rows that carry a
//                 spike in any sample with their sample masks, a bitmap of those rows   (parallel over steps)
//   k_two_run     ONE launch for the whole run: workgroup g owns CW target columns; its [Nin x CW] weight slice
//                 lives in LDS and the membrane state of its (sample, column) pairs in registers for all T
//                 steps; per step it applies PostPre of the previous step to the LDS tile, sums the currents in
//                 the reference's order, steps the LIF neurons, writes the rasters.  No launch boundary, no
//                 inter-workgroup traffic, no weight traffic to HBM until the final write-back.
// Arithmetic order is the reference's (snn_order.hpp / snn_common.hpp); results are bit-identical to the
// generic plan (tests/test_gpu_twolayer.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/snnhip.h"
#include "snn_common.hpp"
#include "snn_order.hpp"

using namespace snn;

bool snn_prof_active();
void snn_set_plan_name(const char *name);

namespace {

constexpr int NT = 1024;
constexpr int MAXB = 128;       // samples per run: sample masks are MW = ceil(B / 32) words wide (one word up to batch 32)
constexpr int META = 136;       // [0] list entries, [1] active rows, [2] flags (1: spike byte > 1, 2: list overflow), [4..4+B] CSR offsets
constexpr int PF = 16;          // digest words prefetched per thread per step

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct TwoCtx {
    int B, Nin, N, T, NinW, CW, G, MW, BC;               // MW mask words per sample set, BC = 32 * MW (array capacity)
    float dt; int learning;
    const uint8_t *in; const uint8_t *sX0;
    const float *xX0; float *xXout; float *xall;       // entry trace, exit trace, [T][B][Nin] trace after each step
    int x_traces; float x_decay, x_scale; int x_additive;
    float *vY, *rY, *xY; uint8_t *sY;
    snn_lif_params pY;
    uint8_t *rasY; float *rasVY;
    float *W; const float *bias;
    int cascade;                                         // 1: MCC (ATen sum order), 0: dense Connection (ascending sequential)
    int rule; float nu0, nu1; int use_dt; float wdecay; int has_min; float wmin; int has_max; float wmax;
    uint32_t *dig; int DW, LCAP, o_ent, o_am, o_ar, o_ab, o_xw;   // digest: words per entry, list capacity, word offsets
    float inv_hwps;
    int prodw;                                           // floats of LDS for the staged products of the dense dot (0: off)
    int mstdp_rows;                                      // MSTDP in its row-per-thread forms (developer switch SNN_TWO_MSTDP_ROWS=0: off)
    int rowmajor;                                        // PostPre in its row-major form (developer switch SNN_TWO_ROWMAJOR=0: off)
    int use_xsl;                                         // stage the source traces of spiking columns in LDS (fits + Nin <= NT)
    // MSTDP (learning.py:1504-1574), factored eligibility: p_plus / p_minus traces, previous-step spike factors
    float *p_plus, *p_minus; uint8_t *s_src_prev, *s_tgt_prev;
    float *pall;                                         // [T+1][B][Nin] p_plus at entry / after every step
    float a_plus, a_minus, d_plus, d_minus, reward; const float *reward_vec;
    int has_norm; float norm; int norm_abs;              // post-run normalisation, done on the LDS tile in the epilogue
    long long *dbg;                                      // developer aid (SNN_TWO_TIMING=1): phase timestamps of workgroup 0
};

#define WMARK() do { if (c.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) c.dbg[(size_t)8 * 4096 + (size_t)t * 16 + (threadIdx.x >> 6)] = (long long)wall_clock64(); } while (0)
#define TMARK(slot) do { if (c.dbg && blockIdx.x == 0 && threadIdx.x == 0) c.dbg[(size_t)t * 8 + (slot)] = (long long)wall_clock64(); } while (0)

__device__ __forceinline__ uint32_t nz4(uint32_t w) {
    const uint32_t t = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu;
}

// ---------------------------------------------------------------------------------------------- input trace
__global__ __launch_bounds__(256) void k_two_xtrace(const TwoCtx c) {
    const int n = c.B * c.Nin;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float x = c.xX0[k];
    int t = 0;
    for (; t + 4 <= c.T; t += 4) {           // the four spike loads are independent of x: issue them together
        uint8_t s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = c.in[(size_t)(t + u) * n + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x = trace_next(x, s[u], c.x_decay, c.x_scale, c.x_additive); if (c.xall) c.xall[(size_t)(t + u) * n + k] = x; }
    }
    for (; t < c.T; ++t) { x = trace_next(x, c.in[(size_t)t * n + k], c.x_decay, c.x_scale, c.x_additive); if (c.xall) c.xall[(size_t)t * n + k] = x; }
    c.xXout[k] = x;
}

// MSTDP's source trace p_plus after every step (learning.py:1564-1565): entry 0 = value at run entry, entry
// k+1 = after step k.  It depends on the inputs alone, like the X trace.
__global__ __launch_bounds__(256) void k_two_pplus(const TwoCtx c) {
    const int n = c.B * c.Nin;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float x = c.p_plus[k];
    c.pall[k] = x;
    int t = 0;
    for (; t + 4 <= c.T; t += 4) {
        uint8_t s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = c.in[(size_t)(t + u) * n + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float p = x * c.d_plus; x = p + c.a_plus * (float)s[u]; c.pall[(size_t)(t + u + 1) * n + k] = x; }
    }
    for (; t < c.T; ++t) { const float p = x * c.d_plus; x = p + c.a_plus * (float)c.in[(size_t)t * n + k]; c.pall[(size_t)(t + 1) * n + k] = x; }
    c.p_plus[k] = x;
}

// ---------------------------------------------------------------------------------------------- spike digest
// Entry e digests the source spikes the step kernel sees as "previous step" in iteration e:
// e = 0 the layer's `s` at entry, e >= 1 inputs[e-1].  One workgroup per entry.
__global__ __launch_bounds__(NT) void k_two_prep(const TwoCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, NinW = c.NinW;
    uint32_t *sXw = (uint32_t *)smem;                    // [B][NinW] bit words
    uint32_t *rowmask = sXw + B * NinW;                  // [Nin][MW] samples in which the row spiked
    const int mw = c.MW;
    int *cntb = (int *)(rowmask + Nin * mw);             // [B+1] per-sample counts -> offsets
    int *misc = cntb + MAXB + 1;                         // [0] nact, [1] flags
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.x;
    // (MSTDP: one more entry, T+1 = the source spikes the rule remembers from its last update before this run)
    const uint8_t *src = (e == 0) ? c.sX0 : (e == c.T + 1 ? c.s_src_prev : c.in + (size_t)(e - 1) * B * Nin);
    uint32_t *D = c.dig + (size_t)e * c.DW;
    uint16_t *D_ent = (uint16_t *)(D + c.o_ent), *D_ar = (uint16_t *)(D + c.o_ar);
    uint32_t *D_am = D + c.o_am, *D_ab = D + c.o_ab, *D_xw = D + c.o_xw;
    for (int k = tid; k < Nin * mw; k += NT) rowmask[k] = 0;
    for (int k = tid; k < NinW; k += NT) D_ab[k] = 0;    // (own entry; finished before the atomics below by the barrier)
    if (tid < 2) misc[tid] = 0;
    __syncthreads();
    {
        const int total16 = (B * Nin) >> 4, hwps = Nin >> 4, HS = NinW * 2;
        uint16_t *sXh = (uint16_t *)sXw;
        uint32_t big = 0;
        for (int k16 = tid; k16 < total16; k16 += NT) {
            const uint4 v = ((const uint4 *)src)[k16];
            const int b = (int)(((float)k16 + 0.5f) * c.inv_hwps), hw = k16 - b * hwps;
            const uint32_t any = v.x | v.y | v.z | v.w;
            uint32_t m16 = 0;
            if (any) {
                if (any & 0xFEFEFEFEu) {
                    big = 1;
                    m16 = nz4(v.x) | (nz4(v.y) << 4) | (nz4(v.z) << 8) | (nz4(v.w) << 12);
                } else {                                  // 0/1 bytes: the multiply gathers the four LSBs into bits 24..27
                    m16 = ((v.x * 0x01020408u) >> 24) | (((v.y * 0x01020408u) >> 24) << 4) |
                          (((v.z * 0x01020408u) >> 24) << 8) | (((v.w * 0x01020408u) >> 24) << 12);
                }
            }
            sXh[b * HS + hw] = (uint16_t)m16;
            if (hw == hwps - 1 && (hwps & 1)) sXh[b * HS + hw + 1] = 0;
            while (m16) {
                const int i = hw * 16 + __ffs(m16) - 1; m16 &= m16 - 1;
                atomicOr(&rowmask[i * mw + (b >> 5)], 1u << (b & 31));
            }
        }
        if (big) atomicOr((unsigned int *)&misc[1], 1u);
    }
    __syncthreads();
    // per-sample spike counts (one wave per sample), then CSR offsets
    for (int b = wave; b < B; b += NT / 64) {
        int n = 0;
        for (int w = lane; w < NinW; w += 64) n += __popc(sXw[b * NinW + w]);
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
        if (lane == 0) cntb[b + 1] = n;
    }
    __syncthreads();
    if (tid == 0) { cntb[0] = 0; for (int b = 0; b < B; ++b) cntb[b + 1] += cntb[b]; }
    __syncthreads();
    const int total = cntb[B];
    // ascending event lists: wave per sample, 64 words per round, running offset
    for (int b = wave; b < B; b += NT / 64) {
        int base = cntb[b];
        const uint64_t below = (1ull << lane) - 1ull;
        for (int w0 = 0; w0 < NinW; w0 += 64) {
            const int w = w0 + lane;
            uint32_t m = w < NinW ? sXw[b * NinW + w] : 0u;
            const int cn = __popc(m);
            int offp = 0, tot = 0;
            for (int k = 0;; ++k) {
                const uint64_t bm = __ballot(cn > k);
                if (!bm) break;
                offp += __popcll(bm & below); tot += __popcll(bm);
            }
            offp += base;
            while (m) {
                const int i = w * 32 + __ffs(m) - 1; m &= m - 1;
                if (offp < c.LCAP) D_ent[offp] = (uint16_t)i;
                ++offp;
            }
            base += tot;
        }
    }
    for (int k = tid; k < B * NinW; k += NT) D_xw[k] = sXw[k];
    for (int base = 0; base < Nin; base += NT) {          // rows with a spike in any sample, with their sample masks
        const int i = base + tid;
        bool o = false;
        if (i < Nin) for (int w = 0; w < mw; ++w) o = o || rowmask[i * mw + w] != 0;
        const uint64_t m = __ballot(o);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&misc[0], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (o) {
            const int cp = wbase + __popcll(m & ((1ull << lane) - 1ull));
            D_ar[cp] = (uint16_t)i;
            for (int w = 0; w < mw; ++w) D_am[cp * mw + w] = rowmask[i * mw + w];
            atomicOr(&D_ab[i >> 5], 1u << (i & 31));
        }
    }
    __syncthreads();
    if (tid <= B) D[4 + tid] = (uint32_t)cntb[tid];
    if (tid == 0) { D[0] = (uint32_t)total; D[1] = (uint32_t)misc[0]; D[2] = (uint32_t)(misc[1] | (total > c.LCAP ? 2 : 0)); }
}

// ---------------------------------------------------------------------------------------------- the run
template <class SUM>
__device__ __forceinline__ float list_dot(const float *wt, int CW, int jj, const uint16_t *ent, int n0, int n1,
                                          const uint8_t *__restrict__ vals, int n_terms) {
    SUM a; a.init();
    for (int k = n0; k < n1; k += 8) {       // eight independent LDS reads per round, then the ordered adds
        int ix[8]; float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ix[u] = (int)ent[min(k + u, n1 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = wt[ix[u] * CW + jj];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k + u < n1) a.add(ix[u], wv[u] * (vals ? (float)vals[ix[u]] : 1.0f), n_terms);
    }
    return a.finish(n_terms);
}

struct RowSumN {          // RowSum4 with the init() signature of the other accumulators
    RowSum4 r;
    __device__ __forceinline__ void init() { r.init(); }
    __device__ __forceinline__ void add(int pos, float term, int n) { r.add(pos, term, n); }
    __device__ __forceinline__ float finish(int n) { return r.finish(n); }
};
struct SeqN {
    float a;
    __device__ __forceinline__ void init() { a = 0.f; }
    __device__ __forceinline__ void add(int, float term, int) { a += term; }
    __device__ __forceinline__ float finish(int) { return a; }
};

// PostPre of one step on the LDS weight tile (MCC_learning.py:224-302 / learning.py:390-420 + base update).
// SUM = CascadeT unless an element index can fall in ATen's <32-element tail of the [Nin*N] batch reduction.
// All global loads a thread needs (source traces of the samples whose neuron spiked) are issued before the
// order-constrained arithmetic; the first contributing sample of a column -- almost always the only one -- is
// prefetched, further ones are fetched on demand.
struct CascT {             // batch sums have at most 32 terms: the branch-free cascade applies
    CascadeFlat c;
    __device__ __forceinline__ void init(bool) { c.init(); }
    __device__ __forceinline__ void add(int pos, float term, int n) { c.add(pos, term, n); }
    __device__ __forceinline__ float finish(int n) { return c.finish(n); }
};

// RL: PostPre, or the two rules with the same outer-product skeleton (generic plan: k_plasticity modes 2 / 3, same
// arithmetic) -- Hebbian  w += nu0 * U1; w += nu1 * U2  (learning.py:1052-1135) and WeightDependentPostPre
// w += 0 - (nu0 U1)(w - wmin) + (nu1 U2)(wmax - w)  (learning.py:562-653), U1 = sum_b s_src x_tgt, U2 = sum_b x_src s_tgt;
// for these `xnu0` holds the plain target trace.
template <class SUM, int MWT, int RL = SNN_RULE_POSTPRE>
__device__ __forceinline__ void two_stdp(const TwoCtx &c, float *wt, const uint16_t *ar, const uint32_t *am,
                                         const uint32_t *ab, const float *xnu0, const uint32_t *cm,
                                         const float *__restrict__ xs, const float *xsl, const uint8_t *__restrict__ sbytes,
                                         int nact, bool full, int c0, int tid, int cwl, int Emain) {
    const int B = c.B, Nin = c.Nin, N = c.N, CW = c.CW, mw = MWT == 1 ? 1 : c.MW;   // (one mask word: the loops fold away)
    auto first_of = [&](const uint32_t *m) -> int { for (int w = 0; w < mw; ++w) if (m[w]) return w * 32 + __ffs(m[w]) - 1; return -1; };
    auto any_of = [&](const uint32_t *m) -> bool { uint32_t o = 0; for (int w = 0; w < mw; ++w) o |= m[w]; return o != 0; };
    // source trace of sample b at row i, for the post-synaptic term of column q: the first spiking sample of q is
    // staged in LDS (xsl), any further one comes from global memory
    auto xsrc = [&](int q, int b, int i, int first_b) -> float {
        return (xsl && b == first_b) ? xsl[q * Nin + i] : xs[b * Nin + i];
    };
    // ---- pass 1: rows with a pre-synaptic spike x own columns
    for (int item = tid; item < (nact << cwl); item += NT) {
        const int kq = item >> cwl, q = item & (CW - 1);
        const int i = (int)ar[kq];
        const uint32_t *m = am + kq * mw;                 // samples in which row i spiked
        const uint32_t *cq = cm + q * mw;                 // samples in which column q spiked
        const bool cany = c.nu1 != 0.f && any_of(cq);
        if (c0 + q >= N) continue;
        const bool tl = i * N + c0 + q >= Emain;
        float w = wt[i * CW + q];
        if constexpr (RL != SNN_RULE_POSTPRE) {
            float u1, u2 = 0.f;
            {
                SUM acc; acc.init(tl);
                for (int wd = 0; wd < mw; ++wd) {
                    uint32_t mm = m[wd];
                    while (mm) {
                        const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1;
                        const float sv = sbytes ? (float)sbytes[b * Nin + i] : 1.0f;
                        acc.add(b, sv * xnu0[b * 8 + q], B);
                    }
                }
                u1 = acc.finish(B);
            }
            if (any_of(cq)) {
                SUM acc; acc.init(tl);
                const int fb = first_of(cq);
                for (int wd = 0; wd < mw; ++wd) {
                    uint32_t mm = cq[wd];
                    while (mm) { const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1; acc.add(b, xsrc(q, b, i, fb) * 1.0f, B); }
                }
                u2 = acc.finish(B);
            }
            if constexpr (RL == SNN_RULE_HEBBIAN) {
                w = w + c.nu0 * u1;
                w = w + c.nu1 * u2;
            } else {
                float upd = 0.f; bool have = false;
                if (c.nu0 != 0.f) { upd = 0.0f - (c.nu0 * u1) * (w - c.wmin); have = true; }
                if (c.nu1 != 0.f) { const float y = (c.nu1 * u2) * (c.wmax - w); upd = have ? upd + y : y; have = true; }
                if (have) w = w + upd;
            }
        } else {
        if (c.nu0 != 0.f) {                               // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
            SUM acc; acc.init(tl);
            for (int wd = 0; wd < mw; ++wd) {             // ascending sample index = the reference's batch-sum order
                uint32_t mm = m[wd];
                while (mm) {
                    const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1;
                    const float sv = sbytes ? (float)sbytes[b * Nin + i] : 1.0f;
                    acc.add(b, sv * xnu0[b * 8 + q], B);
                }
            }
            float uu = acc.finish(B);
            if (c.use_dt) uu = uu * c.dt;
            w = w - uu;
        }
        if (c.nu1 != 0.f) {                               // w += dt * sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
            float uu = 0.f;
            if (cany) {
                SUM acc; acc.init(tl);
                const int fb = first_of(cq);
                for (int wd = 0; wd < mw; ++wd) {
                    uint32_t mm = cq[wd];
                    while (mm) { const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1; acc.add(b, xsrc(q, b, i, fb) * (1.0f * c.nu1), B); }
                }
                uu = acc.finish(B);
            }
            if (c.use_dt) uu = uu * c.dt;
            w = w + uu;
        }
        }
        w = w * c.wdecay;
        if (c.has_min && w < c.wmin) w = c.wmin;
        if (c.has_max && w > c.wmax) w = c.wmax;
        wt[i * CW + q] = w;
    }
    // ---- pass 2: rows WITHOUT a pre-synaptic spike: the columns that spiked (every column when `full`)
    uint32_t todo = 0;                                    // columns to visit, as a bit mask
    for (int q = 0; q < CW; ++q) if (c0 + q < N && (full || ((c.nu1 != 0.f || RL == SNN_RULE_HEBBIAN) && any_of(cm + q * mw)))) todo |= 1u << q;
    if (!todo) return;
    for (int i = tid; i < Nin; i += NT) {
        if ((ab[i >> 5] >> (i & 31)) & 1u) continue;
        uint32_t td = todo;
        while (td) {
            const int q = __ffs(td) - 1; td &= td - 1;
            const uint32_t *cq = cm + q * mw;
            const bool cany = c.nu1 != 0.f && any_of(cq);
            float w = wt[i * CW + q];
            if constexpr (RL != SNN_RULE_POSTPRE) {          // no pre-synaptic spike in this row: U1 is the empty sum
                float u2 = 0.f;
                if (any_of(cq)) {
                    SUM acc; acc.init(i * N + c0 + q >= Emain);
                    const int fb = first_of(cq);
                    for (int wd = 0; wd < mw; ++wd) {
                        uint32_t mm = cq[wd];
                        while (mm) { const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1; acc.add(b, xsrc(q, b, i, fb) * 1.0f, B); }
                    }
                    u2 = acc.finish(B);
                }
                if constexpr (RL == SNN_RULE_HEBBIAN) {
                    w = w + c.nu0 * 0.0f;
                    w = w + c.nu1 * u2;
                } else {
                    float upd = 0.f; bool have = false;
                    if (c.nu0 != 0.f) { upd = 0.0f - (c.nu0 * 0.0f) * (w - c.wmin); have = true; }
                    if (c.nu1 != 0.f) { const float y = (c.nu1 * u2) * (c.wmax - w); upd = have ? upd + y : y; have = true; }
                    if (have) w = w + upd;
                }
            } else {
            if (c.nu0 != 0.f) w = w - (c.use_dt ? 0.0f * c.dt : 0.0f);
            if (c.nu1 != 0.f) {
                float uu = 0.f;
                if (cany) {
                    SUM acc; acc.init(i * N + c0 + q >= Emain);
                    const int fb = first_of(cq);
                    for (int wd = 0; wd < mw; ++wd) {
                        uint32_t mm = cq[wd];
                        while (mm) { const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1; acc.add(b, xsrc(q, b, i, fb) * (1.0f * c.nu1), B); }
                    }
                    uu = acc.finish(B);
                }
                if (c.use_dt) uu = uu * c.dt;
                w = w + uu;
            }
            }
            w = w * c.wdecay;
            if (c.has_min && w < c.wmin) w = c.wmin;
            if (c.has_max && w > c.wmax) w = c.wmax;
            wt[i * CW + q] = w;
        }
    }
}

// PostPre, row-major form (used whenever no element can fall into ATen's <32-element tail, i.e. Nin*N % 32 == 0): thread <->
// source row i, all CW columns of the tile in registers.  The pre-synaptic term walks the row's own sample mask; the
// post-synaptic term walks the SAMPLES (union over the tile's columns of the samples whose neuron spiked, ascending) in
// the outer loop -- coalesced loads of x_src[b, :] for the whole workgroup, eight samples in flight at a time -- and
// feeds each column's own accumulator only for the samples of that column, so every element still sees its terms in
// ascending sample order.  Same arithmetic per element as two_stdp: w -= pre * dt; w += post * dt; w *= decay; clamp.
//
// Batch sums here have B <= 128 terms, for which ATen's cascade (snn_order.hpp CascadeFlat) is: one partial per block of
// 16 samples, the partials of the full blocks added up in ascending order, and the partial of the trailing short block
// (B % 16 samples) added to that: ((tail + sum_blocks) + 0) + 0.  Samples that contribute nothing add +0.0, which never
// changes a partial (partials start at +0.0 and so are never -0.0); the loops below skip them.  With a0 = the running
// partial of the current block and a1 = the sum of the closed ones, the result is ((a0 + a1) + 0) + 0 whether the last
// block is the tail (tail + a1) or a full one (0 + (a1 + a0)).
//
// Where a term applies to some columns only, it is multiplied by a 0/1 factor inside an fma: the product is exact
// (x*1, x*0), so the fma rounds once -- to the same value as the plain add (or no add) it stands for.
// RL (round 5): the two other rules of the same outer-product skeleton take the same walk -- Hebbian  w += nu0 U1; w += nu1 U2  and
// WeightDependentPostPre  w += 0 - (nu0 U1)(w - wmin) + (nu1 U2)(wmax - w)  with U1 = sum_b s_src x_tgt, U2 = sum_b x_src s_tgt (two_stdp's
// statements, element for element; `xnu0` then holds the plain target trace and the post-synaptic terms are x_src * 1.0f): the batch sums
// are the same partials in the same order, only what is done with them differs.  Hebbian visits the columns that spiked whether or not
// nu1 is zero, like two_stdp's second pass.
template <int MWT, int RL = SNN_RULE_POSTPRE>
__device__ __forceinline__ void two_stdp_rowmajor(const TwoCtx &c, float *wt, const uint32_t *am, const uint32_t *ab,
                                                  const uint16_t *ridx, const float *xnu0, const uint32_t *ul,
                                                  const float4 *fac, const float *__restrict__ xs, bool full, int c0, int tid) {
    const int B = c.B, Nin = c.Nin, N = c.N, CW = c.CW, mw = MWT == 1 ? 1 : c.MW;
    // the samples with a post-synaptic spike in this tile, ascending (bytes of ul[0..8 mw)), their number, the columns that
    // spiked at all, and per listed sample one 0/1 factor per column (fac): built by two_union_list
    const bool want_post = c.nu1 != 0.f || RL == SNN_RULE_HEBBIAN;
    const int nun = want_post ? __builtin_amdgcn_readfirstlane(ul[8 * mw]) : 0;
    const uint32_t postcols = want_post ? (uint32_t)__builtin_amdgcn_readfirstlane(ul[8 * mw + 1]) : 0u;
    const float nu1c = RL == SNN_RULE_POSTPRE ? 1.0f * c.nu1 : 1.0f;
    // x_src[b, i] = one buffer load: descriptor of the slab (scalar), row offset b*Nin*4 (scalar), lane offset i*4
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc((void *)xs, 0, B * Nin * 4, 0x00020000);
    for (int i = tid; i < Nin; i += NT) {
        const bool active = (ab[i >> 5] >> (i & 31)) & 1u;
        if (!full && !active && !postcols) continue;
        float w[8];
        if (CW == 8) {
            const float4 lo = *(const float4 *)(wt + i * 8), hi = *(const float4 *)(wt + i * 8 + 4);
            w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = q < CW ? wt[i * CW + q] : 0.f;
        }
        // eight samples per chunk; list entries past nun are stale but valid sample numbers: loaded, never added
        auto load_chunk = [&](int k0, uint32_t (&id)[2], float (&x)[8]) {
            id[0] = __builtin_amdgcn_readfirstlane(ul[k0 >> 2]);
            id[1] = __builtin_amdgcn_readfirstlane(ul[(k0 >> 2) + 1]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t b = (id[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                x[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(slab, i * 4, (int)(b * (uint32_t)Nin * 4u), 0));
            }
        };
        uint32_t idA[2], idB[2];
        float xA[8], xB[8];
        if (nun > 0) load_chunk(0, idA, xA);               // in flight behind the pre-synaptic part
        float a0[8], a1[8];
        float u1[8];                                       // (Hebbian / WeightDependentPostPre: the pre-synaptic batch sums, used below)
#pragma unroll
        for (int q = 0; q < 8; ++q) u1[q] = 0.f;
        if (c.nu0 != 0.f || RL != SNN_RULE_POSTPRE) {      // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
            // per lane: walk the row's samples in ascending order
#pragma unroll
            for (int q = 0; q < 8; ++q) a0[q] = a1[q] = 0.f;
            if (active) {
                uint32_t m[MWT];
                const uint32_t *mp = am + (int)ridx[i] * mw;
#pragma unroll
                for (int wd = 0; wd < MWT; ++wd) m[wd] = wd < mw ? mp[wd] : 0u;
                int cblk = 0;
#pragma unroll
                for (int wd = 0; wd < MWT; ++wd) {
                    uint32_t mm = m[wd];
                    while (mm) {
                        const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1;
                        const float4 lo = *(const float4 *)(xnu0 + b * 8), hi = *(const float4 *)(xnu0 + b * 8 + 4);
                        const float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        // a block change closes the running partial: a1 += a0, a0 = 0 (closing an empty block adds +0.0)
                        const float same = (b >> 4) == cblk ? 1.f : 0.f, diff = 1.f - same;
                        cblk = b >> 4;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            a1[q] = __builtin_fmaf(a0[q], diff, a1[q]);
                            a0[q] = __builtin_fmaf(a0[q], same, 1.0f * xv[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q >= CW) break;
                float uu = ((a0[q] + a1[q]) + 0.f) + 0.0f;
                if constexpr (RL != SNN_RULE_POSTPRE) { u1[q] = active ? uu : 0.f; continue; }   // (a row without a source spike: the empty sum, two_stdp's 0.0f)
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] - uu;
            }
        }
        if (want_post) {                                   // w += dt * sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
#pragma unroll
            for (int q = 0; q < 8; ++q) a0[q] = a1[q] = 0.f;
            int cblk = 0;
            auto accumulate = [&](int k0, const uint32_t (&id)[2], const float (&x)[8]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k0 + k >= nun) break;
                    const int b = (int)((id[k >> 2] >> ((k & 3) * 8)) & 0xFFu);
                    const float4 f0 = fac[2 * (k0 + k)], f1 = fac[2 * (k0 + k) + 1];   // (same address in every lane)
                    const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                    if ((b >> 4) != cblk) {
                        cblk = b >> 4;
#pragma unroll
                        for (int q = 0; q < 8; ++q) { a1[q] = a1[q] + a0[q]; a0[q] = 0.f; }
                    }
                    const float term = x[k] * nu1c;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a0[q] = __builtin_fmaf(term, f[q], a0[q]);
                }
            };
            for (int k0 = 0; k0 < nun; k0 += 16) {         // the next chunk's loads are in flight while this one is added up
                if (k0 + 8 < nun) load_chunk(k0 + 8, idB, xB);
                accumulate(k0, idA, xA);
                if (k0 + 8 >= nun) break;
                if (k0 + 16 < nun) load_chunk(k0 + 16, idA, xA);
                accumulate(k0 + 8, idB, xB);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q >= CW) break;
                float uu = ((postcols >> q) & 1u) ? ((a0[q] + a1[q]) + 0.f) + 0.0f : 0.f;
                if constexpr (RL != SNN_RULE_POSTPRE) { a0[q] = uu; continue; }                  // U2 of column q (0.0f where it did not spike)
                if (c.use_dt) uu = uu * c.dt;
                w[q] = w[q] + uu;
            }
        }
        if constexpr (RL != SNN_RULE_POSTPRE) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q >= CW) break;
                const float u2 = want_post ? a0[q] : 0.f;
                if constexpr (RL == SNN_RULE_HEBBIAN) {
                    w[q] = w[q] + c.nu0 * u1[q];
                    w[q] = w[q] + c.nu1 * u2;
                } else {
                    float upd = 0.f; bool have = false;
                    if (c.nu0 != 0.f) { upd = 0.0f - (c.nu0 * u1[q]) * (w[q] - c.wmin); have = true; }
                    if (c.nu1 != 0.f) { const float y = (c.nu1 * u2) * (c.wmax - w[q]); upd = have ? upd + y : y; have = true; }
                    if (have) w[q] = w[q] + upd;
                }
            }
        }
        const bool whole = CW == 8 && c0 + 8 <= N && (active || full || postcols == 0xFFu);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= CW) break;
            float v = w[q] * c.wdecay;
            if (c.has_min && v < c.wmin) v = c.wmin;
            if (c.has_max && v > c.wmax) v = c.wmax;
            w[q] = v;
            if (!whole && c0 + q < N && (active || full || ((postcols >> q) & 1u))) wt[i * CW + q] = v;
        }
        if (whole) {
            *(float4 *)(wt + i * 8) = make_float4(w[0], w[1], w[2], w[3]);
            *(float4 *)(wt + i * 8 + 4) = make_float4(w[4], w[5], w[6], w[7]);
        }
    }
}

// The row-major PostPre's view of one step's post-synaptic spikes, built by the first B threads once the step's spike
// masks are final: bytes of ul[0..8 mw) = the samples in which any column of the tile spiked (ascending), fac[2k], fac[2k+1]
// = 1.0 / 0.0 per column for the k-th of them, ul[8 mw] = how many, ul[8 mw + 1] = the columns that spiked at all.
template <int MWT>
__device__ __forceinline__ void two_union_list(const TwoCtx &c, const uint32_t *cmn, uint32_t *ul, float4 *fac, int mw, int c0, int tid) {
    if (tid >= c.B) return;
    const int w = tid >> 5, bit = tid & 31;
    uint32_t cmv[8][MWT];                                  // (the words of columns outside the tile / the layer are zero)
    if (MWT == 1) {
        const uint4 lo = ((const uint4 *)cmn)[0], hi = ((const uint4 *)cmn)[1];
        cmv[0][0] = lo.x; cmv[1][0] = lo.y; cmv[2][0] = lo.z; cmv[3][0] = lo.w;
        cmv[4][0] = hi.x; cmv[5][0] = hi.y; cmv[6][0] = hi.z; cmv[7][0] = hi.w;
    } else if (mw == MWT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 v = ((const uint4 *)cmn)[q];
            cmv[q][0] = v.x; cmv[q][1 % MWT] = v.y; cmv[q][2 % MWT] = v.z; cmv[q][3 % MWT] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int w2 = 0; w2 < MWT; ++w2) cmv[q][w2] = w2 < mw ? cmn[q * mw + w2] : 0u;
    }
    uint32_t colbits = 0, any = 0;
    int rank = 0, total = 0;
    bool in = false;
#pragma unroll
    for (int w2 = 0; w2 < MWT; ++w2) {
        uint32_t un = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t v = cmv[q][w2];
            un |= v;
            if (v) any |= 1u << q;
            if (w2 == w && ((v >> bit) & 1u)) colbits |= 1u << q;
        }
        total += __popc(un);
        if (w2 < w) rank += __popc(un);
        else if (w2 == w) { rank += __popc(un & ((1u << bit) - 1u)); in = (un >> bit) & 1u; }
    }
    if (in) {
        ((uint8_t *)ul)[rank] = (uint8_t)tid;
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = ((colbits >> q) & 1u) ? 1.f : 0.f;
        fac[2 * rank] = make_float4(f[0], f[1], f[2], f[3]);
        fac[2 * rank + 1] = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (tid == 0) { ul[8 * mw] = (uint32_t)total; ul[8 * mw + 1] = any; }
}

// MSTDP update of one step on the LDS weight tile (learning.py:1504-1574 with the eligibility factored as in
// snn_mstdp_step): w += nu0 * sum_b reward[b] * (p_plus[b,i] * s_tgt[b,j] + s_src[b,i] * p_minus[b,j]), decay, clamp,
// all four factors being those of the PREVIOUS step.  Samples in which neither side spiked contribute +0.0 and are
// skipped; p_plus >= +0 (a_plus >= 0, host check) so a silent target contributes exactly +0.0 without loading it.
template <class SUM, int MWT>
__device__ __forceinline__ void two_mstdp(const TwoCtx &c, float *wt, const uint16_t *ar, const uint32_t *am,
                                          const uint32_t *ab, const float *pml, const float *zl, const uint32_t *cm,
                                          const float *rvl, const float *__restrict__ pp, const uint8_t *__restrict__ sbytes,
                                          int nact, bool full, int c0, int tid, int cwl, int Emain) {
    const int B = c.B, Nin = c.Nin, N = c.N, CW = c.CW, mw = MWT == 1 ? 1 : c.MW;
    const uint32_t zeros[4] = {0u, 0u, 0u, 0u};
    auto any_of = [&](const uint32_t *m) -> bool { uint32_t o = 0; for (int w = 0; w < mw; ++w) o |= m[w]; return o != 0; };
    // m: samples in which row i spiked (nullptr-safe: `zeros`), cq: samples in which column q spiked
    auto elem = [&](int i, int q, const uint32_t *m, const uint32_t *cq, float w) -> float {
        SUM acc; acc.init(i * N + c0 + q >= Emain);
        const bool cany = any_of(cq);
        for (int wd = 0; wd < mw; ++wd) {                 // ascending sample index = the reference's batch-sum order
            const uint32_t mr = m[wd], mc = cq[wd];
            uint32_t mm = mr | mc;
            if (!cany && !sbytes) {   // no target spike in this column, 0/1 source spikes: the term is zl[b][q], staged by the caller
                while (mm) { const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1; acc.add(b, zl[b * 8 + q], B); }
            } else
            while (mm) {
                const int bb = __ffs(mm) - 1, b = wd * 32 + bb; mm &= mm - 1;
                const float e1 = ((mc >> bb) & 1u) ? pp[b * Nin + i] * 1.0f : 0.0f;                        // p_plus (x) s_tgt
                const float sv = ((mr >> bb) & 1u) ? (sbytes ? (float)sbytes[b * Nin + i] : 1.0f) : 0.0f;
                const float e2 = sv * pml[b * 8 + q];                                                     // s_src (x) p_minus
                acc.add(b, rvl[b] * (e1 + e2), B);
            }
        }
        const float u = acc.finish(B);
        w = w + c.nu0 * u;                                // learning.py:1561
        w = w * c.wdecay;
        if (c.has_min && w < c.wmin) w = c.wmin;
        if (c.has_max && w > c.wmax) w = c.wmax;
        return w;
    };
    // ---- pass 1: rows whose source spiked x own columns
    for (int item = tid; item < (nact << cwl); item += NT) {
        const int kq = item >> cwl, q = item & (CW - 1);
        if (c0 + q >= N) continue;
        const int i = (int)ar[kq];
        wt[i * CW + q] = elem(i, q, am + kq * mw, cm + q * mw, wt[i * CW + q]);
    }
    // ---- pass 2: the other rows: the columns whose target spiked (every column when `full`)
    uint32_t todo = 0;
    for (int q = 0; q < CW; ++q) if (c0 + q < N && (full || any_of(cm + q * mw))) todo |= 1u << q;
    if (!todo) return;
    for (int i = tid; i < Nin; i += NT) {
        if ((ab[i >> 5] >> (i & 31)) & 1u) continue;
        uint32_t td = todo;
        while (td) {
            const int q = __ffs(td) - 1; td &= td - 1;
            wt[i * CW + q] = elem(i, q, zeros, cm + q * mw, wt[i * CW + q]);
        }
    }
}

// MSTDP update of a step in which no column of the tile had a target spike (the usual case: learning.py:1504-1574 then
// reduces to w[i,j] += nu0 * sum_b s_src[b,i] * reward[b] * p_minus[b,j], terms staged in zl): one thread per ACTIVE
// source row, all columns of the tile in registers, the row's samples walked in ascending order with the block partial
// sums of two_stdp_rowmajor.  Rows without a source spike are not touched (the caller sends steps that must touch
// every element -- first step of a run, weight decay -- through two_mstdp).
template <int MWT>
__device__ __forceinline__ void two_mstdp_rows(const TwoCtx &c, float *wt, const uint16_t *ar, const uint32_t *am,
                                               const float *zl, int nact, int c0, int tid) {
    const int N = c.N, CW = c.CW, mw = MWT == 1 ? 1 : c.MW;
    for (int kq = tid; kq < nact; kq += NT) {
        const int i = (int)ar[kq];
        float a0[8], a1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a0[q] = a1[q] = 0.f;
        uint32_t m[MWT];
#pragma unroll
        for (int wd = 0; wd < MWT; ++wd) m[wd] = wd < mw ? am[kq * mw + wd] : 0u;
        int cblk = 0;
#pragma unroll
        for (int wd = 0; wd < MWT; ++wd) {
            uint32_t mm = m[wd];
            while (mm) {
                const int b = wd * 32 + __ffs(mm) - 1; mm &= mm - 1;
                const float4 lo = *(const float4 *)(zl + b * 8), hi = *(const float4 *)(zl + b * 8 + 4);
                const float zv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const float same = (b >> 4) == cblk ? 1.f : 0.f, diff = 1.f - same;   // (0/1 factors: exact products, see two_stdp_rowmajor)
                cblk = b >> 4;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a1[q] = __builtin_fmaf(a0[q], diff, a1[q]);
                    a0[q] = __builtin_fmaf(a0[q], same, zv[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= CW || c0 + q >= N) break;
            const float u = ((a0[q] + a1[q]) + 0.f) + 0.0f;
            float w = wt[i * CW + q];
            w = w + c.nu0 * u;                                // learning.py:1561
            w = w * c.wdecay;
            if (c.has_min && w < c.wmin) w = c.wmin;
            if (c.has_max && w > c.wmax) w = c.wmax;
            wt[i * CW + q] = w;
        }
    }
}

// MSTDP update of a step in which columns of the tile DID spike (bursts: then every row of those columns changes, and
// each element sums up to B terms).  One thread per source row, the tile's columns in registers, the samples walked in
// ascending order by the whole workgroup together: p_plus[b, :] is loaded once per sample (coalesced, eight samples in
// flight) and serves all columns.  A sample that contributes nothing to an element adds reward * (+0 + -+0) = -+0 to its
// partial sum, which (partials start at +0.0) leaves it unchanged: the reference's dense batch sum adds those zeros too.
//   term(b) = reward[b] * (p_plus[b,i] * s_tgt[b,j] + s_src[b,i] * p_minus[b,j]);  the 0/1 spike factors enter through
//   exact products (fma(x, 1 or 0, y) rounds once, like the sum it stands for).
// Rows with a source spike come from the digest's list (pass A), the others from the row bitmap (pass B, no p_minus part).
template <int MWT, int CWT>
__device__ __forceinline__ void two_mstdp_burst(const TwoCtx &c, float *wt, const uint16_t *ar, const uint32_t *am,
                                                const uint32_t *ab, const float *pml, const uint32_t *cm, const float *rvl,
                                                const float *__restrict__ pp, int nact, int c0, int tid) {
    const int B = c.B, Nin = c.Nin, N = c.N, mw = MWT == 1 ? 1 : c.MW;      // (CWT = c.CW, known at compile time here)
    uint32_t cu[MWT] = {0}, cmr[CWT][MWT];                 // column spike masks and their union (scalar registers)
    uint32_t postcols = 0;
#pragma unroll
    for (int q = 0; q < CWT; ++q) {
        uint32_t o = 0;
#pragma unroll
        for (int w = 0; w < MWT; ++w) {
            const bool on = c0 + q < N && w < mw;
            const uint32_t v = on ? (uint32_t)__builtin_amdgcn_readfirstlane(cm[q * mw + w]) : 0u;
            cmr[q][w] = v; cu[w] |= v; o |= v;
        }
        if (o) postcols |= 1u << q;
    }
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc((void *)pp, 0, B * Nin * 4, 0x00020000);
    auto word = [&](const uint32_t (&a)[MWT], int wd) -> uint32_t {   // a[wd] without indexing registers dynamically
        uint32_t v = a[0];
#pragma unroll
        for (int w = 1; w < MWT; ++w) v = wd == w ? a[w] : v;
        return v;
    };
    // The rows of this thread, one after the other: first its share of the digest's list of rows with a source spike,
    // then its share of all rows (those with the bitmap bit clear).  Every thread makes the same number of turns.
    const int SA = (nact + NT - 1) / NT, S = SA + (Nin + NT - 1) / NT, NU = (B + 15) >> 4;
    auto rowinfo = [&](int s_, int &i, int &kq, bool &live, bool &hasrow) {
        if (s_ < SA) { kq = tid + s_ * NT; live = kq < nact; i = live ? (int)ar[kq] : 0; hasrow = true; }
        else { kq = 0; i = tid + (s_ - SA) * NT; live = i < Nin && !((ab[i >> 5] >> (i & 31)) & 1u); hasrow = false; if (!live) i = 0; }
    };
    // p_plus[b0 + k, i].  Loaded unconditionally (a value that is not needed meets a zero factor below; rows past the
    // batch repeat the last one): a load under a branch makes the compiler wait for ALL loads in flight at the join.
    auto issue = [&](int i, int b0, int k) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(slab, i * 4, min(b0 + k, B - 1) * Nin * 4, 0));
    };
    int i, kq; bool live, hasrow;
    rowinfo(0, i, kq, live, hasrow);
    float x[16];                                           // one unit = 16 samples = one block of the batch sum; a slot is refilled
#pragma unroll                                             // for the NEXT unit (of this row, or the first of the next row) once used
    for (int k = 0; k < 16; ++k) x[k] = issue(i, 0, k);
    for (int s_ = 0; s_ < S; ++s_) {
        int i2 = 0, kq2 = 0; bool live2 = false, hasrow2 = false;
        if (s_ + 1 < S) rowinfo(s_ + 1, i2, kq2, live2, hasrow2);
        uint32_t m[MWT];
#pragma unroll
        for (int wd = 0; wd < MWT; ++wd) m[wd] = (hasrow && live && wd < mw) ? am[kq * mw + wd] : 0u;
        float w[CWT], a0[CWT], a1[CWT];
#pragma unroll
        for (int q = 0; q < CWT; ++q) { w[q] = live ? wt[i * CWT + q] : 0.f; a0[q] = a1[q] = 0.f; }
        for (int u = 0; u < NU; ++u) {
            const int b0 = u * 16, wd = b0 >> 5, sh = b0 & 31;
            const bool lastu = u + 1 == NU;
            const int nb0 = lastu ? 0 : b0 + 16, ni = lastu ? i2 : i;
            const uint32_t cbits = (word(cu, wd) >> sh) & 0xFFFFu;
            const uint32_t rbits = (word(m, wd) >> sh) & 0xFFFFu;                      // (per lane)
            uint32_t cw[CWT];
#pragma unroll
            for (int q = 0; q < CWT; ++q) cw[q] = (word(cmr[q], wd) >> sh) & 0xFFFFu;
#pragma unroll
            for (int q = 0; q < CWT; ++q) { a1[q] = a1[q] + a0[q]; a0[q] = 0.f; }      // the previous block is complete (+0.0 at u == 0)
            const bool anyrow = __ballot(rbits != 0) != 0;                             // (uniform)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float xk = x[k];
                x[k] = issue(ni, nb0, k);
                const bool cb = (cbits >> k) & 1u;
                if (!cb && !anyrow) continue;                                          // nobody's sample (uniform)
                const int b = min(b0 + k, B - 1);                                      // (samples past the batch: all factors are zero)
                const float rowf = ((rbits >> k) & 1u) ? 1.0f : 0.0f;
                const float rv = rvl[b];
#pragma unroll
                for (int q = 0; q < CWT; ++q) {
                    const float fq = ((cw[q] >> k) & 1u) ? 1.0f : 0.0f;
                    const float e = __builtin_fmaf(xk, fq, rowf * pml[b * 8 + q]);     // p_plus (x) s_tgt + s_src (x) p_minus
                    a0[q] = a0[q] + rv * e;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < CWT; ++q) {
                if (c0 + q >= N) break;
                if (!hasrow && !((postcols >> q) & 1u)) continue;     // (row silent, column silent: the element is not touched)
                const float uu = ((a0[q] + a1[q]) + 0.f) + 0.0f;
                float v = w[q] + c.nu0 * uu;                          // learning.py:1561
                v = v * c.wdecay;
                if (c.has_min && v < c.wmin) v = c.wmin;
                if (c.has_max && v > c.wmax) v = c.wmax;
                wt[i * CWT + q] = v;
            }
        }
        i = i2; kq = kq2; live = live2; hasrow = hasrow2;
    }
}

// CASC: MulticompartmentConnection (ATen cascade order) vs dense Connection (ascending sequential order);
// RULE: the connection's learning rule.  Compile-time so that each variant carries only its own code (and registers).
// MWT: 1 = batch <= 32 (one sample-mask word, compiled without the word loops), 4 = up to 128 samples.
template <bool CASC, int RULE, int MWT>
__global__ __launch_bounds__(NT) void k_two_run(const TwoCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, N = c.N, NinW = c.NinW, CW = c.CW, mw = MWT == 1 ? 1 : c.MW, BC = 32 * mw, CMS = 8 * mw;
    size_t off = 0;
    float *wt = (float *)(smem + off); off += (size_t)Nin * CW * 4;                       // own weight slice [Nin][CW]
    int *meta = (int *)(smem + off); off += META * 4;
    uint16_t *ent = (uint16_t *)(smem + off); off += ((size_t)c.LCAP * 2 + 15) & ~(size_t)15;
    uint32_t *am = (uint32_t *)(smem + off); off += (size_t)Nin * mw * 4;                 // [active rows][MW] sample masks
    uint16_t *ar = (uint16_t *)(smem + off); off += ((size_t)Nin * 2 + 15) & ~(size_t)15;
    uint32_t *ab = (uint32_t *)(smem + off); off += ((size_t)NinW * 4 + 15) & ~(size_t)15;
    uint16_t *ridx = (uint16_t *)(smem + off); off += c.rowmajor ? (((size_t)Nin * 2 + 15) & ~(size_t)15) : 0;   // source row -> index in ar / am (row-major PostPre)
    float *xnu0 = (float *)(smem + off); off += (size_t)BC * 8 * 4;                       // [B][CW] x_tgt * nu0
    uint32_t *colmask = (uint32_t *)(smem + off); off += (size_t)2 * CMS * 4;             // [2][8][MW]: samples whose neuron (column q) spiked
    float *rvl = (float *)(smem + off); off += (size_t)BC * 4;                            // MSTDP: reward per sample
    float *zl = (float *)(smem + off); off += (size_t)BC * 8 * 4;                         // MSTDP: reward * p_minus per (sample, column)
    uint32_t *ul = (uint32_t *)rvl; float4 *fac = (float4 *)zl;                           // PostPre, row-major form: see two_union_list (the MSTDP arrays are free)
    float *prod = (float *)(smem + off); off += c.prodw ? (size_t)c.prodw * 4 : 0;        // dense dot: staged products
    int *szt = (int *)(smem + off); off += (size_t)(c.T + 2) * 8;                         // (events, active rows) of every digest entry
    float *xsl = (float *)(smem + off); off += c.use_xsl ? (size_t)Nin * CW * 4 : 0;      // source trace of the first spiking sample of each column

    const int tid = threadIdx.x;
    const int g = blockIdx.x, c0 = g * CW;
    const int jj = tid & (CW - 1), bl = tid >> (31 - __clz(CW)), j = c0 + jj;
    const bool mine = tid < B * CW && j < N;
    const int kst = bl * N + j;
    const bool tailcol = CASC && j >= (N / 32) * 32;
    const int Etot = Nin * N, Emain = (Etot / 32) * 32;
    const int cwl = 31 - __clz(CW);                      // CW is a power of two
    constexpr bool kOuter = RULE == SNN_RULE_POSTPRE || RULE == SNN_RULE_HEBBIAN || RULE == SNN_RULE_WDPOSTPRE;   // the rules of two_stdp
    const bool do_stdp = kOuter && c.learning;
    const bool do_mstdp = RULE == SNN_RULE_MSTDP && c.learning;

    // ---- prologue: own weight slice and membrane state
    for (int k = tid; k < Nin * CW; k += NT) { const int i = k / CW, q = k - i * CW; wt[k] = c0 + q < N ? c.W[i * N + c0 + q] : 0.f; }
    float v = 0.f, rc = 0.f, xy = 0.f, bias = 0.f, pm = 0.f;
    uint8_t sp_prev = 0;
    if (mine) {
        v = c.vY[kst]; rc = c.rY[kst];
        if (c.pY.traces) xy = c.xY[kst];
        if (c.bias) bias = c.bias[j];
        if (do_mstdp) pm = c.p_minus[kst];
    }
    if (tid < 2 * CMS) colmask[tid] = 0;
    if (tid < 8 * mw + 2) ul[tid] = 0;
    if (tid < BC) rvl[tid] = (do_mstdp && tid < B) ? (c.reward_vec ? c.reward_vec[tid] : c.reward) : 0.f;
    if (do_mstdp) {      // the target spikes the rule remembers from its last update: "previous step" of iteration 0
        __syncthreads();
        if (mine && c.s_tgt_prev[kst]) atomicOr(&colmask[CMS + jj * mw + (bl >> 5)], 1u << (bl & 31));
    }
    // digest words copied into LDS each step: [meta | entries | row masks | active rows | row bitmap]
    const int region4 = NinW;
    for (int k = tid; k < (c.T + 2) * 2; k += NT)
        szt[k] = ((k >> 1) <= c.T || do_mstdp) ? (int)c.dig[(size_t)(k >> 1) * c.DW + (k & 1)] : 0;
    __syncthreads();
    uint32_t r_dg[PF];
    float x_pf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x_pf[q] = 0.f;
    auto issue = [&](int e) {     // loads of digest entry e; lengths from the LDS size table, so no dependent global read
        const uint32_t *D = c.dig + (size_t)e * c.DW;
        const int tot = min(szt[2 * e], c.LCAP), nact = szt[2 * e + 1];
        const int n1 = (tot + 1) / 2, n2 = nact * mw, n3 = (nact + 1) / 2, n4 = region4;
        const int nw = META + n1 + n2 + n3 + n4;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u * NT >= nw) break;
            int k = tid + u * NT;
            uint32_t val = 0;
            if (k < META) val = D[k];
            else if ((k -= META) < n1) val = D[c.o_ent + k];
            else if ((k -= n1) < n2) val = D[c.o_am + k];
            else if ((k -= n2) < n3) val = D[c.o_ar + k];
            else if ((k -= n3) < n4) val = D[c.o_ab + k];
            r_dg[u] = val;
        }
    };
    issue(0);

    for (int t = 0; t <= c.T; ++t) {
        TMARK(0);
        // ---- stage the digest of step t-1 (entry t) from the prefetch registers
        // sizes first (they sit in the first META words = thread tid < META, u = 0)
        {
            const int tot = min(szt[2 * t], c.LCAP), nact = szt[2 * t + 1];
            const int n1 = (tot + 1) / 2, n2 = nact * mw, n3 = (nact + 1) / 2, n4 = region4;
            const int nw = META + n1 + n2 + n3 + n4;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (u * NT >= nw) break;
                int k = tid + u * NT;
                if (k < META) meta[k] = (int)r_dg[u];
                else if ((k -= META) < n1) ((uint32_t *)ent)[k] = r_dg[u];
                else if ((k -= n1) < n2) am[k] = r_dg[u];
                else if ((k -= n2) < n3) {
                    ((uint32_t *)ar)[k] = r_dg[u];
                    if (c.rowmajor) {
                        ridx[r_dg[u] & 0xFFFFu] = (uint16_t)(2 * k);           // (two u16 rows per word; a stale second half
                        if (2 * k + 1 < nact) ridx[r_dg[u] >> 16] = (uint16_t)(2 * k + 1);   //  past nact is not an active row)
                    }
                }
                else if ((k -= n3) < n4) ab[k] = r_dg[u];
            }
            // traces prefetched at the end of the previous iteration, and the spike masks this iteration will fill
            if (c.use_xsl && t >= 1 && tid < Nin) {
                const uint32_t *cmp = colmask + ((t + 1) & 1) * CMS;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    uint32_t anyq = 0;
                    if (q < CW) for (int w = 0; w < mw; ++w) anyq |= cmp[q * mw + w];
                    if (anyq) xsl[q * Nin + tid] = x_pf[q];
                }
            }
            if (tid < CMS) colmask[(t & 1) * CMS + tid] = 0;
        }
        lds_barrier();
        TMARK(1);
        if (t + 1 <= c.T) issue(t + 1);                  // next step's digest: in flight behind this step's work
        TMARK(2);
        const int tot = meta[0], nact = meta[1], flags = meta[2];
        const bool overflow = (flags & 2) != 0;          // more events than the LDS list holds: walk bit words from global
        const uint8_t *sbytes = (flags & 1) ? ((t == 0) ? c.sX0 : c.in + (size_t)(t - 1) * B * Nin) : nullptr;
        const uint32_t *cm = colmask + ((t + 1) & 1) * CMS;   // spikes of step t-1 (written in iteration t-1)

        // ================================================== phase A: PostPre of step t-1 on the LDS weight tile
        if (t >= 1 && do_stdp) {
            const bool full = (t == 1) || c.wdecay != 1.0f;   // first update of a run (or a real decay) touches every element
            const float *xs = c.xall + (size_t)(t - 1) * B * Nin;
            if constexpr (RULE == SNN_RULE_HEBBIAN || RULE == SNN_RULE_WDPOSTPRE) {
                if (Etot != Emain) two_stdp<OuterSum, MWT, RULE>(c, wt, ar, am, ab, xnu0, cm, xs, c.use_xsl ? xsl : nullptr, sbytes, nact, full, c0, tid, cwl, Emain);
                else if (sbytes || c.rowmajor == 0) two_stdp<CascT, MWT, RULE>(c, wt, ar, am, ab, xnu0, cm, xs, c.use_xsl ? xsl : nullptr, sbytes, nact, full, c0, tid, cwl, Emain);
                else two_stdp_rowmajor<MWT, RULE>(c, wt, am, ab, ridx, xnu0, ul, fac, xs, full, c0, tid);
            } else {
            if (Etot != Emain) two_stdp<OuterSum, MWT>(c, wt, ar, am, ab, xnu0, cm, xs, c.use_xsl ? xsl : nullptr, sbytes, nact, full, c0, tid, cwl, Emain);
            else if (sbytes || c.rowmajor == 0) two_stdp<CascT, MWT>(c, wt, ar, am, ab, xnu0, cm, xs, c.use_xsl ? xsl : nullptr, sbytes, nact, full, c0, tid, cwl, Emain);
            else two_stdp_rowmajor<MWT>(c, wt, am, ab, ridx, xnu0, ul, fac, xs, full, c0, tid);
            }
        }
        WMARK();
        TMARK(3);
        lds_barrier();
        TMARK(4);
        if (t == c.T) break;

        // ================================================== phase B: step t
        uint32_t *cmn = colmask + (t & 1) * CMS;
        uint8_t sp = 0;
        float cur = 0.f;
        // dense Connection with long event lists (wide inputs): one thread per (sample, column) would chase ~hundreds of
        // dependent LDS gathers.  Instead every wave takes (sample, column) pairs in turns: its 64 lanes gather 64
        // products at once, then the sum walks them in ascending order through readlane -- the same sequential f32
        // adds as the reference (topology.py:332-346), without a memory access in the dependency chain.
        const bool wave_dot = !CASC && !overflow && c.prodw > 0 && tot >= 48 * B;
        float r_dot = 0.f;
        if (!CASC && wave_dot) {
            // Phase P: all threads stage the products w[i,j] * s[b,i] of a chunk of every (sample, column) pair's event
            // list in LDS (zero padded: x + 0.0f == x).  Phase S: the pair's own tile thread (lanes of ONE wave, in
            // lockstep) adds its chunk in order -- the chain now holds nothing but the adds.
            const int npairs = B << cwl;
            const int CH = ((c.prodw / npairs) & ~3) - 4, ST = CH + 4;     // chunk length, row stride (banks staggered)
            int maxn = 0;
            for (int b = 0; b < B; ++b) maxn = max(maxn, meta[5 + b] - meta[4 + b]);
            for (int e0 = 0; e0 < maxn; e0 += CH) {
                const int lim = min(CH, (maxn - e0 + 3) & ~3);
                const float inv_lim = 1.0f / (float)lim;
                for (int it = tid; it < npairs * lim; it += NT) {
                    int p = (int)((float)it * inv_lim);          // it / lim without the integer division (it < 2^24: one step of correction)
                    if (p * lim > it) --p; else if ((p + 1) * lim <= it) ++p;
                    const int e = it - p * lim;
                    const int pb = p >> cwl, pq = p & (CW - 1);
                    const int k = meta[4 + pb] + e0 + e;
                    float term = 0.f;
                    if (k < meta[5 + pb]) { const int i = (int)ent[k]; term = wt[i * CW + pq] * (sbytes ? (float)sbytes[pb * Nin + i] : 1.0f); }
                    prod[p * ST + e] = term;
                }
                lds_barrier();
                if (tid < npairs) {
                    const float4 *row = (const float4 *)(prod + tid * ST);
                    for (int e = 0; e < lim; e += 16) {      // four LDS reads in flight per 16 dependent adds (lim is a multiple of 4)
                        float4 x[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[k] = e + 4 * k < lim ? row[(e >> 2) + k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (e + 4 * k >= lim) break;
                            r_dot += x[k].x; r_dot += x[k].y; r_dot += x[k].z; r_dot += x[k].w;
                        }
                    }
                }
                if (e0 + CH < maxn) lds_barrier();
            }
        }
        if (mine) {
            const uint8_t *xb = sbytes ? sbytes + bl * Nin : nullptr;
            float r;
            if (!CASC && wave_dot) {
                r = r_dot;
            } else if (!overflow) {
                const int n0 = meta[4 + bl], n1 = meta[5 + bl];
                if (!CASC) r = list_dot<SeqN>(wt, CW, jj, ent, n0, n1, xb, Nin);
                else if (tailcol) r = list_dot<RowSumN>(wt, CW, jj, ent, n0, n1, xb, Nin);
                else if (Nin < 4096) r = list_dot<CascadeFlat>(wt, CW, jj, ent, n0, n1, xb, Nin);
                else r = list_dot<CascadeN>(wt, CW, jj, ent, n0, n1, xb, Nin);
            } else {             // walk the bit words of the sample from the global digest (rare)
                const uint32_t *xw = c.dig + (size_t)t * c.DW + c.o_xw + bl * NinW;
                OuterSum a; a.init(tailcol);
                float seq = 0.f;
                for (int w = 0; w < NinW; ++w) {
                    uint32_t m = xw[w];
                    while (m) {
                        const int i = w * 32 + __ffs(m) - 1; m &= m - 1;
                        const float term = wt[i * CW + jj] * (xb ? (float)xb[i] : 1.0f);
                        if (CASC) a.add(i, term, Nin); else seq += term;
                    }
                }
                r = CASC ? a.finish(Nin) : seq;
            }
            if (c.bias) r = r + bias;                     // topology.py:345
            cur = 0.0f + r;                               // network.py:240-248
            if (rc > 0.f) cur = 0.f;                      // nodes.py:511
            sp = lif_update(v, rc, cur, c.pY);
            if (c.pY.traces) xy = trace_next(xy, sp, c.pY.trace_decay, c.pY.trace_scale, c.pY.traces_additive);
            sp_prev = sp;
        }
        TMARK(5);
        if (mine) {
            if (do_mstdp) {
                xnu0[bl * 8 + jj] = pm;                   // p_minus of the PREVIOUS step: a factor of this step's update
                zl[bl * 8 + jj] = rvl[bl] * (0.0f + 1.0f * pm);   // its whole term when only the source spiked (value 1)
                const float p = pm * c.d_minus;           // learning.py:1566-1567
                pm = p + c.a_minus * (float)sp;
            } else
            xnu0[bl * 8 + jj] = (RULE == SNN_RULE_POSTPRE) ? xy * c.nu0 : xy;   // target_x * nu[0] (Hebbian / WeightDependentPostPre: nu applied to the batch sum)
            if (sp) atomicOr(&cmn[jj * mw + (bl >> 5)], 1u << (bl & 31));
            if (c.rasY) c.rasY[(size_t)t * B * N + kst] = sp;
            if (c.rasVY) c.rasVY[(size_t)t * B * N + kst] = v;
        }
        lds_barrier();                                   // this step's spike masks are final
        if (kOuter && do_stdp && c.rowmajor) two_union_list<MWT>(c, cmn, ul, fac, mw, c0, tid);   // (read after the next barrier)
        if (c.use_xsl && do_stdp && tid < Nin) {         // source traces the next PostPre needs: in flight across the loop edge
            const float *xs = c.xall + (size_t)t * B * Nin;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int fb = -1;
                if (q < CW) for (int w = 0; w < mw && fb < 0; ++w) if (cmn[q * mw + w]) fb = w * 32 + __ffs(cmn[q * mw + w]) - 1;
                x_pf[q] = fb >= 0 ? xs[fb * Nin + tid] : 0.f;
            }
        }
        if (do_mstdp) {
            // ---- MSTDP update of step t (applied before step t+1 propagates): factors of step t-1 = this iteration's
            //      source digest, the spike masks of iteration t-1, p_minus staged above, p_plus after step t-1
            const float *pp = c.pall + (size_t)t * B * Nin;
            const uint8_t *sb = sbytes;
            int na = nact;
            if (t == 0) {
                // the very first update pairs with what the rule remembers from before this run, not with the input
                // layer's entry spikes: digest entry T+1 replaces the row tables (the event list is not needed)
                const uint32_t *D = c.dig + (size_t)(c.T + 1) * c.DW;
                na = szt[2 * (c.T + 1) + 1];
                for (int k = tid; k < na * mw; k += NT) am[k] = D[c.o_am + k];
                for (int k = tid; k < (na + 1) / 2; k += NT) ((uint32_t *)ar)[k] = D[c.o_ar + k];
                for (int k = tid; k < NinW; k += NT) ab[k] = D[c.o_ab + k];
                sb = (D[2] & 1u) ? c.s_src_prev : nullptr;
                __syncthreads();
            }
            TMARK(7);
            const bool full = (t == 0) || c.wdecay != 1.0f;
            uint32_t anycol = 0;                             // did any column of the tile spike in the step this update pairs with?
            for (int k = 0; k < CMS; ++k) anycol |= cm[k];
            if (c.dbg && blockIdx.x == 0 && tid == 0 && anycol) c.dbg[(size_t)8 * 4096 + (size_t)16 * 4096 - 1 - t] = 1;
            if (Etot == Emain && !full && !sb && c.mstdp_rows && !__builtin_amdgcn_readfirstlane(anycol)) two_mstdp_rows<MWT>(c, wt, ar, am, zl, na, c0, tid);
            else if (Etot == Emain && !full && !sb && c.mstdp_rows && CW >= 2) {
                if (CW == 8) two_mstdp_burst<MWT, 8>(c, wt, ar, am, ab, xnu0, cm, rvl, pp, na, c0, tid);
                else if (CW == 4) two_mstdp_burst<MWT, 4>(c, wt, ar, am, ab, xnu0, cm, rvl, pp, na, c0, tid);
                else two_mstdp_burst<MWT, 2>(c, wt, ar, am, ab, xnu0, cm, rvl, pp, na, c0, tid);
            }
            else
            if (Etot != Emain) two_mstdp<OuterSum, MWT>(c, wt, ar, am, ab, xnu0, zl, cm, rvl, pp, sb, na, full, c0, tid, cwl, Emain);
            else two_mstdp<CascT, MWT>(c, wt, ar, am, ab, xnu0, zl, cm, rvl, pp, sb, na, full, c0, tid, cwl, Emain);
            lds_barrier();                               // tile and row tables are free for the next iteration
        }
        (void)tot;
        TMARK(6);
    }

    // ---- epilogue: write the weight slice and the membrane state back
    __syncthreads();
    if (c.has_norm) {
        // network.py:464-465 + topology.py:383-392 / topology_features.py:250-266: column sums (of |w| for a dense
        // Connection) in ATen's sum(dim=0) order over the LDS tile, zero -> 1, W *= norm * (1 / colsum); the same
        // arithmetic as k_colsum / k_scale_cols
        float *bsum = (float *)am;                     // [nfull][CW] 16-row block sums (the row tables are free now)
        float *sc = xnu0;                              // [CW] column scales
        const int nfull = Nin >> 4;
        if (c0 < (N / 32) * 32) {                      // multi_row_sum columns (a tile never straddles the class boundary)
            for (int item = tid; item < nfull * CW; item += NT) {
                const int blk = item / CW, q = item - blk * CW;
                float a0 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float w = wt[(blk * 16 + k) * CW + q]; a0 += c.norm_abs ? fabsf(w) : w; }
                bsum[item] = a0;
            }
            __syncthreads();
            if (tid < CW) {
                float a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int blk = 0; blk < nfull; ++blk) {
                    a1 += bsum[blk * CW + tid];
                    const int m = blk + 1;
                    if ((m & 15) == 0) { a2 += a1; a1 = 0.f; if ((m & 255) == 0) { a3 += a2; a2 = 0.f; } }
                }
                float a0 = 0.f;
                for (int i = nfull * 16; i < Nin; ++i) { const float w = wt[i * CW + tid]; a0 += c.norm_abs ? fabsf(w) : w; }
                float cs = ((a0 + a1) + a2) + a3;
                if (cs == 0.f) cs = 1.0f;
                sc[tid] = (1.0f / cs) * c.norm;
            }
        } else if (tid < CW * 4) {                     // row_sum columns: four interleaved lanes per column
            const int q = tid >> 2, s4 = tid & 3, n4 = Nin >> 2, nf4 = n4 >> 4;
            Cascade cc; cc.init();
            for (int p_ = 0; p_ < n4; ++p_) { const float w = wt[(4 * p_ + s4) * CW + q]; cc.add(p_, c.norm_abs ? fabsf(w) : w, nf4); }
            float lsum = cc.finish(nf4);
            if (s4 == 0)
                for (int i = n4 * 4; i < Nin; ++i) { const float w = wt[i * CW + q]; lsum += c.norm_abs ? fabsf(w) : w; }
            const float l1 = __shfl_down(lsum, 1, 4), l2 = __shfl_down(lsum, 2, 4), l3 = __shfl_down(lsum, 3, 4);
            float cs = ((lsum + l1) + l2) + l3;
            if (cs == 0.f) cs = 1.0f;
            if (s4 == 0) sc[q] = (1.0f / cs) * c.norm;
        }
        __syncthreads();
        for (int k = tid; k < Nin * CW; k += NT) { const int i = k / CW, q = k - i * CW; if (c0 + q < N) c.W[i * N + c0 + q] = wt[k] * sc[q]; }
    } else
    for (int k = tid; k < Nin * CW; k += NT) { const int i = k / CW, q = k - i * CW; if (c0 + q < N) c.W[i * N + c0 + q] = wt[k]; }
    if (mine) {
        c.vY[kst] = v; c.rY[kst] = rc; c.sY[kst] = sp_prev;
        if (c.pY.traces) c.xY[kst] = xy;
        if (do_mstdp) { c.p_minus[kst] = pm; c.s_tgt_prev[kst] = sp_prev; }
    }
}

int digest_layout(TwoCtx &c) {
    c.o_ent = META;
    c.o_am = c.o_ent + c.LCAP / 2;
    c.o_ar = c.o_am + c.Nin * c.MW;
    c.o_ab = c.o_ar + (c.Nin + 1) / 2;
    c.o_xw = c.o_ab + c.NinW;
    return (c.o_xw + c.B * c.NinW + 3) & ~3;
}

size_t run_lds(const TwoCtx &c) {
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return (size_t)c.Nin * c.CW * 4 + META * 4 + al((size_t)c.LCAP * 2) + (size_t)c.Nin * 4 + al((size_t)c.Nin * 2) +
           (size_t)c.Nin * (c.MW - 1) * 4 +
           al((size_t)c.NinW * 4) + (c.rowmajor ? al((size_t)c.Nin * 2) : 0) + (size_t)c.BC * 8 * 4 + (size_t)16 * c.MW * 4 + (size_t)c.BC * 4 + (size_t)c.BC * 8 * 4 + (size_t)c.prodw * 4 +
           (size_t)(c.T + 2) * 8 +
           (c.use_xsl ? (size_t)c.Nin * c.CW * 4 : 0);
}

bool plan(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, TwoCtx &c) {
    if (nL != 2 || nC != 1) return false;
    if (L[0].kind != SNN_LAYER_INPUT || L[1].kind != SNN_LAYER_LIF) return false;
    if (C[0].src != 0 || C[0].dst != 1) return false;
    if (C[0].kind != SNN_CONN_MCC && C[0].kind != SNN_CONN_DENSE) return false;
    const bool outer2 = C[0].rule == SNN_RULE_HEBBIAN || C[0].rule == SNN_RULE_WDPOSTPRE;     // dense Connection only
    if (C[0].rule != SNN_RULE_NONE && C[0].rule != SNN_RULE_POSTPRE && C[0].rule != SNN_RULE_MSTDP && !(outer2 && C[0].kind == SNN_CONN_DENSE)) return false;
    if (C[0].rule == SNN_RULE_WDPOSTPRE && !(C[0].has_min && C[0].has_max)) return false;
    if (C[0].rule == SNN_RULE_MSTDP && (!C[0].p_plus || !C[0].p_minus || !C[0].s_src_prev ||
                                        !C[0].s_tgt_prev || !(C[0].a_plus >= 0.f))) return false;
    if ((C[0].rule == SNN_RULE_POSTPRE || outer2) && (!L[0].x || !L[1].x || !L[0].p.lif.traces || !L[1].p.lif.traces)) return false;
    if (C[0].kind == SNN_CONN_MCC && C[0].bias) return false;
    const int B = R->B, Nin = L[0].n, N = L[1].n;
    if (B > MAXB || R->T < 1 || Nin % 16 != 0 || Nin > 65535 || Nin > kMaxTerms) return false;
    if ((double)Nin * N >= 2147483648.0 || (double)(R->T + 1) * B * Nin >= 2147483648.0) return false;
    memset(&c, 0, sizeof(c));
    c.B = B; c.Nin = Nin; c.N = N; c.T = R->T; c.NinW = (Nin + 31) / 32;
    c.MW = (B + 31) / 32; c.BC = 32 * c.MW;
    c.dt = R->dt; c.learning = R->learning;
    c.rule = C[0].rule;
    c.rowmajor = (c.rule == SNN_RULE_POSTPRE || c.rule == SNN_RULE_HEBBIAN || c.rule == SNN_RULE_WDPOSTPRE) && c.learning &&
                 !(getenv("SNN_TWO_ROWMAJOR") && atoi(getenv("SNN_TWO_ROWMAJOR")) == 0);
    c.mstdp_rows = !(getenv("SNN_TWO_MSTDP_ROWS") && atoi(getenv("SNN_TWO_MSTDP_ROWS")) == 0);
    // list capacity: all events of a step in LDS (u16 each), up to 24 KiB
    c.LCAP = (int)((((size_t)B * Nin < 12288 ? (size_t)B * Nin : 12288) + 7) & ~(size_t)7);
    // widest column tile whose weight slice + digest fit LDS, with a tile thread per (sample, column)
    int cw = 8;
    while (cw > 1) {
        c.CW = cw;
        if (B * cw <= NT && run_lds(c) <= 140 * 1024) break;
        cw >>= 1;
    }
    c.CW = cw;
    if (B * cw > NT || run_lds(c) > 140 * 1024) return false;
    c.G = (N + cw - 1) / cw;
    if (c.T + 1 > 4096) return false;
    c.prodw = 0;
    if (C[0].kind == SNN_CONN_DENSE && B * cw <= 64) { c.prodw = 4096; if (run_lds(c) > 140 * 1024) c.prodw = 0; }
    c.use_xsl = 0;
    if (Nin <= NT && (c.rule == SNN_RULE_POSTPRE || c.rule == SNN_RULE_HEBBIAN || c.rule == SNN_RULE_WDPOSTPRE) && !(c.rowmajor && ((size_t)Nin * N) % 32 == 0)) { c.use_xsl = 1; if (run_lds(c) > 140 * 1024) c.use_xsl = 0; }
    // the per-step digest copy must fit the prefetch registers
    if (META + c.LCAP / 2 + Nin * c.MW + (Nin + 1) / 2 + c.NinW > PF * NT) return false;
    return true;
}

}  // namespace

static size_t two_workspace(const TwoCtx &c0) {
    TwoCtx c = c0;
    const int DW = digest_layout(c);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // digests (T+2 entries: MSTDP's extra one) | X trace of every step (PostPre) or p_plus of every step (MSTDP)
    return al((size_t)(c.T + 2) * DW * 4) + al((size_t)(c.T + 1) * c.B * c.Nin * 4);
}

unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                                const snn_run_desc *R) {
    TwoCtx c;
    if (!L || !C || !R || !plan(L, nL, C, nC, R, c)) return 0;
    return two_workspace(c);
}

int snn_try_fused_twolayer(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                           hipStream_t st, int *handled, unsigned *normalized) {
    *handled = 0;
    TwoCtx c;
    if (!plan(L, nL, C, nC, R, c)) return SNN_OK;
    if (!R->workspace || R->workspace_bytes < two_workspace(c)) return SNN_OK;
    if (snn_prof_active()) return SNN_OK;                 // per-timestep event timing only exists for per-step plans
    const int B = c.B, Nin = c.Nin;
    c.DW = digest_layout(c);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    unsigned char *ws = (unsigned char *)R->workspace;
    c.dig = (uint32_t *)ws;
    float *big = (float *)(ws + al((size_t)(c.T + 2) * c.DW * 4));
    const bool mstdp = C[0].rule == SNN_RULE_MSTDP && R->learning;
    c.xall = (C[0].rule == SNN_RULE_POSTPRE || C[0].rule == SNN_RULE_HEBBIAN || C[0].rule == SNN_RULE_WDPOSTPRE) ? big : nullptr;
    c.pall = mstdp ? big : nullptr;
    c.p_plus = C[0].p_plus; c.p_minus = C[0].p_minus; c.s_src_prev = C[0].s_src_prev; c.s_tgt_prev = C[0].s_tgt_prev;
    c.a_plus = C[0].a_plus; c.a_minus = C[0].a_minus; c.d_plus = C[0].decay_plus; c.d_minus = C[0].decay_minus;
    c.reward = C[0].reward; c.reward_vec = C[0].reward_vec;
    c.has_norm = C[0].has_norm; c.norm = C[0].norm; c.norm_abs = C[0].norm_abs;
    c.inv_hwps = 1.0f / (float)(Nin >> 4);
    c.in = L[0].ext_spikes; c.sX0 = L[0].s;
    c.x_traces = L[0].p.lif.traces; c.x_decay = L[0].p.lif.trace_decay; c.x_scale = L[0].p.lif.trace_scale;
    c.x_additive = L[0].p.lif.traces_additive;
    c.xX0 = L[0].x; c.xXout = L[0].x;
    c.vY = L[1].v; c.rY = L[1].refrac; c.xY = L[1].x; c.sY = L[1].s; c.pY = L[1].p.lif;
    c.rasY = L[1].raster_s; c.rasVY = L[1].raster_v;
    c.W = C[0].w; c.bias = C[0].bias; c.cascade = C[0].kind == SNN_CONN_MCC;
    c.rule = C[0].rule; c.nu0 = C[0].nu0; c.nu1 = C[0].nu1; c.use_dt = C[0].use_dt; c.wdecay = C[0].wdecay;
    c.has_min = C[0].has_min; c.wmin = C[0].wmin; c.has_max = C[0].has_max; c.wmax = C[0].wmax;
    static bool attr = false;
    if (!attr) {
        const void *variants[16] = {
            (const void *)k_two_run<false, SNN_RULE_HEBBIAN, 1>, (const void *)k_two_run<false, SNN_RULE_WDPOSTPRE, 1>,
            (const void *)k_two_run<false, SNN_RULE_HEBBIAN, 4>, (const void *)k_two_run<false, SNN_RULE_WDPOSTPRE, 4>,
            (const void *)k_two_run<true, SNN_RULE_NONE, 1>, (const void *)k_two_run<true, SNN_RULE_POSTPRE, 1>,
            (const void *)k_two_run<false, SNN_RULE_NONE, 1>, (const void *)k_two_run<false, SNN_RULE_POSTPRE, 1>,
            (const void *)k_two_run<false, SNN_RULE_MSTDP, 1>, (const void *)k_two_run<true, SNN_RULE_MSTDP, 1>,
            (const void *)k_two_run<true, SNN_RULE_NONE, 4>, (const void *)k_two_run<true, SNN_RULE_POSTPRE, 4>,
            (const void *)k_two_run<false, SNN_RULE_NONE, 4>, (const void *)k_two_run<false, SNN_RULE_POSTPRE, 4>,
            (const void *)k_two_run<false, SNN_RULE_MSTDP, 4>, (const void *)k_two_run<true, SNN_RULE_MSTDP, 4>};
        for (const void *f : variants)
            if (snn_check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) return SNN_ERR_LAUNCH;
        if (snn_check(hipFuncSetAttribute((const void *)k_two_prep, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) return SNN_ERR_LAUNCH;
        attr = true;
    }
    const size_t prep_lds = (size_t)(B * c.NinW + Nin * c.MW + MAXB + 1 + 4) * 4;
    if (prep_lds > 150 * 1024) return SNN_OK;
    if (c.x_traces) hipLaunchKernelGGL(k_two_xtrace, dim3((B * Nin + 255) / 256), dim3(256), 0, st, c);
    if (mstdp) hipLaunchKernelGGL(k_two_pplus, dim3((B * Nin + 255) / 256), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_two_prep, dim3(mstdp ? c.T + 2 : c.T + 1), dim3(NT), prep_lds, st, c);
    static long long *dbg = nullptr;
    if (getenv("SNN_TWO_TIMING")) {
        if (!dbg) (void)hipMalloc(&dbg, sizeof(long long) * 24 * 4096);
        if (c.T + 1 <= 4096) { (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 24 * 4096, st); c.dbg = dbg; }
    }
    {
        const dim3 grid(c.G), blk(NT);
        const size_t lds = run_lds(c);
#define TWO_LAUNCH(MWV) do { \
        if (c.cascade && c.rule == SNN_RULE_POSTPRE) hipLaunchKernelGGL((k_two_run<true, SNN_RULE_POSTPRE, MWV>), grid, blk, lds, st, c); \
        else if (c.cascade && c.rule == SNN_RULE_MSTDP) hipLaunchKernelGGL((k_two_run<true, SNN_RULE_MSTDP, MWV>), grid, blk, lds, st, c); \
        else if (c.cascade) hipLaunchKernelGGL((k_two_run<true, SNN_RULE_NONE, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_POSTPRE) hipLaunchKernelGGL((k_two_run<false, SNN_RULE_POSTPRE, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_MSTDP) hipLaunchKernelGGL((k_two_run<false, SNN_RULE_MSTDP, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_HEBBIAN) hipLaunchKernelGGL((k_two_run<false, SNN_RULE_HEBBIAN, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_WDPOSTPRE) hipLaunchKernelGGL((k_two_run<false, SNN_RULE_WDPOSTPRE, MWV>), grid, blk, lds, st, c); \
        else hipLaunchKernelGGL((k_two_run<false, SNN_RULE_NONE, MWV>), grid, blk, lds, st, c); } while (0)
        if (c.MW == 1) TWO_LAUNCH(1); else TWO_LAUNCH(4);
#undef TWO_LAUNCH
    }
    int rc = snn_check_launch();
    if (rc) return rc;
    if (mstdp)       // the rule's memory of the source spikes = the last input slice
        if ((rc = snn_check(hipMemcpyAsync(c.s_src_prev, c.in + (size_t)(c.T - 1) * B * Nin, (size_t)B * Nin, hipMemcpyDeviceToDevice, st)))) return rc;
    if (c.dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<long long> h((size_t)8 * (c.T + 1));
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double a[7] = {0}; int n = 0;
        for (int t = 2; t < c.T; ++t, ++n) {
            const long long *r = &h[(size_t)t * 8];
            for (int k = 1; k < 7; ++k) a[k] += (double)(r[k] - r[k - 1]) / 100.0;
            a[0] += (double)(h[(size_t)(t + 1) * 8] - r[0]) / 100.0;
        }
        fprintf(stderr, "[twolayer timing, us/step %.2f] commit %.2f | issue %.2f | stdp %.2f | barrier %.2f | currents+lif %.2f | publish %.2f\n",
                a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n);
        std::vector<long long> hw((size_t)16 * (c.T + 1));
        (void)hipMemcpy(hw.data(), dbg + (size_t)8 * 4096, hw.size() * 8, hipMemcpyDeviceToHost);
        if (mstdp) {
            double pub = 0, upd = 0, upd_slow = 0; int slow = 0;
            std::vector<long long> fl((size_t)c.T + 1);
            (void)hipMemcpy(fl.data(), dbg + (size_t)8 * 4096 + (size_t)16 * 4096 - 1 - c.T, fl.size() * 8, hipMemcpyDeviceToHost);
            for (int t = 2; t < c.T; ++t) {
                const long long *r = &h[(size_t)t * 8];
                pub += (double)(r[7] - r[5]) / 100.0; upd += (double)(r[6] - r[7]) / 100.0;
                if (fl[(size_t)c.T - t] != 0) { ++slow; upd_slow += (double)(r[6] - r[7]) / 100.0; }
            }
            fprintf(stderr, "[twolayer timing] publish = spikes/state %.2f + MSTDP update %.2f (%d steps with a target spike in workgroup 0: %.2f each, the other %d: %.2f)\n",
                    pub / n, upd / n, slow, slow ? upd_slow / slow : 0.0, n - slow, n - slow ? (upd - upd_slow) / (n - slow) : 0.0);
        }
        fprintf(stderr, "[twolayer timing] learning phase per wave, us:");
        for (int w = 0; w < 16; ++w) {
            double sum = 0;
            for (int t = 2; t < c.T; ++t) sum += (double)(hw[(size_t)t * 16 + w] - h[(size_t)t * 8 + 2]) / 100.0;
            fprintf(stderr, " %.2f", sum / n);
        }
        fprintf(stderr, "\n");
    }
    if (c.has_norm) *normalized |= 1u;                    // connection 0 was normalised in the kernel's epilogue
    snn_set_plan_name("twolayer-fused");
    *handled = 1;
    return SNN_OK;
}

// snn_dc2015.hip -- fused plan "dc2015-fused": ONE kernel launch per timestep for the
// DiehlAndCook2015 graph (Input -X->Ae (PostPre)-> DiehlAndCookNodes <-> LIFNodes), i.e. the loop body
// of bindsnet/network/network.py:380-461 for the wiring of bindsnet/models/models.py:156-244.
//
// Decomposition.  Workgroup g owns 32 consecutive target columns (neurons c0..c0+31 of BOTH Ae and
// Ai) for every sample of the batch: their membrane state, adaptive thresholds, post-synaptic
// traces and the [Nin x 8] column slice of the learned weights.  Everything a step needs from
// other columns is spikes, exchanged as bit masks through global memory across the kernel boundary:
//   crossE[t&1][b][byte]  Ae threshold crossings of step t (before one_spike arbitration), 1 byte per workgroup
//   spikeI[t&1][b][byte]  Ai spikes of step t                                          (read back as u32 words)
// The only cross-column computation -- the one_spike arbitration, which needs every crossing of a row
// and the host generator's noise stream -- is tiny, so every workgroup repeats it redundantly instead
// of paying a second exchange.  That makes the step a software pipeline: launch t
//   phase A  finishes step t-1: arbitration -> final Ae spikes -> Ae trace, raster, STDP on own columns
//   phase B  starts step t:     currents from step t-1 spikes -> Ae/Ai membrane update -> publish bits,
//                               X-trace update for an own slice of input rows
// and launch T runs phase A only.  All arithmetic follows the reference's f32 operation order
// (snn_order.hpp, snn_common.hpp, snn_rng.hpp); results are bit-identical to the generic plan.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/snnhip.h"
#include "snn_common.hpp"
#include "snn_order.hpp"
#include "snn_rng.hpp"

using namespace snn;

unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                                const snn_run_desc *R);
bool snn_prof_begin(int t, hipStream_t st);
bool snn_prof_active();
void snn_prof_end(hipStream_t st);

namespace {

// A single wave retires roughly one instruction every 4 cycles however idle the chip is, so the cost of a
// launch is the instruction count on each thread's critical path.  Hence: MANY workgroups (8 columns each)
// so the per-column work of a thread is small, each with MANY threads (1024) so the work every workgroup
// repeats (staging the step's spikes, arbitration) and the STDP items are spread thin.
constexpr int CW = 8;           // columns per workgroup
constexpr int MAXB = 32;        // samples (batch) per workgroup
constexpr int TT = MAXB * CW;   // "tile threads": thread tid < TT <-> (sample tid / CW, column tid % CW)
constexpr int NT = 1024;        // threads per workgroup
constexpr int NU = 2;           // staged 16-byte pieces per thread: B*Nin <= NU*NT*16 = 32 KiB

// Barrier for LDS-only hand-offs: waits for this wave's LDS traffic but NOT for its outstanding global
// stores (a plain __syncthreads() drains vmcnt and costs a full memory round trip every time).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct DcCtx {
    int B, Nin, N, T, NW, NinW, G, RS;     // NW = ceil(N/32), NinW = ceil(Nin/32), G workgroups, RS input rows per WG
    float dt; int learning;
    // X (Input)
    const uint8_t *in;          // [T,B,Nin]
    const uint8_t *sX0;         // [B,Nin] X.s at entry
    float *xX[2];               // trace after step t lives in xX[t&1]; entry trace in xX[1]
    int x_traces; float x_decay, x_scale; int x_additive;
    // Ae (DiehlAndCookNodes)
    float *vE, *rE, *xE, *theta; uint8_t *sE;
    snn_dc_params pE;
    uint8_t *rasE; float *rasVE;
    // Ai (LIFNodes)
    float *vI, *rI, *xI; uint8_t *sI;
    snn_lif_params pI;
    uint8_t *rasI; float *rasVI;
    // weights
    float *Wxe; const float *Wei; const float *Wie;
    int rule; float nu0, nu1; int use_dt; int has_min; float wmin; int has_max; float wmax;
    // exchange + generator
    uint32_t *crossE[2], *spikeI[2];
    snn_rng_state *rng[2];
    float inv_hwps, inv_NW, inv_RS;   // reciprocals of Nin/16, NW, RS for the exact float-multiply divisions
    // per-step digest of the X spikes, produced once per run by k_dc2015_prep (entry e <-> spikes of step e-1):
    // [B*NinW] bit words | [B*LX/2] u16 event lists | [40] meta (counts, n active rows, flags) | [Nin] row masks |
    // [Nin/2] u16 active rows | [Nin/2] u16 row -> compact index | [B*LX/2] u16 event lists grouped by row_sum lane |
    // [B] group sizes (5 bits each: lanes 0..3, leftover sources) | [B] events per 256-position group (5 bits each)
    uint32_t *dig; int DW, DGW, OXW;        // words per entry, words of its LDS part, offset of its bit words
    // resident plan (k_dc2015_run): 8-byte {epoch, bits} exchange granules [2][G][KB], the X trace after every
    // step [T+1][B][Nin] (entry 0 = trace at run entry), device status word
    unsigned long long *ex; int KB;
    float *xtr;
    int *status;
    int has_norm; float norm; int norm_abs;   // post-run normalisation of Wxe, done in the resident kernel's epilogue
    int dbg_wg;
    long long *dbg;             // developer aid (SNN_DC_TIMING=1): per-launch phase timestamps of workgroup 0
};

#define DBG_MARK(slot) do { if (c.dbg && blockIdx.x == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + (slot)] = (long long)wall_clock64(); } while (0)

__device__ __forceinline__ bool bit_of(const uint32_t *w, int j) { return (w[j >> 5] >> (j & 31)) & 1u; }

// Input currents of neuron j of sample b from the previous step's spikes, in connection insertion
// order (network.py:225-248): Ae <- (zeros + X->Ae) + Ai->Ae ; Ai <- zeros + Ae->Ai.  SUM selects the
// ATen column class of j (multi_row_sum for j < 32*floor(N/32), row_sum otherwise).
// ---- small helpers ----------------------------------------------------------------------------
// 4-bit mask of the non-zero bytes of a 32-bit word (byte k -> bit k).
__device__ __forceinline__ uint32_t nz4(uint32_t w) {
    const uint32_t t = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    return ((t >> 7) | (t >> 14) | (t >> 21) | (t >> 28)) & 0xFu;
}

// Ordered sum of W[i, j] * value(i) over the sources i whose bit is set in `words`, visited in
// ascending i: `wmask` has one bit per NON-ZERO word, so silent stretches cost nothing.  Weight loads
// are issued 8 at a time before the (order-constrained) adds.  vals == nullptr: all spikes are 1.
template <class SUM>
__device__ __forceinline__ float ordered_dot(const float *__restrict__ W, int N, int j, const uint32_t *words,
                                             uint64_t wmask, const uint8_t *__restrict__ vals, int n_terms) {
    SUM a; a.init();
    int idx[8]; float wv[8];
    int nq = 0;
    while (wmask) {
        const int w = __ffsll((unsigned long long)wmask) - 1; wmask &= wmask - 1;
        uint32_t m = words[w];
        while (m) {
            idx[nq++] = w * 32 + __ffs(m) - 1; m &= m - 1;
            if (nq == 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = W[idx[u] * N + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) a.add(idx[u], wv[u] * (vals ? (float)vals[idx[u]] : 1.0f), n_terms);
                nq = 0;
            }
        }
    }
    if (nq) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = W[idx[u < nq ? u : 0] * N + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < nq) a.add(idx[u], wv[u] * (vals ? (float)vals[idx[u]] : 1.0f), n_terms);
    }
    return a.finish(n_terms);
}

// PostPre for (row, column) items of the own weight slice (MCC_learning.py:224-302, :86-110): rows listed in
// `arows` (all rows when FULL) x the CW own columns.  SUM is CascadeT unless the [Nin*N] element index can
// fall into ATen's <32-element tail (OuterSum).  sbytes: the step's X spike bytes when some spike value
// is not 0/1 (else nullptr: every spike counts 1.0).
template <class SUM, bool FULL>
__device__ __forceinline__ void stdp_rows(const DcCtx &c, int nact, const uint16_t *arows, const uint32_t *rowmask,
                                          const uint32_t *colmask, const uint8_t *__restrict__ sbytes,
                                          const float *xnu0, const float *__restrict__ xsrc, float *wtile, int c0, int tid,
                                          int Emain, const float *pre_w) {
    const int B = c.B, Nin = c.Nin, N = c.N;
    const int nitems = nact * CW;
    const int q = tid % CW;                                       // NT % CW == 0: a thread keeps its column
    for (int base = 0; base < nitems; base += NT * 8) {
        float wv[8]; int ev[8]; int iv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int item = base + u * NT + tid;
            ev[u] = -1; iv[u] = 0; wv[u] = 0.f;
            if (item < nitems) {
                const int i = FULL ? (item / CW) : (int)arows[item / CW];
                const int jq = c0 + q;
                if (jq < N) { iv[u] = i; ev[u] = i * N + jq; wv[u] = (base == 0) ? pre_w[u] : c.Wxe[ev[u]]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ev[u] < 0) continue;
            const int i = iv[u];
            float w = wv[u];
            if (c.nu0 != 0.f) {                                  // w -= dt * sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
                uint32_t m = rowmask[i];
                float uu = 0.f;
                if (m) {
                    SUM acc; acc.init(ev[u] >= Emain);
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
            if (c.nu1 != 0.f) {                                  // w += dt * sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
                uint32_t m = colmask[q];
                float uu = 0.f;
                if (m) {
                    SUM acc; acc.init(ev[u] >= Emain);
                    while (m) {
                        const int b = __ffs(m) - 1; m &= m - 1;
                        acc.add(b, xsrc[b * Nin + i] * (1.0f * c.nu1), B);
                    }
                    uu = acc.finish(B);
                }
                if (c.use_dt) uu = uu * c.dt;
                w = w + uu;
            }
            if (c.has_min && w < c.wmin) w = c.wmin;
            if (c.has_max && w > c.wmax) w = c.wmax;
            c.Wxe[ev[u]] = w;
            wtile[((base + u * NT + tid) / CW) * CW + q] = w;     // compact row index x column, read back by phase B
        }
    }
}

// Columns with a post-synaptic spike x rows WITHOUT a pre-synaptic spike (those rows were not
// visited by stdp_rows): w = (w - 0) + dt * sum_b ... ; clamp.
template <class SUM>
__device__ __forceinline__ void stdp_cols(const DcCtx &c, uint32_t active_cols, const uint32_t *rowmask,
                                          const uint32_t *colmask, const float *__restrict__ xsrc, int c0, int tid,
                                          int Emain) {
    const int B = c.B, Nin = c.Nin, N = c.N;
    while (active_cols) {
        const int q = __ffs(active_cols) - 1; active_cols &= active_cols - 1;
        const uint32_t cm = colmask[q];
        const int jq = c0 + q;
        for (int i = tid; i < Nin; i += NT) {
            if (rowmask[i]) continue;
            const int e = i * N + jq;
            float w = c.Wxe[e];
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
            c.Wxe[e] = w;
        }
    }
}

// Cascade taking (and ignoring) the `tail` flag at init, interface-compatible with OuterSum.
struct CascadeT {
    Cascade c;
    __device__ __forceinline__ void init(bool) { c.init(); }
    __device__ __forceinline__ void add(int pos, float term, int n) { c.add(pos, term, n >> 4); }
    __device__ __forceinline__ float finish(int n) { return c.finish(n >> 4); }
};

constexpr int LX = 32, LR = 8;   // per-sample event-list capacities (X sources / recurrent sources)
constexpr int NCAND = 2048;      // one_spike candidates evaluated one per thread (more: serial fallback)

// One wave turns a row of spike bit words into the ascending list of set-bit indices (first `cap`
// entries stored) and returns the total count.  nwords <= 64.
__device__ __forceinline__ int build_list(const uint32_t *words, int nwords, int lane, uint16_t *out, int cap) {
    uint32_t m = lane < nwords ? words[lane] : 0u;
    const int cn = __popc(m);
    // exclusive prefix of cn over lanes = sum_k popc(ballot(cn > k) & lanes_below): counts are tiny, so a
    // few ballots beat a 6-step cross-lane scan
    const uint64_t below = (1ull << lane) - 1ull;
    int offp = 0, total = 0;
    for (int k = 0; ; ++k) {
        const uint64_t bm = __ballot(cn > k);
        if (!bm) break;
        offp += __popcll(bm & below);
        total += __popcll(bm);
    }
    while (m) {
        const int i = lane * 32 + __ffs(m) - 1; m &= m - 1;
        if (offp < cap) out[offp] = (uint16_t)i;
        ++offp;
    }
    return total;
}

// Two rows per wave: lanes 0..31 list row A, lanes 32..63 row B (nwords <= 32).  `words` / `out` are the calling
// lane's own row; returns that row's total count.  Rows past the end: pass words == nullptr.
__device__ __forceinline__ int build_list_half(const uint32_t *words, int nwords, int lane, uint16_t *out, int cap) {
    const int hl = lane & 31;
    const uint64_t halfmask = (lane & 32) ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
    uint32_t m = (words && hl < nwords) ? words[hl] : 0u;
    const int cn = __popc(m);
    const uint64_t below = ((1ull << lane) - 1ull) & halfmask;
    int offp = 0, total = 0;
    for (int k = 0; ; ++k) {
        const uint64_t bm = __ballot(cn > k);
        if (!bm) break;
        offp += __popcll(bm & below);
        total += __popcll(bm & halfmask);
    }
    while (m) {
        const int i = hl * 32 + __ffs(m) - 1; m &= m - 1;
        if (offp < cap) out[offp] = (uint16_t)i;
        ++offp;
    }
    return total;
}

// Input currents of neuron j of sample b from the previous step's spikes, in connection insertion order
// (network.py:225-248): Ae <- (zeros + X->Ae) + Ai->Ae ; Ai <- zeros + Ae->Ai, each summed in ascending source
// order.  X->Ae weights come from the LDS tile the STDP pass just refreshed (wtile != nullptr: row `rowpos[i]`
// of the compacted active rows, or row i itself when rowpos == nullptr) or from global memory; the recurrent
// weights wi / we were prefetched by the caller.
template <class SUM, int CWL = CW>
__device__ __forceinline__ void tile_currents(const DcCtx &c, const uint16_t *lx, int nX, const uint16_t *li, int nI,
                                              const uint16_t *le, int nE, const float *wi, const float *we,
                                              const float *wtile, const uint16_t *rowpos, int jj,
                                              const uint8_t *__restrict__ xb, int j, float &curE, float &curI) {
    const int Nin = c.Nin, N = c.N;
    int ix[16], ii[4], ie[4]; float wx[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ii[u] = (int)li[u]; ie[u] = (int)le[u]; }
    // unconditional, clamped gathers (entries past nX are stale but in range): the 16 reads of each stage
    // are independent, so the three dependent LDS stages cost three latencies, not forty-eight
#pragma unroll
    for (int u = 0; u < 16; ++u) ix[u] = min((int)lx[u], Nin - 1);
    if (wtile) {
        int rr[16];
        if (rowpos) {
#pragma unroll
            for (int u = 0; u < 16; ++u) rr[u] = min((int)rowpos[ix[u]], Nin - 1);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) rr[u] = ix[u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) wx[u] = wtile[rr[u] * CWL + jj];
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) wx[u] = c.Wxe[ix[u] * N + j];
    }
    SUM a; a.init();
    if (xb) {
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nX) a.add(ix[u], wx[u] * (float)xb[ix[u]], Nin);
    } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nX) a.add(ix[u], wx[u] * 1.0f, Nin);
    }
    curE = 0.0f + a.finish(Nin);
    a.init();
#pragma unroll
    for (int u = 0; u < 4; ++u) if (u < nI) a.add(ii[u], wi[u] * 1.0f, N);
    curE = curE + a.finish(N);
    a.init();
#pragma unroll
    for (int u = 0; u < 4; ++u) if (u < nE) a.add(ie[u], we[u] * 1.0f, N);
    curI = 0.0f + a.finish(N);
}

// ATen "row_sum" columns (j >= 32*floor(N/32)): the reference sums the sources in four interleaved lanes
// (source index mod 4), each lane a cascade over its n/4 sources, leftovers (n % 4) added to lane 0, lanes
// combined ((l0+l1)+l2)+l3.  Four adjacent threads take one lane each of the same (sample, column) -- the
// lanes really are independent -- and lane 0 combines them with quad shuffles.
template <int NMAX>
__device__ __forceinline__ float quad_lane_sum(const int *ix, int cnt, const float *wv, const uint8_t *vals, int n, int L) {
    const int n4 = n >> 2;
    CascadeFlat a; a.init();
    float tailsum = 0.f;                 // lane 0 only: its combined cascade + leftovers, once the first leftover arrives
    bool closed = false;
#pragma unroll
    for (int u = 0; u < NMAX; ++u) {
        if (u < cnt) {
            const int i = ix[u];
            const float term = wv[u] * (vals ? (float)vals[i] : 1.0f);
            if (i >= (n4 << 2)) {
                if (L == 0) { if (!closed) { tailsum = a.finish(n4); closed = true; } tailsum += term; }
            } else if ((i & 3) == L) {
                a.add(i >> 2, term, n4);
            }
        }
    }
    float v = closed ? tailsum : a.finish(n4);
    const float v1 = __shfl_down(v, 1, 4), v2 = __shfl_down(v, 2, 4), v3 = __shfl_down(v, 3, 4);
    return ((v + v1) + v2) + v3;         // meaningful in lane 0 of the quad
}

// Once per run: digest the X spikes of every step (entry 0 = the layer's `s` at entry, entry e = inputs[e-1]),
// fully parallel over steps and off the per-timestep critical path.  One workgroup per entry.
__global__ __launch_bounds__(NT) void k_dc2015_prep(const DcCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, NinW = c.NinW;
    uint32_t *sXw = (uint32_t *)smem;                               // [B][NinW]
    uint32_t *rowmask = sXw + B * NinW;                             // [Nin]
    int *misc = (int *)(rowmask + Nin);                             // [0] nact, [1] flags
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = blockIdx.x;
    const uint8_t *src = (e == 0) ? c.sX0 : c.in + (size_t)(e - 1) * B * Nin;
    uint32_t *D = c.dig + (size_t)e * c.DW;
    uint32_t *D_xl = D, *D_meta = D_xl + B * (LX / 2), *D_rm = D_meta + 40, *D_xw = D + c.OXW;   // (bit words last)
    uint16_t *D_ar = (uint16_t *)(D_rm + Nin), *D_rp = D_ar + 2 * ((Nin + 1) / 2);
    for (int k = tid; k < Nin; k += NT) rowmask[k] = 0;
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
                if (any & 0xFEFEFEFEu) {           // some byte is not 0/1: generic non-zero test
                    big = 1;
                    m16 = nz4(v.x) | (nz4(v.y) << 4) | (nz4(v.z) << 8) | (nz4(v.w) << 12);
                } else {                           // 0/1 bytes: byte k contributes 2^(8k) * 2^(24-7k) = 2^(24+k); the cross
                    m16 = ((v.x * 0x01020408u) >> 24) | (((v.y * 0x01020408u) >> 24) << 4) |      // terms fall on distinct
                          (((v.z * 0x01020408u) >> 24) << 8) | (((v.w * 0x01020408u) >> 24) << 12);   // lower bits or overflow
                }
            }
            sXh[b * HS + hw] = (uint16_t)m16;
            if (hw == hwps - 1 && (hwps & 1)) sXh[b * HS + hw + 1] = 0;
            while (m16) {
                const int i = hw * 16 + __ffs(m16) - 1; m16 &= m16 - 1;
                atomicOr(&rowmask[i], 1u << b);
            }
        }
        if (big) atomicOr((unsigned int *)&misc[1], 1u);
    }
    __syncthreads();
    uint16_t *D_l2 = D_rp + 2 * ((Nin + 1) / 2);                     // [B][LX] events grouped by row_sum lane
    uint32_t *D_gc = (uint32_t *)(D_l2 + B * LX);                    // [B] five 5-bit group sizes
    uint16_t *lscr = (uint16_t *)(misc + 4) + wave * LX;             // this wave's scratch list
    for (int b = wave; b < B; b += NT / 64) {
        const int nx = build_list(sXw + b * NinW, NinW, lane, lscr, LX);
        if (lane == 0) { D_meta[b] = (uint32_t)nx; if (nx > 16) atomicOr((unsigned int *)&misc[1], 2u); }
        // (LDS operations of one wave execute in program order: the list is readable right away)
        const bool have = lane < LX && lane < nx;
        const int i = have ? (int)lscr[lane] : 0;
        if (lane < LX) ((uint16_t *)D_xl)[b * LX + lane] = (uint16_t)i;
        // the same events grouped by ATen row_sum lane (index mod 4; group 4 = the n % 4 leftover sources), ascending
        // inside a group: what a quad of threads walks for a column >= 32*floor(N/32)
        const bool in16 = have && lane < 16;
        const int grp = (i >= ((Nin >> 2) << 2)) ? 4 : (i & 3);
        int start = 0, my = 0; uint32_t gc = 0;
        for (int k = 0; k < 5; ++k) {
            const uint64_t mk = __ballot(in16 && grp == k);
            const int ck = __popcll(mk);
            if (grp == k) my = start + __popcll(mk & ((1ull << lane) - 1ull));
            start += ck; gc |= (uint32_t)ck << (5 * k);
        }
        uint16_t *perm = lscr + (NT / 64) * LX;        // second per-wave scratch: permute in LDS, store each slot once
        if (lane < LX) perm[lane] = 0;
        if (in16) perm[my] = (uint16_t)i;
        if (lane < LX) D_l2[b * LX + lane] = perm[lane];
        if (lane == 0) D_gc[b] = gc;
        // ... and how many of the (ascending) events fall into each 256-position group of the cascade order: a quad of
        // threads of a multi_row_sum column sums one group each
        {
            uint32_t gq = 0;
            for (int k = 0; k < 4; ++k) {
                const uint64_t mk = __ballot(in16 && min(i >> 8, 3) == k);
                gq |= (uint32_t)__popcll(mk) << (5 * k);
            }
            if (lane == 0) D_gc[B + b] = gq;
        }
    }
    for (int k = tid; k < B * NinW; k += NT) D_xw[k] = sXw[k];
    for (int base = 0; base < Nin; base += NT) {       // compact the rows with a spike in any sample
        const int i = base + tid;
        const bool o = i < Nin && rowmask[i] != 0;
        const uint64_t m = __ballot(o);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&misc[0], __popcll(m));
        wbase = __shfl(wbase, 0);
        if (o) { const int cp = wbase + __popcll(m & ((1ull << lane) - 1ull)); D_ar[cp] = (uint16_t)i; D_rp[i] = (uint16_t)cp; }
        if (i < Nin) D_rm[i] = rowmask[i];
    }
    __syncthreads();
    if (tid == 0) { D_meta[32] = (uint32_t)misc[0]; D_meta[33] = (uint32_t)misc[1]; }
}

__global__ __launch_bounds__(NT) void k_dc2015_step(const DcCtx c, const int t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, N = c.N, NW = c.NW, NinW = c.NinW;
    // ---- LDS carve-up (all offsets multiples of 16 bytes)
    size_t off = 0;
    // digest of the X spikes of step t-1, copied verbatim from c.dig (layout: see DcCtx)
    uint32_t *dg = (uint32_t *)(smem + off); off += ((size_t)c.DGW * 4 + 15) & ~(size_t)15;
    uint16_t *lstX = (uint16_t *)dg;                                  // [B][LX] per-sample X event lists
    int *meta = (int *)(dg + B * (LX / 2));                           // [0..31] list lengths, [32] n active rows, [33] flags
    int *cntX = meta;
    uint32_t *rowmask = dg + B * (LX / 2) + 40;                       // [Nin] samples in which row i spiked
    uint16_t *arows = (uint16_t *)(rowmask + Nin);                    // compacted active rows
    uint16_t *rowpos = arows + 2 * ((Nin + 1) / 2);                   // row -> compact active-row index
    uint32_t *crs = (uint32_t *)(smem + off); off += ((size_t)B * NW * 4 + 15) & ~(size_t)15;     // Ae crossings t-1
    uint32_t *finE = (uint32_t *)(smem + off); off += ((size_t)B * NW * 4 + 15) & ~(size_t)15;    // Ae final spikes t-1
    uint32_t *spI = (uint32_t *)(smem + off); off += ((size_t)B * NW * 4 + 15) & ~(size_t)15;     // Ai spikes t-1
    float *xnu0 = (float *)(smem + off); off += (size_t)MAXB * CW * 4;                            // [B][CW] x_tgt*nu0
    uint32_t *mt = (uint32_t *)(smem + off); off += 8 * 624 * 4;                                  // mt19937 blocks m, m+1, ... in slot (block & 7)
    uint32_t *cand = (uint32_t *)(smem + off); off += NCAND * 4;                                  // one_spike candidates (sample << 16 | column)
    unsigned long long *keys = (unsigned long long *)(smem + off); off += MAXB * 8;               // argmax keys per sample
    uint16_t *lstI = (uint16_t *)(smem + off); off += MAXB * LR * 2;                              // ... Ai spikes
    uint16_t *lstE = (uint16_t *)(smem + off); off += MAXB * LR * 2;                              // ... final Ae spikes
    int *cntI = (int *)(smem + off); off += MAXB * 4;
    int *cntE = (int *)(smem + off); off += MAXB * 4;
    int *cnt = (int *)(smem + off); off += 32 * 4;                                                // crossings per column
    uint32_t *colmask = (uint32_t *)(smem + off); off += 32 * 4;                                  // samples whose final Ae spike is column jj
    int *misc = (int *)(smem + off); off += 32;      // [4] number of one_spike candidates; [0] n active rows, [1] active column mask, [2] spike value > 1, [3] samples with a crossing
    float *wtile = (float *)(smem + off); off += (size_t)Nin * CW * 4;                            // refreshed own weights of active rows
    float *curbuf = (float *)(smem + off); off += 2 * MAXB * CW * 4;                              // tail-column currents handed to the tile threads

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, c0 = g * CW;
    const int jj = tid % CW, bl = tid / CW;          // tile threads (tid < TT) <-> (sample bl, column c0+jj)
    const int j = c0 + jj;
    const bool colv = j < N;
    const bool tailcol = c0 >= (N / 32) * 32;
    const bool phaseA = t >= 1, phaseB = t < c.T;
    const int pprev = (t + 1) & 1, pcur = t & 1;     // parity of step t-1 / step t
    const int BW = B * NW;
    DBG_MARK(0);
    if (c.dbg && blockIdx.x == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + 8] = (long long)clock64();
    if (c.dbg && threadIdx.x == 0) atomicMin((unsigned long long *)&c.dbg[(size_t)t * 24 + 20], (unsigned long long)wall_clock64());

    // ------------------------------------------------------------------ stage inputs
    // Every global load that does not depend on this launch's arbitration is issued here, together,
    // so the prologue costs ONE memory round trip: X spikes of step t-1, exchanged bit words, generator
    // state, the own tile's membrane state and the own slice of the X trace.
    const bool mine = tid < TT && bl < B && colv;
    const int kst = bl * N + j;
    const int stepoff = t * B * Nin;                                   // < 2^31 (host check)
    const uint8_t *sprev_g = (t == 0) ? c.sX0 : c.in + (stepoff - B * Nin);
    const uint32_t *Dg = c.dig + (size_t)t * c.DW;                   // digest of step t-1 (entry t): the part staged in LDS comes first
    uint32_t r_dg[3];                                                  // DGW <= 3 * NT (host check)
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int k = tid + u * NT; r_dg[u] = k < c.DGW ? Dg[k] : 0u; }
    // exchange word owned by this thread: (sample wb, word wj)
    const int wb = (int)(((float)tid + 0.5f) * c.inv_NW), wj = tid - wb * NW;
    uint32_t r_crs = 0, r_spi = 0, r_mt = 0;                           // B*NW <= NT and 624 <= NT
    int rng_pos = 0; long long rng_consumed = 0;
    if (tid < BW) {
        if (phaseA) { r_crs = c.crossE[pprev][tid]; r_spi = c.spikeI[pprev][tid]; }
        else {   // t == 0: previous spikes come from the layers' `s` tensors (bytes -> bits)
            uint32_t me = 0, mi = 0;
            for (int qq = 0; qq < 32; ++qq) {
                const int jx = wj * 32 + qq;
                if (jx < N) { me |= (uint32_t)(c.sE[wb * N + jx] != 0) << qq; mi |= (uint32_t)(c.sI[wb * N + jx] != 0) << qq; }
            }
            r_crs = me; r_spi = mi;
        }
    }
    const bool use_rng = phaseA && c.pE.one_spike;
    if (use_rng) {
        if (tid < 624) r_mt = c.rng[pprev]->mt[tid];
        rng_pos = c.rng[pprev]->pos; rng_consumed = c.rng[pprev]->consumed;
    }
    float r_vE = 0.f, r_rE = 0.f, r_vI = 0.f, r_rI = 0.f, r_xE = 0.f, r_xI = 0.f, r_theta = 0.f;
    if (mine) {
        if (phaseB) {
            r_vE = c.vE[kst]; r_rE = c.rE[kst]; r_vI = c.vI[kst]; r_rI = c.rI[kst]; r_theta = c.theta[j];
            if (c.pI.traces) r_xI = c.xI[kst];
        }
        if (phaseA && c.pE.lif.traces) r_xE = c.xE[kst];
    }
    // own slice of the X trace: RS rows x B samples, one item per thread (loop below covers larger slices)
    const int xr0 = g * c.RS, xr1 = min(Nin, xr0 + c.RS);
    const int xb_ = (int)(((float)tid + 0.5f) * c.inv_RS), xi_ = xr0 + (tid - xb_ * c.RS);
    const bool xmine = phaseB && c.x_traces && xb_ < B && xi_ < xr1;
    float r_xo = 0.f; uint8_t r_xs = 0;
    if (xmine) { r_xo = c.xX[pprev][xb_ * Nin + xi_]; r_xs = c.in[stepoff + xb_ * Nin + xi_]; }
    if (tid < 32) { cnt[tid] = 0; colmask[tid] = 0; }
    if (tid < 8) misc[tid] = 0;
    if (tid < MAXB) keys[tid] = 0ull;
    DBG_MARK(10);
    __syncthreads();                                   // zeroed arrays visible; the loads above have landed
    DBG_MARK(11);
    // ---- into LDS: the digest verbatim, the exchanged bit words, the generator block
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int k = tid + u * NT; if (k < c.DGW) dg[k] = r_dg[u]; }
    if (tid < BW) { (phaseA ? crs : finE)[tid] = r_crs; spI[tid] = r_spi; }
    if (use_rng && tid < 624) mt[tid] = r_mt;
    lds_barrier();
    DBG_MARK(15);
    const uint8_t *sbytes = (meta[33] & 1) ? sprev_g : nullptr;    // a spike byte other than 0/1: multiply by its value
    if (tid == 0 && (meta[33] & 2)) atomicOr((unsigned int *)&misc[2], 2u);   // a busy sample: phase B takes the generic path
    // ---- PostPre weight rows: which (row, column) items this thread updates is known from the digest, so the
    //      loads are issued NOW and complete behind the arbitration
    const bool do_stdp = phaseA && c.learning && c.rule == SNN_RULE_POSTPRE;
    const bool stdp_full = t == 1;                     // first update of a run touches (clamps) every element
    const int nact = stdp_full ? Nin : meta[32];
    float pre_w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pre_w[u] = 0.f;
        const int item = u * NT + tid;
        if (do_stdp && item < nact * CW && c0 + (tid % CW) < N) {
            const int i = stdp_full ? item / CW : (int)arows[item / CW];
            pre_w[u] = c.Wxe[i * N + c0 + (tid % CW)];
        }
    }
    DBG_MARK(16);
    // ---- per sample (one wave each, in turns): event list of its Ai spikes; does it have an Ae crossing?
    //      Meanwhile the LAST wave runs the generator two blocks ahead (lockstep, no barrier): a step with up to
    //      ~1.5 crossing rows then needs no further twisting on the critical path.
    constexpr int NWV = NT / 64;
    if (wave == NWV - 1 && use_rng) {
        mt_twist_block_wave(mt, mt + 624, lane);
        mt_twist_block_wave(mt + 624, mt + 2 * 624, lane);
    } else if (wave < NWV - 1 || !use_rng) {
        const int stride = use_rng ? NWV - 1 : NWV;
        for (int b = wave; b < B; b += stride) {
            const int ni = build_list(spI + b * NW, NW, lane, lstI + b * LR, LR);
            const uint64_t mc = __ballot(use_rng && lane < NW && crs[b * NW + lane] != 0);
            if (lane == 0) {
                cntI[b] = ni;
                if (mc) atomicOr((unsigned int *)&misc[3], 1u << b);
                if (ni > 4) atomicOr((unsigned int *)&misc[2], 2u);   // a busy sample: phase B takes the generic path
            }
        }
    }
    DBG_MARK(1);
    lds_barrier();
    // who computes input currents: the tile threads, or -- in a workgroup of ATen row_sum columns -- every
    // thread as (sample, column, lane) with four lanes per (sample, column)
    const int cb_ = tailcol ? tid / (CW * 4) : bl, cj_ = tailcol ? (tid >> 2) % CW : jj, cL = tid & 3;
    const int cjg = c0 + cj_;
    const bool cvalid = phaseB && cb_ < B && cjg < N && (tailcol || tid < TT);
    // recurrent Ai->Ae weights: issued now, consumed in phase B
    float wi[4] = {0.f, 0.f, 0.f, 0.f};
    if (cvalid) {
        const int nI = cntI[cb_];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < nI) wi[u] = c.Wie[(int)lstI[cb_ * LR + u] * N + cjg];
    }

    // ================================================================== phase A: finish step t-1
    int E = 0, ntw = 0, rows = 0;
    if (use_rng) {
        // ---- A1: one_spike arbitration, identical in every workgroup (nodes.py:1097-1105)
        DBG_MARK(12);
        const uint32_t anym = (uint32_t)misc[3];
        rows = __popc(anym);
        const int pos = rng_pos;                               // <= 624
        E = pos + 2 * rows * N;
        ntw = rows ? (E - 1) / 624 : 0;
        const int myrank = __popc(anym & ((1u << (wb & 31)) - 1u));   // rank of sample wb among rows with a crossing
        // ---- enumerate the candidates (threshold crossers of rows that crossed) into a compact list
        if (tid < BW) {
            uint32_t bits = crs[tid];
            if (bits) {
                int at = atomicAdd(&misc[4], __popc(bits));
                while (bits) {
                    const int jx = wj * 32 + __ffs(bits) - 1; bits &= bits - 1;
                    if (at < NCAND) cand[at] = ((uint32_t)wb << 16) | (uint32_t)jx;
                    ++at;
                }
            }
        }
        // the generator blocks the step consumes: 0..2 were produced speculatively; up to 7 fit in the ring
        if (ntw > 2 && ntw <= 7 && wave == NWV - 1)
            for (int m = 2; m < ntw; ++m) mt_twist_block_wave(mt + (m & 7) * 624, mt + ((m + 1) & 7) * 624, lane);
        lds_barrier();
        const int ncand = misc[4];
        if (ncand <= NCAND && ntw <= 7) {
            // fast path: every block is resident, one candidate per thread
            for (int k = tid; k < ncand; k += NT) {
                const uint32_t cd = cand[k];
                const int b = (int)(cd >> 16), jx = (int)(cd & 0xFFFFu);
                const int d = __popc(anym & ((1u << b) - 1u)) * N + jx;
                const int w0 = pos + 2 * d, w1 = w0 + 1;
                const int m0 = w0 / 624, m1 = w1 / 624;
                const float q = exp1_from_words(mt_temper(mt[(m0 & 7) * 624 + w0 - 624 * m0]),
                                                mt_temper(mt[(m1 & 7) * 624 + w1 - 624 * m1]));
                const float val = 1.0f / q;                             // p / q with p = 1
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx);
                atomicMax(&keys[b], key);                              // max value, ties -> lowest index
            }
        } else {
            // slow path (very many crossing rows / candidates): walk the stream block by block
            uint32_t parked = 0;
            int lo = 0, hi = min(ntw, 2);
            while (rows) {
                if (tid < BW) {
                    uint32_t bits = crs[tid];
                    while (bits) {
                        const int jx = wj * 32 + __ffs(bits) - 1; bits &= bits - 1;
                        const int d = myrank * N + jx;
                        const int w0 = pos + 2 * d, w1 = w0 + 1;
                        const int m0 = w0 / 624, m1 = w1 / 624;
                        float q; bool have = false;
                        if (m0 >= lo && m1 <= hi) {
                            q = exp1_from_words(mt_temper(mt[(m0 & 7) * 624 + w0 - 624 * m0]),
                                                mt_temper(mt[(m1 & 7) * 624 + w1 - 624 * m1])); have = true;
                        } else if (m0 >= lo && m0 <= hi) {             // pair straddles the resident range: park the high word
                            parked = mt_temper(mt[(m0 & 7) * 624 + w0 - 624 * m0]);
                        } else if (m1 >= lo && m1 <= hi) {
                            q = exp1_from_words(parked, mt_temper(mt[(m1 & 7) * 624 + w1 - 624 * m1])); have = true;
                        }
                        if (have) {
                            const float val = 1.0f / q;
                            const unsigned long long key =
                                ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx);
                            atomicMax(&keys[wb], key);
                        }
                    }
                }
                if (hi >= ntw) break;
                lds_barrier();                                         // everyone is done reading blocks <= hi
                if (wave == NWV - 1) {
                    mt_twist_block_wave(mt + (hi & 7) * 624, mt + ((hi + 1) & 7) * 624, lane);
                    if (hi + 2 <= ntw) mt_twist_block_wave(mt + ((hi + 1) & 7) * 624, mt + ((hi + 2) & 7) * 624, lane);
                }
                lo = hi + 1; hi = min(ntw, hi + 2);
                lds_barrier();
            }
        }
        lds_barrier();                                                 // keys final
        DBG_MARK(13);
        if (tid < BW) {        // final spikes: the winner's bit, or nothing
            uint32_t wbits = 0;
            if ((anym >> wb) & 1u) {
                const int win = (int)(0xFFFFFFFFu - (uint32_t)(keys[wb] & 0xFFFFFFFFull));
                if ((win >> 5) == wj) wbits = 1u << (win & 31);
            }
            finE[tid] = wbits;
        }
        if (g == 0) {   // publish the advanced generator for the next launch
            snn_rng_state *wr = c.rng[pcur];
            const uint32_t *fin = mt + (ntw & 7) * 624;
            if (tid < 624) wr->mt[tid] = fin[tid];
            if (tid == 0) {
                wr->pos = E - 624 * ntw;
                wr->consumed = rng_consumed + (long long)rows * N;
            }
        }
    } else if (phaseA) {
        if (tid < BW) finE[tid] = crs[tid];
    }
    lds_barrier();
    DBG_MARK(14);
    for (int b = wave; b < B; b += NT / 64) {          // event lists of the final Ae spikes
        const int ne = build_list(finE + b * NW, NW, lane, lstE + b * LR, LR);
        if (lane == 0) { cntE[b] = ne; if (ne > 4) atomicOr((unsigned int *)&misc[2], 2u); }
    }
    DBG_MARK(2);
    if (phaseA) {
        // ---- A2: final Ae spikes of step t-1 for the own tile: trace, raster, layer.s
        if (tid < TT && bl < B) {
            const bool sp = colv && bit_of(finE + bl * NW, j);
            float xn = 0.f;
            if (colv) {
                if (c.pE.lif.traces) {
                    xn = trace_next(r_xE, sp, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive);
                    c.xE[kst] = xn;
                }
                c.sE[kst] = sp;
                if (c.rasE) c.rasE[(size_t)(t - 1) * B * N + kst] = sp;
            }
            xnu0[bl * CW + jj] = xn * c.nu0;                           // target_x * nu[0] (MCC_learning.py:235)
            if (sp) atomicOr(&colmask[jj], 1u << bl);
        }
    }
    lds_barrier();
    // recurrent Ae->Ai weights: issued now, consumed in phase B
    float we[4] = {0.f, 0.f, 0.f, 0.f};
    if (cvalid) {
        const int nE = cntE[cb_];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < nE) we[u] = c.Wei[(int)lstE[cb_ * LR + u] * N + cjg];
    }
    bool tile_fresh = false;       // the STDP pass below left the active rows of the own slice in `wtile`
    bool tile_full = false;
    if (phaseA) {
        DBG_MARK(3);
        // ---- A3: PostPre on the own [Nin x CW] weight slice
        if (do_stdp) {
            const bool full = stdp_full;
            tile_fresh = true; tile_full = full;
            const int Etot = Nin * N, Emain = (Etot / 32) * 32;
            const bool anytail = Etot != Emain;
            const float *xsrc = c.xX[pprev];
            if (full) {
                if (anytail) stdp_rows<OuterSum, true>(c, Nin, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain, pre_w);
                else stdp_rows<CascadeT, true>(c, Nin, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain, pre_w);
            } else {
                uint32_t acols = 0;                     // own columns with a post-synaptic spike
                if (c.nu1 != 0.f) {
#pragma unroll
                    for (int q = 0; q < CW; ++q) acols |= (colmask[q] != 0 ? 1u : 0u) << q;
                }
                if (anytail) {
                    stdp_rows<OuterSum, false>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain, pre_w);
                    stdp_cols<OuterSum>(c, acols, rowmask, colmask, xsrc, c0, tid, Emain);
                } else {
                    stdp_rows<CascadeT, false>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain, pre_w);
                    stdp_cols<CascadeT>(c, acols, rowmask, colmask, xsrc, c0, tid, Emain);
                }
            }
        }
    }
    const bool busy = (misc[2] & 2) != 0;              // uniform: some sample overflowed the fixed-size fast path
    if (busy) __syncthreads();     // generic path re-reads the own weight slice from global: drain the STDP stores
    else lds_barrier();            // fast path reads the refreshed weights from the LDS tile
    DBG_MARK(4);
    if (!phaseB) return;

    // ================================================================== phase B: start step t
    // ---- B1: input currents of the own tile from step t-1 spikes
    float curE = 0.f, curI = 0.f;
    if (!busy && tailcol) {
        // row_sum columns: quad of threads per (sample, column)
        if (cvalid) {
            const int nX = cntX[cb_], nI = cntI[cb_], nE = cntE[cb_];
            const uint16_t *lx = lstX + cb_ * LX;
            const uint8_t *xb = sbytes ? sbytes + cb_ * Nin : nullptr;
            int ix[16], ii[4], ie[4];
            float wx[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) ix[u] = min((int)lx[u], Nin - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) { ii[u] = min((int)lstI[cb_ * LR + u], N - 1); ie[u] = min((int)lstE[cb_ * LR + u], N - 1); }
            if (tile_fresh) {
                int rr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) rr[u] = tile_full ? ix[u] : min((int)rowpos[ix[u]], Nin - 1);
#pragma unroll
                for (int u = 0; u < 16; ++u) wx[u] = wtile[rr[u] * CW + cj_];
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) wx[u] = c.Wxe[ix[u] * N + cjg];
            }
            const float e1 = quad_lane_sum<16>(ix, nX, wx, xb, Nin, cL);
            const float e2 = quad_lane_sum<4>(ii, nI, wi, nullptr, N, cL);
            const float e3 = quad_lane_sum<4>(ie, nE, we, nullptr, N, cL);
            if (cL == 0) {
                curbuf[(cb_ * CW + cj_) * 2] = (0.0f + e1) + e2;        // (zeros + X->Ae) + Ai->Ae
                curbuf[(cb_ * CW + cj_) * 2 + 1] = 0.0f + e3;            // zeros + Ae->Ai
            }
        }
        lds_barrier();
        if (mine) { curE = curbuf[(bl * CW + jj) * 2]; curI = curbuf[(bl * CW + jj) * 2 + 1]; }
    } else if (mine) {
        const int nX = cntX[bl], nI = cntI[bl], nE = cntE[bl];
        const uint8_t *xb = sbytes ? sbytes + bl * Nin : nullptr;
        if (!busy) {
            const float *wt = tile_fresh ? wtile : nullptr;
            const uint16_t *rp = tile_full ? nullptr : rowpos;
            tile_currents<CascadeFlat>(c, lstX + bl * LX, nX, lstI + bl * LR, nI, lstE + bl * LR, nE, wi, we, wt, rp, jj, xb, j, curE, curI);
        } else {   // generic bit-scan path
            const uint32_t *xw = c.dig + (size_t)t * c.DW + c.OXW + bl * NinW, *iw = spI + bl * NW, *ew = finE + bl * NW;
            const uint64_t ax = ~0ull >> (64 - NinW), ar = ~0ull >> (64 - NW);
            if (tailcol) {
                curE = 0.0f + ordered_dot<RowSum4>(c.Wxe, N, j, xw, ax, xb, Nin);
                curE = curE + ordered_dot<RowSum4>(c.Wie, N, j, iw, ar, nullptr, N);
                curI = 0.0f + ordered_dot<RowSum4>(c.Wei, N, j, ew, ar, nullptr, N);
            } else {
                curE = 0.0f + ordered_dot<CascadeN>(c.Wxe, N, j, xw, ax, xb, Nin);
                curE = curE + ordered_dot<CascadeN>(c.Wie, N, j, iw, ar, nullptr, N);
                curI = 0.0f + ordered_dot<CascadeN>(c.Wei, N, j, ew, ar, nullptr, N);
            }
        }
    }
    DBG_MARK(5);
    // ---- B2: membrane updates
    bool spE = false, spIn = false;
    float o_vE = 0.f, o_rE = 0.f, o_vI = 0.f, o_rI = 0.f, th = 0.f;
    if (mine) {
        th = r_theta;
        if (c.pE.learning) th = th * c.pE.theta_decay;                 // nodes.py:1079
        o_vE = r_vE; o_rE = r_rE;
        spE = dc_update(o_vE, o_rE, curE, c.pE.lif.thresh + th, c.pE.lif);   // nodes.py:1088
        if (spE) atomicAdd(&cnt[jj], 1);
        o_vI = r_vI; o_rI = r_rI;
        float ci = curI;
        if (o_rI > 0.f) ci = 0.f;                                      // nodes.py:511
        spIn = lif_update(o_vI, o_rI, ci, c.pI);
    }
    lds_barrier();
    if (mine) {
        if (bl == 0) {                                                 // one thread per column owns theta
            if (c.pE.learning) th = th + c.pE.theta_plus * (float)cnt[jj];   // nodes.py:1094
            c.theta[j] = th;
        }
        c.vE[kst] = o_vE; c.rE[kst] = o_rE;
        if (c.rasVE) c.rasVE[(size_t)t * B * N + kst] = o_vE;
        c.vI[kst] = o_vI; c.rI[kst] = o_rI;
        c.sI[kst] = spIn;
        if (c.pI.traces) c.xI[kst] = trace_next(r_xI, spIn, c.pI.trace_decay, c.pI.trace_scale, c.pI.traces_additive);
        if (c.rasI) c.rasI[(size_t)t * B * N + kst] = spIn;
        if (c.rasVI) c.rasVI[(size_t)t * B * N + kst] = o_vI;
    }
    {   // publish crossing / spike bits: a wave holds 64/CW samples x CW columns -> one byte per sample
        const uint64_t mE = __ballot(spE), mI = __ballot(spIn);
        constexpr int SPW = 64 / CW;                               // samples per wave
        const int sidx = lane / CW, b = wave * SPW + sidx;
        if (tid < TT && (lane % CW) == 0 && b < B) {
            ((uint8_t *)c.crossE[pcur])[(b * NW) * 4 + g] = (uint8_t)(mE >> (sidx * CW));
            ((uint8_t *)c.spikeI[pcur])[(b * NW) * 4 + g] = (uint8_t)(mI >> (sidx * CW));
        }
    }
    DBG_MARK(6);
    // ---- B3: X trace of step t for an own slice of input rows (nodes.py:96-103)
    if (c.x_traces) {
        float *xn = c.xX[pcur];
        if (xmine) xn[xb_ * Nin + xi_] = trace_next(r_xo, r_xs, c.x_decay, c.x_scale, c.x_additive);
        for (int item = tid + NT; item < B * c.RS; item += NT) {        // only when the slice has > NT items
            const int b = item / c.RS, i = xr0 + (item - b * c.RS);
            if (i < xr1) {
                const int k = b * Nin + i;
                xn[k] = trace_next(c.xX[pprev][k], c.in[stepoff + k], c.x_decay, c.x_scale, c.x_additive);
            }
        }
    }
    DBG_MARK(7);
    if (c.dbg && blockIdx.x == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + 9] = (long long)clock64();
    if (c.dbg && threadIdx.x == 0) atomicMax((unsigned long long *)&c.dbg[(size_t)t * 24 + 21], (unsigned long long)wall_clock64());
}


// =====================================================================================================
// Resident plan "dc2015-resident": the SAME decomposition and arithmetic as k_dc2015_step, but ONE launch for
// the whole run.  What used to cross the kernel boundary now stays put or moves through tagged granules:
//   * the [Nin x CW] weight slice lives in LDS for all T steps, membrane state / traces / theta in registers;
//   * the X trace of every step is precomputed (k_dc2015_xtrace) -- it depends on the inputs alone;
//   * each workgroup keeps its own copy of the generator (all copies advance identically);
//   * the per-step spike exchange uses 8-byte {epoch, bits} granules written with ONE relaxed agent-scope
//     (write-through) store and polled with relaxed agent-scope loads: the data is the flag, no fence
//     (cdna_hip_programming.md Guideline 16, form R2).  Granule k of workgroup g carries samples 2k, 2k+1:
//     crossing byte | Ai spike byte << 8 | (same for the odd sample) << 16.  Two buffers by epoch parity: a
//     workgroup overwrites a buffer only after every other workgroup has published the epoch in between,
//     which it does only after consuming the overwritten one.
// All G <= 128 workgroups are co-resident (one 1024-thread workgroup per CU), polls are bounded (status word).
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

// X trace after every step: entry 0 = trace at run entry, entry e = trace after step e-1 (nodes.py:96-103).
__global__ __launch_bounds__(256) void k_dc2015_xtrace(const DcCtx c) {
    const int n = c.B * c.Nin;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float x = c.xX[1][k];
    c.xtr[k] = x;
    int t = 0;
    for (; t + 8 <= c.T; t += 8) {            // the spike loads do not depend on x: issue them together
        uint8_t s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = c.in[(size_t)(t + u) * n + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x = trace_next(x, s[u], c.x_decay, c.x_scale, c.x_additive); c.xtr[(size_t)(t + u + 1) * n + k] = x; }
    }
    for (; t < c.T; ++t) { x = trace_next(x, c.in[(size_t)t * n + k], c.x_decay, c.x_scale, c.x_additive); c.xtr[(size_t)(t + 1) * n + k] = x; }
    c.xX[1][k] = x;
}

// ---- cold paths of the resident kernel, kept out of line so that their register needs do not spill the
//      per-step state of the hot path (LDS pointers arrive as generic pointers: slower, and irrelevant here)
struct Cur2 { float e, i; };

// Input currents by bit-scan over the spike words (a sample overflowed the fixed-size event lists).
__device__ __attribute__((noinline)) Cur2 busy_currents(const float *wtile, const float *wieT, const float *weiT,
                                                        const uint32_t *xw, const uint32_t *iw, const uint32_t *ew,
                                                        const uint8_t *xb, int NinW, int NW, int Nin, int N, int jj, bool tail,
                                                        int CW) {
    const uint64_t ax = ~0ull >> (64 - NinW), ar = ~0ull >> (64 - NW);
    Cur2 r;
    if (tail) {
        r.e = 0.0f + ordered_dot<RowSum4>(wtile, CW, jj, xw, ax, xb, Nin);
        r.e = r.e + ordered_dot<RowSum4>(wieT, CW, jj, iw, ar, nullptr, N);
        r.i = 0.0f + ordered_dot<RowSum4>(weiT, CW, jj, ew, ar, nullptr, N);
    } else {
        r.e = 0.0f + ordered_dot<CascadeN>(wtile, CW, jj, xw, ax, xb, Nin);
        r.e = r.e + ordered_dot<CascadeN>(wieT, CW, jj, iw, ar, nullptr, N);
        r.i = 0.0f + ordered_dot<CascadeN>(weiT, CW, jj, ew, ar, nullptr, N);
    }
    return r;
}

// one_spike arbitration when the step needs more generator blocks than the ring holds or has more candidates
// than the compact list: walk the stream block by block.  Blocks 0..min(ntw,2) (relative to slot mb) are resident
// on entry; everything beyond is (re)computed on the way.  Every thread of the workgroup must call it.
__device__ __attribute__((noinline)) void arbitrate_slow(uint32_t *mt, const uint32_t *crs, unsigned long long *keys,
                                                         int mb, int pos, int N, int ntw, int rows, int myrank,
                                                         int wb, int wj, int BW, int tid, int nthreads) {
    const int lane = tid & 63, wave = tid >> 6;
    const int NWV = nthreads / 64;
    uint32_t parked = 0;
    int lo = 0, hi = min(ntw, 2);
    while (rows) {
        if (tid < BW) {
            uint32_t bits = crs[tid];
            while (bits) {
                const int jx = wj * 32 + __ffs(bits) - 1; bits &= bits - 1;
                const int d = myrank * N + jx;
                const int w0 = pos + 2 * d, w1 = w0 + 1;
                const int m0 = w0 / 624, m1 = w1 / 624;
                float q; bool have = false;
                if (m0 >= lo && m1 <= hi) {
                    q = exp1_from_words(mt_temper(mt[((mb + m0) & 7) * 624 + w0 - 624 * m0]),
                                        mt_temper(mt[((mb + m1) & 7) * 624 + w1 - 624 * m1])); have = true;
                } else if (m0 >= lo && m0 <= hi) {             // pair straddles the resident range: park the high word
                    parked = mt_temper(mt[((mb + m0) & 7) * 624 + w0 - 624 * m0]);
                } else if (m1 >= lo && m1 <= hi) {
                    q = exp1_from_words(parked, mt_temper(mt[((mb + m1) & 7) * 624 + w1 - 624 * m1])); have = true;
                }
                if (have) {
                    const float val = 1.0f / q;
                    const unsigned long long key =
                        ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx);
                    atomicMax(&keys[wb], key);
                }
            }
        }
        if (hi >= ntw) break;
        lds_barrier();
        if (wave == NWV - 1) {
            mt_twist_block_wave(mt + ((mb + hi) & 7) * 624, mt + ((mb + hi + 1) & 7) * 624, lane);
            if (hi + 2 <= ntw) mt_twist_block_wave(mt + ((mb + hi + 1) & 7) * 624, mt + ((mb + hi + 2) & 7) * 624, lane);
        }
        lo = hi + 1; hi = min(ntw, hi + 2);
        lds_barrier();
    }
}

constexpr unsigned kPollLimit = 400000u;
constexpr int kResidentDefaultNT = 1024;
constexpr int kResidentDefaultCW = 4;   // measured at cfg2: 8 -> 93.8 k, 4 -> 98.2 k, 2 -> 96.6 k timesteps/s (same GPU box)
constexpr int kBitWords = 1024;      // capacity of the [B][NW] bit-word arrays (independent of the workgroup size)
constexpr size_t resident_fixed_lds(int cw) {
    return 4 * kBitWords * 4 + MAXB * cw * 4 + 8 * 624 * 4 + NCAND * 4 + MAXB * 8 + 2 * MAXB * LR * 2 + 2 * MAXB * 4 + 2 * 32 * 4 + 32 +
           2 * MAXB * cw * 4 + 7 * MAXB * cw * 4;
}     // bounded spin: ~0.5 s, then the run is flagged SNN_ERR_TIMEOUT

// CWR = columns per workgroup (8, 4 or 2): the PostPre stage is ALU-throughput bound inside a CU, so narrower tiles on
// more CUs shorten it, while the stages every workgroup repeats (receive, lists, arbitration) stay as they are.
// NTR = threads per workgroup: 1024, or 512 (twice the registers per thread: no spills; needs B*NW <= 512 and CW <= 4).
template <int CWR, int NTR>
__global__ __launch_bounds__(NTR) void k_dc2015_run(const DcCtx c) {
    constexpr int CW = CWR, TT = MAXB * CWR, NT = NTR;     // (shadow the per-step kernel's constants)
    constexpr int SPG = 16 / CW;                           // exchange: samples per granule (CW crossing bits + CW Ai-spike bits each)
    constexpr int WPB = 8 / CW;                            //           workgroups sharing one byte of a sample's bit string
    constexpr uint32_t FM = (1u << CW) - 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, N = c.N, NW = c.NW, NinW = c.NinW, T = c.T;
    // ---- LDS carve-up.  Everything of fixed size sits at a compile-time offset (addresses fold into the
    //      instructions' immediate offsets instead of occupying registers); the four size-dependent arrays follow.
    constexpr size_t O_CRS = 0, O_FINE = O_CRS + kBitWords * 4, O_SPI = O_FINE + kBitWords * 4, O_XNU0 = O_SPI + 2 * kBitWords * 4,
                     O_MT = O_XNU0 + MAXB * CW * 4, O_CAND = O_MT + 8 * 624 * 4, O_KEYS = O_CAND + NCAND * 4,
                     O_LSTI = O_KEYS + MAXB * 8, O_LSTE = O_LSTI + MAXB * LR * 2, O_CNTI = O_LSTE + MAXB * LR * 2,
                     O_CNTE = O_CNTI + MAXB * 4, O_CNT = O_CNTE + MAXB * 4, O_COLM = O_CNT + 32 * 4, O_MISC = O_COLM + 32 * 4,
                     O_CURB = O_MISC + 32, O_ST = O_CURB + 2 * MAXB * CW * 4, O_WT = O_ST + 7 * MAXB * CW * 4;
    static_assert(O_WT == resident_fixed_lds(CW) && O_WT % 16 == 0, "fixed LDS part");
    uint32_t *crs = (uint32_t *)(smem + O_CRS);            // [B][NW] Ae crossings of step t-1 (B * NW <= NT)
    uint32_t *finE = (uint32_t *)(smem + O_FINE);          // ... final Ae spikes
    uint32_t *spI2 = (uint32_t *)(smem + O_SPI);           // ... Ai spikes, two buffers by step parity (the raster rows of
                                                           //     one step are written while the next receive may already run)
    float *xnu0 = (float *)(smem + O_XNU0);
    uint32_t *mt = (uint32_t *)(smem + O_MT);              // generator ring: block base+m in slot (mb + m) & 7
    unsigned long long *keys = (unsigned long long *)(smem + O_KEYS);
    uint16_t *lstI = (uint16_t *)(smem + O_LSTI);
    uint16_t *lstE = (uint16_t *)(smem + O_LSTE);
    int *cntI = (int *)(smem + O_CNTI);
    int *cntE = (int *)(smem + O_CNTE);
    int *cnt = (int *)(smem + O_CNT);
    uint32_t *colmask = (uint32_t *)(smem + O_COLM);
    int *misc = (int *)(smem + O_MISC);
    float *curbuf = (float *)(smem + O_CURB);
    // membrane state of the tile threads' (sample, column) pairs: [7][TT] = vE, rE, vI, rI, xE, xI, theta.  Each slot is
    // touched by its own thread only; it lives in LDS rather than in registers because seven values that are used once
    // per step are exactly what the register allocator spills to (much slower) scratch memory in this kernel.
    float *stl = (float *)(smem + O_ST);
    float *wtile = (float *)(smem + O_WT);                 // [Nin][CW] the own weight slice, resident for the run
    float *wieT = wtile + (size_t)Nin * CW;                // [N][CW] own column slices of the recurrent weights
    float *weiT = wieT + (size_t)N * CW;
    const int DGS = (c.DGW + 63) & ~63;                    // digest buffer stride (words): whole wave chunks
    uint32_t *dgbuf = (uint32_t *)(weiT + (size_t)N * CW); // digests of iteration t (buffer t & 1) and t + 1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, c0 = g * CW;
    const int jj = tid % CW, bl = tid / CW;
    const int j = c0 + jj;
    const bool colv = j < N;
    const bool tailcol = c0 >= (N / 32) * 32;
    const int BW = B * NW;
    const bool mine = tid < TT && bl < B && colv;
    const unsigned kst = (unsigned)(bl * N + j);
    const int wb = (int)(((float)tid + 0.5f) * c.inv_NW), wj = tid - wb * NW;   // exchange word (sample wb, word wj)
    const int KB = c.KB, NG = c.G * KB;                    // granules per epoch
    const int Etot = Nin * N, Emain = (Etot / 32) * 32;
    const bool anytail = Etot != Emain;
    constexpr int NWV = NT / 64;

    // ---- one-time staging: weight slice, state, generator, zeroed scratch
    for (int k = tid; k < Nin * CW; k += NT) {
        const int i = k / CW, q = k % CW;
        wtile[k] = (c0 + q < N) ? c.Wxe[i * N + c0 + q] : 0.f;
    }
    for (int k = tid; k < N * CW; k += NT) {
        const int i = k / CW, q = k % CW;
        wieT[k] = (c0 + q < N) ? c.Wie[i * N + c0 + q] : 0.f;
        weiT[k] = (c0 + q < N) ? c.Wei[i * N + c0 + q] : 0.f;
    }
    // digest of iteration e straight into LDS (global_load_lds: no register round trip); wave-uniform 256-byte chunks
    auto fetch_digest = [&](int e) {
        const uint32_t *Dg = c.dig + (size_t)e * c.DW;     // 16-byte aligned (DW % 4 == 0), LDS part first
        uint32_t *dst = dgbuf + (e & 1) * DGS;
        for (int base = wave * 256; base < c.DGW; base += (NT / 64) * 256) {
            const int ub = __builtin_amdgcn_readfirstlane(base);
            if (ub + lane * 4 < c.DGW)                     // DGW % 4 == 0: a lane's four words are all in or all out
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Dg + ub + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + ub), 16, 0, 0);
        }
    };
    fetch_digest(0);
    bool last_sE = false, last_sI = false;
    if (mine) {
        stl[0 * TT + tid] = c.vE[kst]; stl[1 * TT + tid] = c.rE[kst]; stl[2 * TT + tid] = c.vI[kst]; stl[3 * TT + tid] = c.rI[kst];
        stl[4 * TT + tid] = c.pE.lif.traces ? c.xE[kst] : 0.f;
        stl[5 * TT + tid] = c.pI.traces ? c.xI[kst] : 0.f;
        stl[6 * TT + tid] = c.theta[j];
        last_sE = c.sE[kst] != 0; last_sI = c.sI[kst] != 0;
    }
    int rng_pos = 0, mb = 0, ahead = 0; long long rng_consumed = 0;
    if (c.pE.one_spike) {
        for (int k = tid; k < 624; k += NT) mt[k] = c.rng[0]->mt[k];
        rng_pos = __builtin_amdgcn_readfirstlane(c.rng[0]->pos);
        const long long cons0 = c.rng[0]->consumed;
        rng_consumed = ((long long)__builtin_amdgcn_readfirstlane((int)(cons0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)cons0);
    }
    if (tid < 32) { cnt[tid] = 0; colmask[tid] = 0; }
    if (tid < 8) misc[tid] = 0;
    if (tid < MAXB) keys[tid] = 0ull;
    if (tid < BW) { crs[tid] = 0; finE[tid] = 0; spI2[tid] = 0; spI2[kBitWords + tid] = 0; }
    bool failed = false;
    const bool early_fetch = tailcol || (Nin <= 1024 && NT - TT >= B * CW * 4);   // = the currents stage always has that barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t <= T; ++t) {
        const bool phaseA = t >= 1, phaseB = t < T;
        DBG_MARK(0);
        if (c.dbg && blockIdx.x == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + 8] = (long long)clock64();
        if (c.dbg && threadIdx.x == 0) atomicMin((unsigned long long *)&c.dbg[(size_t)t * 24 + 20], (unsigned long long)wall_clock64());
        const int stepoff = t * B * Nin;
        uint32_t *spI = spI2 + (t & 1) * kBitWords;
        const uint32_t *dg = dgbuf + (t & 1) * DGS;                       // digest of the X spikes of step t-1
        const uint16_t *lstX = (const uint16_t *)dg;
        const int *meta = (const int *)(dg + B * (LX / 2));
        const int *cntX = meta;
        const uint32_t *rowmask = dg + B * (LX / 2) + 40;
        const uint16_t *arows = (const uint16_t *)(rowmask + Nin);
        const uint16_t *lst2 = arows + 4 * ((Nin + 1) / 2);              // X events grouped by row_sum lane
        const uint32_t *gcnt = (const uint32_t *)(lst2 + B * LX);
        const uint32_t *gqn = gcnt + B;                                   // X events per 256-position cascade group
        const uint8_t *sprev_g = (t == 0) ? c.sX0 : c.in + (stepoff - B * Nin);
        // ------------------------------------------------------------------ receive step t-1
        const bool use_rng = phaseA && c.pE.one_spike;
        if (use_rng) {
            // while the other waves wait for the exchange, the LAST wave runs the generator ahead (lockstep twists,
            // no barrier) until the ring is full: blocks base+1 .. base+7.  A step consumes 2 * N words per sample
            // with a crossing, i.e. a few blocks, so the arbitration below finds its blocks already there.
            if (wave == NWV - 1)
                for (int m = ahead; m < 7; ++m) mt_twist_block_wave(mt + ((mb + m) & 7) * 624, mt + ((mb + m + 1) & 7) * 624, lane);
            ahead = 7;
        }
        // next iteration's digest, issued before the wait for the exchange when nothing can still be reading the buffer
        // it overwrites (the previous iteration's currents stage ended with a barrier behind its last digest read)
        if (t < T && early_fetch) fetch_digest(t + 1);
        if (phaseA) {
            const unsigned long long *exr = c.ex + (size_t)(t & 1) * NG;
            const int NH = (c.G + WPB - 1) / WPB;           // bytes per sample = groups of WPB workgroups
            for (int it = tid; it < NH * KB; it += NT) {
                const int h = it / KB, k = it - h * KB;
                unsigned long long x[WPB];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int w = 0; w < WPB; ++w) {
                        const int gi = h * WPB + w;
                        x[w] = gi < c.G ? granule_load(exr + gi * KB + k) : ((unsigned long long)(uint32_t)t << 32);
                        ok = ok && (uint32_t)(x[w] >> 32) == (uint32_t)t;
                    }
                    if (ok || failed) break;
                    if (++spins > kPollLimit) { failed = true; if (c.status) __hip_atomic_store(c.status, (int)SNN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int sidx = 0; sidx < SPG; ++sidx) {
                    const int b = k * SPG + sidx;
                    if (b >= B) break;
                    uint32_t be = 0, bi = 0;
#pragma unroll
                    for (int w = 0; w < WPB; ++w) {      // granule payload: crossings of its SPG samples | << 16: their Ai spikes
                        be |= (((uint32_t)x[w] >> (sidx * CW)) & FM) << (w * CW);
                        bi |= (((uint32_t)x[w] >> (16 + sidx * CW)) & FM) << (w * CW);
                    }
                    ((uint8_t *)crs)[(b * NW) * 4 + h] = (uint8_t)be;
                    ((uint8_t *)spI)[(b * NW) * 4 + h] = (uint8_t)bi;
                    if (be && c.pE.one_spike) atomicOr((unsigned int *)&misc[3], 1u << b);   // samples with an Ae crossing: known at the barrier below
                }
            }
        } else if (tid < BW) {   // t == 0: previous spikes come from the layers' `s` tensors (bytes -> bits)
            uint32_t me = 0, mi = 0;
            for (int qq = 0; qq < 32; ++qq) {
                const int jx = wj * 32 + qq;
                if (jx < N) { me |= (uint32_t)(c.sE[wb * N + jx] != 0) << qq; mi |= (uint32_t)(c.sI[wb * N + jx] != 0) << qq; }
            }
            finE[tid] = me; spI[tid] = mi;
        }
        DBG_MARK(10);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this iteration's digest (issued one iteration ago) has landed
        DBG_MARK(11);
        lds_barrier();
        DBG_MARK(15);
        if (tid < CW) cnt[(t & 1) * CW + tid] = 0;         // this step's spike counts (buffer last read one iteration ago)
        if (t < T && !early_fetch) fetch_digest(t + 1);    // next iteration's digest: in flight behind this one
        DBG_MARK(16);
        const int mflags = __builtin_amdgcn_readfirstlane(meta[33]);
        const uint8_t *sbytes = (mflags & 1) ? sprev_g : nullptr;
        if (tid == 0 && (mflags & 2)) atomicOr((unsigned int *)&misc[2], 2u);
        const bool do_stdp = phaseA && c.learning && c.rule == SNN_RULE_POSTPRE;
        const bool stdp_full = t == 1;
        const int nact = stdp_full ? Nin : __builtin_amdgcn_readfirstlane(meta[32]);
        // ---- per sample (one wave each, in turns): event list of its Ai spikes; does it have an Ae crossing?
        if (NW <= 32) {                                    // two samples per wave, one per half
            const int hl = lane & 31;
            for (int b2 = wave * 2; b2 < B; b2 += NWV * 2) {
                const int b = b2 + (lane >> 5);
                const bool bv = b < B;
                const int ni = build_list_half(bv ? spI + b * NW : nullptr, NW, lane, lstI + (bv ? b : 0) * LR, LR);
                if (hl == 0 && bv) {
                    cntI[b] = ni;
                    if (ni > 4) atomicOr((unsigned int *)&misc[2], 2u);
                }
            }
        } else {
            for (int b = wave; b < B; b += NWV) {
                const int ni = build_list(spI + b * NW, NW, lane, lstI + b * LR, LR);
                if (lane == 0) {
                    cntI[b] = ni;
                    if (ni > 4) atomicOr((unsigned int *)&misc[2], 2u);
                }
            }
        }
        // ---- one_spike arbitration, identical in every workgroup (nodes.py:1097-1105): among the crossings of a sample
        //      the winner is argmax(1 / q[j]), q = the generator's next Exp(1) draws, one per neuron of every sample that
        //      has a crossing, in sample order.  Which samples crossed is known since the receive barrier, so each thread
        //      scores the crossings of ITS (sample, word) right here, next to the list building: the barrier below
        //      closes both.
        uint32_t anym = 0;
        int arb_rows = 0, arb_E = 0, arb_ntw = 0;
        if (use_rng) {
            anym = (uint32_t)__builtin_amdgcn_readfirstlane(misc[3]);
            arb_rows = __popc(anym);
            arb_E = rng_pos + 2 * arb_rows * N;
            arb_ntw = arb_rows ? (arb_E - 1) / 624 : 0;
            if (arb_ntw <= 7 && tid < BW) {            // every block the step consumes is resident (ring run ahead at the top)
                uint32_t bits = crs[tid];
                const int myrank = __popc(anym & ((1u << (wb & 31)) - 1u));
                while (bits) {
                    const int jx = wj * 32 + __ffs(bits) - 1; bits &= bits - 1;
                    const int d = myrank * N + jx;
                    const int w0 = rng_pos + 2 * d, w1 = w0 + 1;
                    const int m0 = w0 / 624, m1 = w1 / 624;
                    const float q = exp1_from_words(mt_temper(mt[((mb + m0) & 7) * 624 + w0 - 624 * m0]),
                                                    mt_temper(mt[((mb + m1) & 7) * 624 + w1 - 624 * m1]));
                    const float val = 1.0f / q;                             // p / q with p = 1
                    const unsigned long long key =                          // max value, ties -> lowest index
                        ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx);
                    atomicMax(&keys[wb], key);
                }
            }
        }
        DBG_MARK(1);
        lds_barrier();
        if (tid == 35) misc[3] = 0;                        // crossing-sample mask: every thread has read it; next set by the next receive
        const int cb_ = tailcol ? tid / (CW * 4) : bl, cj_ = tailcol ? (tid >> 2) % CW : jj, cL = tid & 3;
        const int cjg = c0 + cj_;
        const bool cvalid = phaseB && cb_ < B && cjg < N && (tailcol || tid < TT);

        // ================================================================== phase A: finish step t-1
        if (use_rng) {
            DBG_MARK(12);
            const int rows = arb_rows, pos = rng_pos, E = arb_E, ntw = arb_ntw;
            if (ntw > 7) {
                // more generator blocks than the ring holds: walk the stream block by block (out of line)
                const int myrank = __popc(anym & ((1u << (wb & 31)) - 1u));
                arbitrate_slow(mt, crs, keys, mb, pos, N, ntw, rows, myrank, wb, wj, BW, tid, NT);
                ahead = ntw;            // blocks beyond ntw may have been overwritten by the walk
                lds_barrier();
            }
            DBG_MARK(13);
            if (tid < BW) {        // final spikes: the winner's bit, or nothing -- and with them the event lists (<= 1 entry)
                uint32_t wbits = 0;
                if ((anym >> wb) & 1u) {
                    const int win = (int)(0xFFFFFFFFu - (uint32_t)(keys[wb] & 0xFFFFFFFFull));
                    if ((win >> 5) == wj) { wbits = 1u << (win & 31); lstE[wb * LR] = (uint16_t)win; }
                }
                finE[tid] = wbits;
            }
            if (tid < B) cntE[tid] = (int)((anym >> tid) & 1u);
            mb = (mb + ntw) & 7; ahead -= ntw;
            rng_pos = E - 624 * ntw;
            rng_consumed += (long long)rows * N;
            // (no barrier: the trace stage below takes the winners straight from `keys`; finE / lstE / cntE are for
            //  the stages behind its barrier)
        } else {
            if (phaseA && tid < BW) finE[tid] = crs[tid];
            lds_barrier();
            for (int b = wave; b < B; b += NT / 64) {      // event lists of the final Ae spikes
                const int ne = build_list(finE + b * NW, NW, lane, lstE + b * LR, LR);
                if (lane == 0) { cntE[b] = ne; if (ne > 4) atomicOr((unsigned int *)&misc[2], 2u); }
            }
        }
        DBG_MARK(14);
        DBG_MARK(2);
        if (phaseA) {
            if (tid < TT && bl < B) {
                bool sp;
                if (use_rng) sp = colv && ((anym >> bl) & 1u) && (int)(0xFFFFFFFFu - (uint32_t)(keys[bl] & 0xFFFFFFFFull)) == j;
                else sp = colv && bit_of(finE + bl * NW, j);
                float xn = 0.f;
                if (colv) {
                    if (c.pE.lif.traces) { xn = trace_next(stl[4 * TT + tid], sp, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive); stl[4 * TT + tid] = xn; }
                    last_sE = sp;
                }
                xnu0[bl * CW + jj] = xn * c.nu0;
                if (sp) atomicOr(&colmask[jj], 1u << bl);
            }
        }
        lds_barrier();
        if (phaseA) {
            DBG_MARK(3);
            if (do_stdp) {
                const float *xsrc = c.xtr + (size_t)t * B * Nin;          // X trace after step t-1
                if (stdp_full) {
                    if (anytail) stdp_rows_lds<OuterSum, true, CW, NT>(c, Nin, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                    else stdp_rows_lds<CascadeT, true, CW, NT>(c, Nin, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                } else {
                    uint32_t acols = 0;
                    if (c.nu1 != 0.f) {
#pragma unroll
                        for (int q = 0; q < CW; ++q) acols |= (colmask[q] != 0 ? 1u : 0u) << q;
                    }
                    if (anytail) {
                        stdp_rows_lds<OuterSum, false, CW, NT>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                        stdp_cols_lds<OuterSum, CW, NT>(c, acols, rowmask, colmask, xsrc, wtile, c0, tid, Emain);
                    } else {
                        stdp_rows_lds<CascadeT, false, CW, NT>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                        stdp_cols_lds<CascadeT, CW, NT>(c, acols, rowmask, colmask, xsrc, wtile, c0, tid, Emain);
                    }
                }
            }
        }
        const bool busy = (__builtin_amdgcn_readfirstlane(misc[2]) & 2) != 0;
        lds_barrier();
        DBG_MARK(4);
        auto write_raster_rows = [&]() {
            if (phaseA && tid >= TT) {
            // spike rasters of step t-1: every workgroup holds the complete bit strings of the step (final Ae spikes,
            // received Ai spikes), so whole [N]-byte rows are written by ONE workgroup each (row r of the 2*B rows by
            // workgroup r mod G) instead of CW-byte pieces by all of them -- full coalesced lines instead of partial
            // sectors -- and by the threads that have nothing to do while the tile threads compute currents
            for (int r = g; r < 2 * B; r += c.G) {
                const int b = r < B ? r : r - B;
                uint8_t *ras = r < B ? c.rasE : c.rasI;
                const uint32_t *bitsrc = (r < B ? finE : spI) + b * NW;
                if (ras) { uint8_t *row = ras + ((size_t)(t - 1) * B + b) * N; for (int jx = tid - TT; jx < N; jx += NT - TT) row[jx] = (uint8_t)bit_of(bitsrc, jx); }
            }
            }
        };
        if (!phaseB) write_raster_rows();                  // (last iteration: nothing to overlap with)
        if (!phaseB) break;
        // scratch of phase A: everyone is past its last read
        if (tid < 32) colmask[tid] = 0;
        if (tid == 34) misc[2] = 0;                          // busy flag (set in the list stage, read just above)
        if (tid >= 64 && tid < 64 + MAXB) keys[tid - 64] = 0ull;

        // ================================================================== phase B: start step t
        float curE = 0.f, curI = 0.f;
        const bool quadx = !tailcol && Nin <= 1024 && NT - TT >= B * CW * 4;
        if (!busy && tailcol) {
            if (cvalid) {
                const int nX = cntX[cb_], nI = cntI[cb_], nE = cntE[cb_];
                const uint8_t *xb = sbytes ? sbytes + cb_ * Nin : nullptr;
                int ii[4], ie[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ii[u] = min((int)lstI[cb_ * LR + u], N - 1); ie[u] = min((int)lstE[cb_ * LR + u], N - 1); }
                float wi[4], we[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { wi[u] = wieT[ii[u] * CW + cj_]; we[u] = weiT[ie[u] * CW + cj_]; }
                // X -> Ae: this thread's row_sum lane walks ITS sub-list of the sample's events (digest, grouped by
                // source index mod 4), lane 0 then adds the n % 4 leftover sources in order
                float e1;
                {
                    (void)nX;
                    const uint32_t gc = gcnt[cb_];
                    const int st = (cL > 0 ? (int)(gc & 31u) : 0) + (cL > 1 ? (int)((gc >> 5) & 31u) : 0) + (cL > 2 ? (int)((gc >> 10) & 31u) : 0);
                    const int nL = (int)((gc >> (5 * cL)) & 31u);
                    const uint16_t *l2 = lst2 + cb_ * LX;
                    const int n4 = Nin >> 2;
                    int ix[8]; float wx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ix[u] = min((int)l2[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
                    for (int u = 0; u < 8; ++u) wx[u] = wtile[ix[u] * CW + cj_];
                    CascadeFlat a; a.init();
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (u < nL) a.add(ix[u] >> 2, wx[u] * (xb ? (float)xb[ix[u]] : 1.0f), n4);
                    for (int u = 8; u < nL; ++u) {
                        const int i = (int)l2[st + u];
                        a.add(i >> 2, wtile[i * CW + cj_] * (xb ? (float)xb[i] : 1.0f), n4);
                    }
                    float v = a.finish(n4);
                    if (cL == 0) {
                        const int s4 = (int)(gc & 31u) + (int)((gc >> 5) & 31u) + (int)((gc >> 10) & 31u) + (int)((gc >> 15) & 31u);
                        const int n5 = (int)((gc >> 20) & 31u);
                        for (int u = 0; u < n5; ++u) {
                            const int i = (int)l2[s4 + u];
                            v += wtile[i * CW + cj_] * (xb ? (float)xb[i] : 1.0f);
                        }
                    }
                    const float v1 = __shfl_down(v, 1, 4), v2 = __shfl_down(v, 2, 4), v3 = __shfl_down(v, 3, 4);
                    e1 = ((v + v1) + v2) + v3;
                }
                // recurrent sums: with at most one spiking source the row_sum of a column is that one term (every other
                // lane and level contributes +0.0), so the quad machinery is only needed for two or more
                const float e2 = nI <= 1 ? (nI ? wi[0] * 1.0f + 0.0f : 0.0f) : quad_lane_sum<4>(ii, nI, wi, nullptr, N, cL);
                const float e3 = nE <= 1 ? (nE ? we[0] * 1.0f + 0.0f : 0.0f) : quad_lane_sum<4>(ie, nE, we, nullptr, N, cL);
                if (cL == 0) {
                    curbuf[(cb_ * CW + cj_) * 2] = (0.0f + e1) + e2;
                    curbuf[(cb_ * CW + cj_) * 2 + 1] = 0.0f + e3;
                }
            }
            lds_barrier();
            if (mine) { curE = curbuf[(bl * CW + jj) * 2]; curI = curbuf[(bl * CW + jj) * 2 + 1]; }
        } else if (!busy && quadx) {
            // multi_row_sum columns, X -> Ae part: the cascade's 256-position groups are independent partial sums, so four
            // threads (taken from the waves that are idle in this stage) sum one group of a (sample, column) pair each and
            // lane 0 folds them in the cascade's order; meanwhile the pair's tile thread sums the recurrent parts.
            const int qt = tid - TT;                           // spare threads <-> (sample, column, group)
            if (phaseB && qt >= 0 && qt < (B * CW * 4)) {
                const int pb = qt / (CW * 4), pq = (qt >> 2) % CW, pL = qt & 3;
                const bool pv = c0 + pq < N;
                const uint32_t gq = gqn[pb];
                const int st = (pL > 0 ? (int)(gq & 31u) : 0) + (pL > 1 ? (int)((gq >> 5) & 31u) : 0) + (pL > 2 ? (int)((gq >> 10) & 31u) : 0);
                const int nL = (int)((gq >> (5 * pL)) & 31u);
                const uint16_t *lx = lstX + pb * LX;
                const uint8_t *xb = sbytes ? sbytes + pb * Nin : nullptr;
                int ix[8]; float wx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ix[u] = min((int)lx[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) wx[u] = wtile[ix[u] * CW + pq];
                CascadeFlat a; a.init();
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < nL) a.add(ix[u], wx[u] * (xb ? (float)xb[ix[u]] : 1.0f), Nin);
                for (int u = 8; u < nL; ++u) {
                    const int ii2 = (int)lx[st + u];
                    a.add(ii2, wtile[ii2 * CW + pq] * (xb ? (float)xb[ii2] : 1.0f), Nin);
                }
                const float G = a.a1 + a.a0;                   // the group's sum as the cascade would carry it upward
                const float G1 = __shfl_down(G, 1, 4), G2 = __shfl_down(G, 2, 4), G3 = __shfl_down(G, 3, 4);
                if (pL == 0 && pv) {
                    const int GL = (Nin >> 4) >> 4;            // group holding the cascade's final (pseudo-)block
                    const float Gs[4] = {G, G1, G2, G3};
                    float A2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < GL) A2 = A2 + Gs[k];
                    float Gl = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k == GL) Gl = Gs[k];
                    const float res = ((0.0f + Gl) + A2) + 0.0f;
                    curbuf[(pb * CW + pq) * 2] = 0.0f + res;   // zeros + X->Ae (network.py:225-248)
                }
            }
            float e2 = 0.f, e3 = 0.f;
            if (mine) {
                const int nI = cntI[bl], nE = cntE[bl];
                int ii[4], ie[4]; float wi[4], we[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ii[u] = min((int)lstI[bl * LR + u], N - 1); ie[u] = min((int)lstE[bl * LR + u], N - 1); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { wi[u] = wieT[ii[u] * CW + jj]; we[u] = weiT[ie[u] * CW + jj]; }
                CascadeFlat a; a.init();
#pragma unroll
                for (int u = 0; u < 4; ++u) if (u < nI) a.add(ii[u], wi[u] * 1.0f, N);
                e2 = a.finish(N);
                a.init();
#pragma unroll
                for (int u = 0; u < 4; ++u) if (u < nE) a.add(ie[u], we[u] * 1.0f, N);
                e3 = a.finish(N);
            }
            lds_barrier();
            if (mine) { curE = curbuf[(bl * CW + jj) * 2] + e2; curI = 0.0f + e3; }
        } else if (mine) {
            const int nX = cntX[bl], nI = cntI[bl], nE = cntE[bl];
            const uint8_t *xb = sbytes ? sbytes + bl * Nin : nullptr;
            if (!busy) {
                float wi[4], we[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    wi[u] = wieT[min((int)lstI[bl * LR + u], N - 1) * CW + jj];
                    we[u] = weiT[min((int)lstE[bl * LR + u], N - 1) * CW + jj];
                }
                tile_currents<CascadeFlat, CW>(c, lstX + bl * LX, nX, lstI + bl * LR, nI, lstE + bl * LR, nE, wi, we, wtile, nullptr, jj, xb, j, curE, curI);
            } else {   // generic bit-scan path
                const Cur2 r = busy_currents(wtile, wieT, weiT, c.dig + (size_t)t * c.DW + c.OXW + bl * NinW, spI + bl * NW,
                                             finE + bl * NW, xb, NinW, NW, Nin, N, jj, tailcol, CW);
                curE = r.e; curI = r.i;
            }
        }
        write_raster_rows();           // by the threads that idle while the tile threads update the membranes
        if (busy) lds_barrier();       // (uniform) the bit-scan path reads the exchanged words the next poll overwrites
        DBG_MARK(5);
        // ---- B2: membrane updates
        bool spE = false, spIn = false;
        float r_vE = 0.f, r_vI = 0.f;
        if (mine) {
            float r_rE = stl[1 * TT + tid], r_rI = stl[3 * TT + tid], th = stl[6 * TT + tid];
            r_vE = stl[0 * TT + tid]; r_vI = stl[2 * TT + tid];
            // theta += theta_plus * (spikes of the previous step, summed over the batch), nodes.py:1094 -- applied
            // here, right before this step's decay, instead of behind a barrier of its own at the end of that step
            if (c.pE.learning && t >= 1) th = th + c.pE.theta_plus * (float)cnt[((t - 1) & 1) * CW + jj];
            if (c.pE.learning) th = th * c.pE.theta_decay;                 // nodes.py:1079
            spE = dc_update(r_vE, r_rE, curE, c.pE.lif.thresh + th, c.pE.lif);
            if (spE) atomicAdd(&cnt[(t & 1) * CW + jj], 1);
            float ci = curI;
            if (r_rI > 0.f) ci = 0.f;
            spIn = lif_update(r_vI, r_rI, ci, c.pI);
            last_sI = spIn;
            stl[0 * TT + tid] = r_vE; stl[1 * TT + tid] = r_rE; stl[2 * TT + tid] = r_vI; stl[3 * TT + tid] = r_rI; stl[6 * TT + tid] = th;
            if (c.pI.traces) stl[5 * TT + tid] = trace_next(stl[5 * TT + tid], spIn, c.pI.trace_decay, c.pI.trace_scale, c.pI.traces_additive);
        }
        {   // publish crossing / spike bits of step t: epoch t+1.  A wave holds 64/CW samples x CW columns; SPG
            // consecutive samples share a granule
            const uint64_t mE = __ballot(spE), mI = __ballot(spIn);
            constexpr int SPW = 64 / CW;
            const int sidx = lane / CW, b = wave * SPW + sidx;
            // SPG consecutive samples x CW columns = 16 consecutive bits of each ballot: no cross-lane traffic needed
            const uint32_t v = (uint32_t)((mE >> (sidx * CW)) & 0xFFFFull) | ((uint32_t)((mI >> (sidx * CW)) & 0xFFFFull) << 16);
            if (tid < TT && (lane % CW) == 0 && (sidx % SPG) == 0 && b < B)
                granule_store(c.ex + (size_t)((t + 1) & 1) * NG + g * KB + b / SPG, ((unsigned long long)(uint32_t)(t + 1) << 32) | v);
        }
        if (mine) {
            if (c.rasVE) (c.rasVE + (size_t)t * B * N)[kst] = r_vE;
            if (c.rasVI) (c.rasVI + (size_t)t * B * N)[kst] = r_vI;
        }
        DBG_MARK(6);
        DBG_MARK(7);
        if (c.dbg && blockIdx.x == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + 9] = (long long)clock64();
        if (c.dbg && threadIdx.x == 0) atomicMax((unsigned long long *)&c.dbg[(size_t)t * 24 + 21], (unsigned long long)wall_clock64());
    }

    // ---- epilogue: state and weights back to the tensors the caller owns
    if (mine) {
        float th = stl[6 * TT + tid];
        if (c.pE.learning) th = th + c.pE.theta_plus * (float)cnt[((T - 1) & 1) * CW + jj];   // the last step's spikes
        c.vE[kst] = stl[0 * TT + tid]; c.rE[kst] = stl[1 * TT + tid]; c.vI[kst] = stl[2 * TT + tid]; c.rI[kst] = stl[3 * TT + tid];
        if (bl == 0) c.theta[j] = th;
        if (c.pI.traces) c.xI[kst] = stl[5 * TT + tid];
        if (c.pE.lif.traces) c.xE[kst] = stl[4 * TT + tid];
        c.sE[kst] = last_sE; c.sI[kst] = last_sI;
    }
    if (c.has_norm) {
        // network.py:464-465 + topology_features.py:250-266: column sums in ATen's sum(dim=0) order over the
        // LDS-resident slice, zero -> 1, W *= norm * (1 / colsum)   (same arithmetic as k_colsum / k_scale_cols)
        float *bsum = (float *)dgbuf;                  // [nfull][CW] block sums; the digest buffers are free now
        float *sc = xnu0;                              // [CW] column scales
        const int nfull = Nin >> 4;
        __syncthreads();
        if (!tailcol) {
            for (int item = tid; item < nfull * CW; item += NT) {
                const int blk = item / CW, q = item % CW;
                float a0 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float w = wtile[(blk * 16 + k) * CW + q]; a0 += c.norm_abs ? fabsf(w) : w; }
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
                for (int i = nfull * 16; i < Nin; ++i) { const float w = wtile[i * CW + tid]; a0 += c.norm_abs ? fabsf(w) : w; }
                float cs = ((a0 + a1) + a2) + a3;
                if (cs == 0.f) cs = 1.0f;
                sc[tid] = (1.0f / cs) * c.norm;
            }
        } else {
            __syncthreads();
            if (tid < CW * 4) {                         // row_sum columns: four interleaved lanes per column
                const int q = tid >> 2, s4 = tid & 3, n4 = Nin >> 2, nf4 = n4 >> 4;
                Cascade cc; cc.init();
                for (int p_ = 0; p_ < n4; ++p_) { const float w = wtile[(4 * p_ + s4) * CW + q]; cc.add(p_, c.norm_abs ? fabsf(w) : w, nf4); }
                float lsum = cc.finish(nf4);
                if (s4 == 0)
                    for (int i = n4 * 4; i < Nin; ++i) { const float w = wtile[i * CW + q]; lsum += c.norm_abs ? fabsf(w) : w; }
                const float l1 = __shfl_down(lsum, 1, 4), l2 = __shfl_down(lsum, 2, 4), l3 = __shfl_down(lsum, 3, 4);
                float cs = ((lsum + l1) + l2) + l3;
                if (cs == 0.f) cs = 1.0f;
                if (s4 == 0) sc[q] = (1.0f / cs) * c.norm;
            }
        }
        __syncthreads();
        for (int k = tid; k < Nin * CW; k += NT) {
            const int i = k / CW, q = k % CW;
            if (c0 + q < N) c.Wxe[i * N + c0 + q] = wtile[k] * sc[q];
        }
    } else if (c.learning && c.rule == SNN_RULE_POSTPRE) {
        for (int k = tid; k < Nin * CW; k += NT) {
            const int i = k / CW, q = k % CW;
            if (c0 + q < N) c.Wxe[i * N + c0 + q] = wtile[k];
        }
    }
    if (g == 0 && c.pE.one_spike) {
        snn_rng_state *wr = c.rng[0];
        for (int k = tid; k < 624; k += NT) wr->mt[k] = mt[mb * 624 + k];
        if (tid == 0) { wr->pos = rng_pos; wr->consumed = rng_consumed; }
    }
}

// words of one digest entry / of its part that the step kernel copies into LDS
// one digest entry: [part staged in LDS: lists | meta | row masks | active rows | row -> index | lane-grouped lists | group
// sizes] padded to 4 words, then the [B][NinW] bit words (read from global memory by the bit-scan path only)
int digest_lds_words(int B, int Nin) { return (B * (LX / 2) + 40 + Nin + 2 * ((Nin + 1) / 2) + B * (LX / 2) + 2 * B + 3) & ~3; }
int digest_words(int B, int Nin) { return (digest_lds_words(B, Nin) + B * ((Nin + 31) / 32) + 3) & ~3; }

size_t lds_bytes(int B, int Nin, int N) {
    const int NW = (N + 31) / 32;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return al((size_t)digest_lds_words(B, Nin) * 4) + 3 * al((size_t)B * NW * 4) +
           (size_t)MAXB * CW * 4 + 8 * 624 * 4 + NCAND * 4 + MAXB * 8 + 2 * MAXB * LR * 2 + 2 * MAXB * 4 + 2 * 32 * 4 + 32 +
           (size_t)Nin * CW * 4 + 2 * MAXB * CW * 4;
}

size_t lds_bytes_resident(int B, int Nin, int N, int cw) {
    const int DGS = (digest_lds_words(B, Nin) + 63) & ~63;
    return resident_fixed_lds(cw) + (size_t)Nin * cw * 4 + (size_t)2 * N * cw * 4 + (size_t)2 * DGS * 4;
}

// tile width of the resident kernel: the narrowest of 8 / 4 / 2 columns whose grid still fits one workgroup per CU with
// room to spare (SNN_DC_CW overrides)
int resident_nt() {
    if (const char *e = getenv("SNN_DC_NT")) { const int v = atoi(e); if (v == 512 || v == 1024) return v; }
    return kResidentDefaultNT;
}
int resident_cw(int N) {
    if (const char *e = getenv("SNN_DC_CW")) { const int v = atoi(e); if (v == 8 || v == 4 || v == 2) return v; }
    return kResidentDefaultCW;
}

size_t prep_lds_bytes(int B, int Nin) { return (size_t)(B * ((Nin + 31) / 32) + Nin + 4) * 4 + 2 * (NT / 64) * LX * 2; }

}  // namespace

// Device scratch the fused plan needs (bytes); 0 if the graph does not match the plan.
static size_t fused_workspace(int B, int Nin, int N) {
    const int NW = (N + 31) / 32;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return 4 * al((size_t)B * NW * 4) + al((size_t)B * Nin * 4) + al(sizeof(snn_rng_state));
}

static size_t resident_extra(int B, int Nin, int N, int T) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t gran = (size_t)2 * ((N + 1) / 2) * ((B + 1) / 2) * 8;      // >= 2 * G * KB * 8 for every tile width
    return al(gran) + al((size_t)(T + 1) * B * Nin * 4);
}

// The resident form keeps the X trace of every step ((T+1)*B*Nin floats): beyond this it is not offered and long runs
// take the one-launch-per-timestep form, whose scratch does not grow with B*Nin*T.
constexpr size_t kResidentMaxExtra = (size_t)2 << 30;

static size_t fused_workspace_total(int B, int Nin, int N, int T) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t extra = resident_extra(B, Nin, N, T);
    return fused_workspace(B, Nin, N) + al((size_t)(T + 1) * digest_words(B, Nin) * 4) + (extra <= kResidentMaxExtra ? extra : 0);
}

static bool matches(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R) {
    if (nL != 3 || nC != 3) return false;
    if (L[0].kind != SNN_LAYER_INPUT || L[1].kind != SNN_LAYER_DC || L[2].kind != SNN_LAYER_LIF) return false;
    if (L[1].n != L[2].n) return false;
    for (int k = 0; k < 3; ++k) if (C[k].kind != SNN_CONN_MCC || C[k].bias) return false;
    if (C[0].src != 0 || C[0].dst != 1 || C[1].src != 1 || C[1].dst != 2 || C[2].src != 2 || C[2].dst != 1) return false;
    if (C[1].rule != SNN_RULE_NONE || C[2].rule != SNN_RULE_NONE) return false;
    if (C[0].rule != SNN_RULE_NONE && C[0].rule != SNN_RULE_POSTPRE) return false;
    if (C[0].rule == SNN_RULE_POSTPRE && (C[0].wdecay != 1.0f || !L[0].x || !L[1].x)) return false;
    if (R->B > MAXB) return false;
    if (L[0].n > 2048 || L[0].n % 16 != 0 || L[1].n > 1024 || R->B * ((L[1].n + 31) / 32) > NT) return false;
    if ((size_t)R->B * L[0].n > (size_t)NU * NT * 16) return false;
    if ((double)(R->T + 1) * R->B * L[0].n >= 2147483648.0) return false;
    if (L[1].p.one_spike && !R->rng) return false;
    if (R->T < 1) return false;
    if (lds_bytes(R->B, L[0].n, L[1].n) > 150 * 1024) return false;
    if (!R->workspace || R->workspace_bytes < fused_workspace_total(R->B, L[0].n, L[1].n, R->T)) return false;
    if (digest_lds_words(R->B, L[0].n) > 3 * NT) return false;
    return true;
}

extern "C" unsigned long long snn_net_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                          const snn_run_desc *R) {
    if (!L || !R || !C) return 0;
    if (nL == 2 && nC == 1) return snn_twolayer_workspace_bytes(L, nL, C, nC, R);
    if (nL != 3 || nC != 3) return 0;
    return fused_workspace_total(R->B, L[0].n, L[1].n, R->T);
}

void snn_set_plan_name(const char *name);
unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                                const snn_run_desc *R);
int g_graph_stats[3] = {0, 0, 0};   // runs enqueued as plain launches / captured / replayed

extern "C" void snn_graph_stats(int *h_plain, int *h_captured, int *h_replayed) {
    if (h_plain) *h_plain = g_graph_stats[0];
    if (h_captured) *h_captured = g_graph_stats[1];
    if (h_replayed) *h_replayed = g_graph_stats[2];
}

int snn_try_fused_dc2015(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                         hipStream_t st, int resident, int *handled, unsigned *normalized) {
    *handled = 0;
    *normalized = 0;
    if (!matches(L, nL, C, nC, R)) return SNN_OK;
    const int B = R->B, Nin = L[0].n, N = L[1].n;
    DcCtx c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.Nin = Nin; c.N = N; c.T = R->T; c.NW = (N + 31) / 32; c.NinW = (Nin + 31) / 32;
    c.G = (N + CW - 1) / CW; c.RS = (Nin + c.G - 1) / c.G;
    c.dt = R->dt; c.learning = R->learning;
    c.inv_hwps = 1.0f / (float)(Nin >> 4); c.inv_NW = 1.0f / (float)c.NW; c.inv_RS = 1.0f / (float)c.RS;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    unsigned char *ws = (unsigned char *)R->workspace;
    const size_t wb = al((size_t)B * c.NW * 4);
    c.crossE[0] = (uint32_t *)ws; c.crossE[1] = (uint32_t *)(ws + wb);
    c.spikeI[0] = (uint32_t *)(ws + 2 * wb); c.spikeI[1] = (uint32_t *)(ws + 3 * wb);
    float *xscratch = (float *)(ws + 4 * wb);
    snn_rng_state *rng2 = (snn_rng_state *)(ws + 4 * wb + al((size_t)B * Nin * 4));
    c.dig = (uint32_t *)(ws + fused_workspace(B, Nin, N));
    c.DW = digest_words(B, Nin); c.DGW = digest_lds_words(B, Nin); c.OXW = c.DGW;
    c.in = L[0].ext_spikes; c.sX0 = L[0].s;
    c.x_traces = L[0].p.lif.traces; c.x_decay = L[0].p.lif.trace_decay; c.x_scale = L[0].p.lif.trace_scale;
    c.x_additive = L[0].p.lif.traces_additive;
    c.xX[1] = L[0].x; c.xX[0] = xscratch;
    c.vE = L[1].v; c.rE = L[1].refrac; c.xE = L[1].x; c.theta = L[1].theta; c.sE = L[1].s; c.pE = L[1].p;
    c.rasE = L[1].raster_s; c.rasVE = L[1].raster_v;
    c.vI = L[2].v; c.rI = L[2].refrac; c.xI = L[2].x; c.sI = L[2].s; c.pI = L[2].p.lif;
    c.rasI = L[2].raster_s; c.rasVI = L[2].raster_v;
    c.Wxe = C[0].w; c.Wei = C[1].w; c.Wie = C[2].w;
    c.rule = C[0].rule; c.nu0 = C[0].nu0; c.nu1 = C[0].nu1; c.use_dt = C[0].use_dt;
    c.has_min = C[0].has_min; c.wmin = C[0].wmin; c.has_max = C[0].has_max; c.wmax = C[0].wmax;
    // generator: launch t reads rng[(t-1)&1], workgroup 0 writes rng[t&1]; entry state must sit in rng[0]
    c.rng[0] = R->rng; c.rng[1] = rng2;
    {
        auto al2 = [](size_t x) { return (x + 255) & ~(size_t)255; };
        unsigned char *p = (unsigned char *)c.dig + al2((size_t)(R->T + 1) * c.DW * 4);
        c.ex = (unsigned long long *)p;
        c.xtr = (float *)(p + al2((size_t)2 * ((N + 1) / 2) * ((B + 1) / 2) * 8));
        c.status = R->status;
        c.has_norm = C[0].has_norm; c.norm = C[0].norm; c.norm_abs = C[0].norm_abs;
    }
    // resident plan needs every workgroup on its own CU at once and its first (clamp-everything) PostPre pass at t = 1
    if (getenv("SNN_DC_RESIDENT")) resident = atoi(getenv("SNN_DC_RESIDENT"));
    int rcw = resident_cw(N);
    while (rcw < 8 && (N + rcw - 1) / rcw > 256) rcw *= 2;          // one workgroup per CU (256 CUs), all co-resident
    const int rG = (N + rcw - 1) / rcw, rKB = (B + 16 / rcw - 1) / (16 / rcw);
    if (rG > 256 || (c.rule == SNN_RULE_POSTPRE && !c.x_traces) || lds_bytes_resident(B, Nin, N, rcw) > 150 * 1024 ||
        resident_extra(B, Nin, N, R->T) > kResidentMaxExtra) resident = 0;
    int rnt = resident_nt();
    if (rnt == 512 && (rcw > 4 || B * c.NW > 512)) rnt = 1024;       // single-pass stages of the 512-thread variant
    if (resident) { c.G = rG; c.KB = rKB; }
    static long long *dbg = nullptr;
    static int dbg_T = 0;
    if (getenv("SNN_DC_TIMING")) {
        if (!dbg || dbg_T < R->T + 1) { if (dbg) (void)hipFree(dbg); (void)hipMalloc(&dbg, sizeof(long long) * 24 * (R->T + 1)); dbg_T = R->T + 1; }
        (void)hipMemsetAsync(dbg, 0, sizeof(long long) * 24 * (R->T + 1), st);
        for (int t = 0; t <= R->T; ++t) (void)hipMemsetAsync(dbg + (size_t)t * 24 + 20, 0x7F, sizeof(long long), st);
        c.dbg = dbg; c.dbg_wg = atoi(getenv("SNN_DC_TIMING")); if (c.dbg_wg < 0 || c.dbg_wg >= c.G) c.dbg_wg = c.G - 1;
    }
    const size_t lds = lds_bytes(B, Nin, N);
    static bool lds_attr = false;
    if (!lds_attr) {   // the kernel may use more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        if (snn_check(hipFuncSetAttribute((const void *)k_dc2015_step, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) return SNN_ERR_LAUNCH;
        const void *rv[5] = {(const void *)k_dc2015_run<8, 1024>, (const void *)k_dc2015_run<4, 1024>, (const void *)k_dc2015_run<2, 1024>,
                             (const void *)k_dc2015_run<4, 512>, (const void *)k_dc2015_run<2, 512>};
        for (const void *f : rv)
            if (snn_check(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) return SNN_ERR_LAUNCH;
        lds_attr = true;
    }
    // One run = memset of the exchange words (pad bytes for columns >= N are never written by a workgroup),
    // T+1 launches, and up to two small copies (final X trace sits in xX[(T-1)&1], final generator in rng[T&1]).
    hipStream_t qs = st;           // stream the run is enqueued on (a private one when graphs are in play)
    auto enqueue = [&](bool with_events) -> int {
        int rc0;
        if (resident) {
            // memset of the exchange granules (epochs restart at 1 every run), input-only pre-passes, ONE launch
            if ((rc0 = snn_check(hipMemsetAsync(c.ex, 0, (size_t)2 * c.G * c.KB * 8, qs)))) return rc0;
            hipLaunchKernelGGL(k_dc2015_prep, dim3(R->T + 1), dim3(NT), prep_lds_bytes(B, Nin), qs, c);
            if (c.x_traces) hipLaunchKernelGGL(k_dc2015_xtrace, dim3((B * Nin + 255) / 256), dim3(256), 0, qs, c);
            const bool prof = with_events && snn_prof_begin(0, qs);
            const size_t rl = lds_bytes_resident(B, Nin, N, rcw);
            if (rnt == 512 && rcw == 4) hipLaunchKernelGGL((k_dc2015_run<4, 512>), dim3(c.G), dim3(512), rl, qs, c);
            else if (rnt == 512) hipLaunchKernelGGL((k_dc2015_run<2, 512>), dim3(c.G), dim3(512), rl, qs, c);
            else if (rcw == 8) hipLaunchKernelGGL((k_dc2015_run<8, 1024>), dim3(c.G), dim3(1024), rl, qs, c);
            else if (rcw == 4) hipLaunchKernelGGL((k_dc2015_run<4, 1024>), dim3(c.G), dim3(1024), rl, qs, c);
            else hipLaunchKernelGGL((k_dc2015_run<2, 1024>), dim3(c.G), dim3(1024), rl, qs, c);
            if (prof) snn_prof_end(qs);
            return snn_check_launch();
        }
        rc0 = snn_check(hipMemsetAsync(ws, 0, 4 * wb, qs));
        if (rc0) return rc0;
        hipLaunchKernelGGL(k_dc2015_prep, dim3(R->T + 1), dim3(NT), prep_lds_bytes(B, Nin), qs, c);
        for (int t = 0; t <= R->T; ++t) {
            const bool prof = with_events && snn_prof_begin(t, qs);
            hipLaunchKernelGGL(k_dc2015_step, dim3(c.G), dim3(NT), lds, qs, c, t);
            if (prof) snn_prof_end(qs);
        }
        if ((rc0 = snn_check_launch())) return rc0;
        if (c.x_traces && ((R->T - 1) & 1) == 0)
            if ((rc0 = snn_check(hipMemcpyAsync(L[0].x, xscratch, sizeof(float) * (size_t)B * Nin, hipMemcpyDeviceToDevice, qs)))) return rc0;
        if (L[1].p.one_spike && (R->T & 1))
            if ((rc0 = snn_check(hipMemcpyAsync(R->rng, rng2, sizeof(snn_rng_state), hipMemcpyDeviceToDevice, qs)))) return rc0;
        return SNN_OK;
    };


    // hipGraph replay (opt-in, SNN_GRAPH=1): the launch sequence of a run is fully determined by the context (pointers + sizes), and
    // training loops present the same contexts again and again (same network, recycled input / monitor buffers).
    // First sight of a context: plain launches.  Second sight: capture + instantiate.  Afterwards: one
    // hipGraphLaunch per run instead of T+1 kernel launches (the host stops being the bottleneck).
    struct GraphEntry { DcCtx key; hipGraphExec_t exec; unsigned long long stamp; };
    static std::vector<GraphEntry> cache;
    static unsigned long long clock_ = 0;
    static const bool graphs_on = getenv("SNN_GRAPH") != nullptr;   // opt-in: replay measured no faster than eager launches
    int rc = SNN_OK;
    if (!graphs_on || resident || c.dbg || snn_prof_active()) {
        rc = enqueue(true); g_graph_stats[0]++;
    } else {
        // capture is not allowed on the legacy default stream torch usually runs on: fork to a private
        // non-blocking stream (event edge in, event edge out), so `st` still orders everything around the run
        static hipStream_t side = nullptr;
        static hipEvent_t ev_in = nullptr, ev_out = nullptr;
        if (!side) {
            if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) { side = nullptr; (void)hipGetLastError(); }
        }
        GraphEntry *hit = nullptr;
        for (auto &e : cache) if (memcmp(&e.key, &c, sizeof(DcCtx)) == 0) { hit = &e; break; }
        if (!side) {
            rc = enqueue(false); g_graph_stats[0]++;
        } else if (hit) {
            if ((rc = snn_check(hipEventRecord(ev_in, st))) || (rc = snn_check(hipStreamWaitEvent(side, ev_in, 0)))) return rc;
            qs = side;
            if (!hit->exec) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                bool ok = hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    const int rce = enqueue(false);
                    ok = hipStreamEndCapture(side, &graph) == hipSuccess && rce == SNN_OK && graph;
                }
                if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (ok) { hit->exec = exec; g_graph_stats[1]++; }
                else { (void)hipGetLastError(); hit->exec = nullptr; }
            } else {
                g_graph_stats[2]++;
            }
            hit->stamp = ++clock_;
            if (hit->exec) rc = snn_check(hipGraphLaunch(hit->exec, side));
            else { rc = enqueue(false); g_graph_stats[0]++; }
            if (rc) return rc;
            if ((rc = snn_check(hipEventRecord(ev_out, side))) || (rc = snn_check(hipStreamWaitEvent(st, ev_out, 0)))) return rc;
        } else {
            if (cache.size() >= 16) {           // evict the least recently used context
                size_t v = 0;
                for (size_t k = 1; k < cache.size(); ++k) if (cache[k].stamp < cache[v].stamp) v = k;
                if (cache[v].exec) (void)hipGraphExecDestroy(cache[v].exec);
                cache.erase(cache.begin() + v);
            }
            cache.push_back(GraphEntry{c, nullptr, ++clock_});
            rc = enqueue(false); g_graph_stats[0]++;
        }
    }
    if (rc) return rc;
    if (c.dbg) {   // developer aid: average phase durations (100 MHz wall clock ticks -> us)
        (void)hipStreamSynchronize(st);
        std::vector<long long> h((size_t)24 * (R->T + 1));
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double acc[8] = {0}; int n = 0;
        for (int t = 2; t < R->T; ++t, ++n)
            for (int k = 1; k < 8; ++k) acc[k] += (double)(h[(size_t)t * 24 + k] - h[(size_t)t * 24 + k - 1]) / 100.0;
        double cyc = 0, us = 0;
        for (int t = 2; t < R->T; ++t) { cyc += (double)(h[(size_t)t * 24 + 9] - h[(size_t)t * 24 + 8]); us += (double)(h[(size_t)t * 24 + 7] - h[(size_t)t * 24]) / 100.0; }
        {   // whole-grid view: first workgroup start -> last workgroup end, and the gap to the next launch
            double span = 0, gap = 0; int m = 0;
            for (int t = 2; t + 1 < R->T; ++t, ++m) {
                span += (double)(h[(size_t)t * 24 + 21] - h[(size_t)t * 24 + 20]) / 100.0;
                gap += (double)(h[(size_t)(t + 1) * 24 + 20] - h[(size_t)t * 24 + 21]) / 100.0;
            }
            fprintf(stderr, "[dc2015 grid, us] first-WG-start -> last-WG-end %.2f | last-WG-end -> next launch first-WG-start %.2f\n", span / m, gap / m);
        }
        fprintf(stderr, "[dc2015 clock] %.0f MHz shader clock during the kernel\n", cyc / us);
        double sub[3] = {0, 0, 0};
        for (int t = 2; t < R->T; ++t) { sub[0] += (h[(size_t)t*16+10]-h[(size_t)t*16+0])/100.0; sub[1] += (h[(size_t)t*16+11]-h[(size_t)t*16+10])/100.0; sub[2] += (h[(size_t)t*16+1]-h[(size_t)t*16+11])/100.0; }
        {
            double q[5] = {0, 0, 0, 0, 0};
            for (int t = 2; t < R->T; ++t) { const long long *r = &h[(size_t)t * 24]; q[0] += (r[10]-r[0])/100.0; q[1] += (r[11]-r[10])/100.0; q[2] += (r[15]-r[11])/100.0; q[3] += (r[16]-r[15])/100.0; q[4] += (r[1]-r[16])/100.0; }
            fprintf(stderr, "[dc2015 stage detail, us] issue-loads (resident: poll) %.2f | barrier(loads land) (resident: digest->LDS) %.2f | lds-store+barrier %.2f | prefetch issue %.2f | Ai lists + twists %.2f\n", q[0]/n, q[1]/n, q[2]/n, q[3]/n, q[4]/n);
        }
        double ab[4] = {0, 0, 0, 0};
        for (int t = 2; t < R->T; ++t) { const long long *r = &h[(size_t)t * 24]; ab[0] += (r[12]-r[1])/100.0; ab[1] += (r[13]-r[12])/100.0; ab[2] += (r[14]-r[13])/100.0; ab[3] += (r[2]-r[14])/100.0; }
        fprintf(stderr, "[dc2015 arb detail, us] barrier %.2f | candidates+twist %.2f | winners+publish+barrier %.2f | lstE %.2f\n", ab[0]/n, ab[1]/n, ab[2]/n, ab[3]/n);
        fprintf(stderr, "[dc2015 timing, us] stage %.2f | arb %.2f | A2 %.2f | stdp %.2f | cur %.2f | membrane %.2f | xtrace %.2f\n",
                acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, acc[7] / n);
    }
    if (resident && c.has_norm) *normalized |= 1u;          // connection 0 was normalised in the kernel's epilogue
    snn_set_plan_name(resident ? "dc2015-resident" : "dc2015-fused");
    *handled = 1;
    return SNN_OK;
}

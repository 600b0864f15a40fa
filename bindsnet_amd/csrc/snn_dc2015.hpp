// snn_dc2015.hpp -- shared by the two translation units of the DiehlAndCook2015 plans (snn_dc2015.hip: digest
// pre-pass, one-launch-per-timestep kernel, host side; snn_dc2015_resident.hip: the resident whole-run kernel):
// tile constants, the kernel context, bit / list / ordered-sum helpers, digest layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"
#include "snn_order.hpp"
#include "snn_rng.hpp"

// Kernel context of both plans (external linkage: it crosses the two translation units).
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
    // [B] group sizes (GCB = 6 bits each: lanes 0..3, leftover sources) | [B] events per 256-position group (6 bits each)
    // meta[33] flags: 1 = a spike byte other than 0/1, 2 = a sample with > LXF events (the 16-entry fast paths do not
    // apply), 4 = a sample with > LX - 1 events (lists / packed group sizes overflow: the lean resident form gives up)
    uint32_t *dig; int DW, DGW, OXW;        // words per entry, words of its LDS part, offset of its bit words
    // resident plan (k_dc2015_run): 8-byte {epoch, bits} exchange granules [2][G][KB], the X trace after every
    // step [T+1][B][Nin] (entry 0 = trace at run entry), device status word
    unsigned long long *ex; int KB;
    unsigned long long *exs;    // lean form: summary granules [2][G][tile waves]
    float *xtr;
    int *status;
    int has_norm; float norm; int norm_abs;   // post-run normalisation of Wxe, done in the resident kernel's epilogue
    int zone_shift;             // lean arbitration: draws within 2^-zone_shift (relative) of the smallest are evaluated exactly
                                // (19; test hook SNN_DC_TEST_ZONE lowers it so that the exact branch runs all the time)
    int spec_flags;             // second-generation lean form, developer switch SNN_DC_SPECFLAGS: 1 = no prepared won branch, 2 = every iteration on the slow order
    int rows4;                  // lean form: PostPre one thread per active row (developer switch SNN_DC_ROWS4=0: per (row, column))
    int stall_wg;               // test hook (SNN_DC_TEST_STALL=<workgroup>, -1 = none): that workgroup of the resident kernel
                                // exits at once, as if it had never been scheduled -> every other one times out
    int dbg_wg;
    long long *dbg;             // developer aid (SNN_DC_TIMING=1): per-launch phase timestamps of workgroup 0
    // third-generation lean form (k_dc2015_async, snn_dc2015_async.hip): winners granules [kWinRing][kWinGr] written by the arbiter
    // workgroup, progress words of the raster workgroups, number of raster workgroups, and the first digest entry the lean forms do
    // not handle (k_dc2015_prep: meta[33] & 5; INT_MAX when none) -- its exchange granules ex / exs are rings of FOUR steps
    unsigned long long *wing;
    int *rprog;
    int NRW;
    int *tbad;                  // (zeroed with the exchange area by the run's memset; k_dc2015_prep ORs 1 into it for an entry the lean forms do not take)
    // second attempt enqueued behind the first (pipelined callers, snn_run_desc.status2): the general resident kernel runs only if the word
    // at `gate` holds SNN_ERR_RETRY -- otherwise every workgroup returns at once
    const int *gate;
    // third generation with producer workgroups (NP > 0): workgroups behind the raster writers compute the digest entries and the X traces
    // while the compute workgroups run.  dready[e] (e <= T): 0 = not yet, 1 = entry e is there, 2 = there and one the lean forms do not take;
    // dready[T+1]: entries finished, dready[T+2]: X-trace chunks (256 (sample, source) pairs each) finished.  Release / acquire at agent scope.
    int *dready;
    int NP;
    uint8_t *rasX;              // NP > 0, nullable: the Input layer's spike raster [T][B][Nin] -- a copy of the input, made by the producers
    // gated second attempt (gate != nullptr): its workgroups first clear two exchange areas for the NEXT run of a pipelined caller (the
    // third generation's, and the general form's other copy) -- the memsets a run otherwise starts with
    uint4 *zeroA, *zeroG;
    unsigned zeroA_n16, zeroG_n16;
};

namespace {
using namespace snn;

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

// Cascade taking (and ignoring) the `tail` flag at init, interface-compatible with OuterSum.
struct CascadeT {
    Cascade c;
    __device__ __forceinline__ void init(bool) { c.init(); }
    __device__ __forceinline__ void add(int pos, float term, int n) { c.add(pos, term, n >> 4); }
    __device__ __forceinline__ float finish(int n) { return c.finish(n >> 4); }
};

constexpr int LX = 64, LR = 8;   // per-sample event-list capacities (X sources / recurrent sources)
constexpr int LXF = 16;          // ... of them the fixed-size fast paths of the per-step / general resident kernels unroll
constexpr int GCB = 6;           // bits per group size in the digest's packed group counts (a group holds <= LX - 1 events)
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

// one digest entry: [part staged in LDS: lists | meta | row masks | active rows | row -> index | lane-grouped lists | group
// sizes] padded to 4 words, then the [B][NinW] bit words (read from global memory by the bit-scan path only)
int digest_lds_words(int B, int Nin) { return (B * (LX / 2) + 40 + Nin + 2 * ((Nin + 1) / 2) + B * (LX / 2) + 2 * B + 3) & ~3; }
int digest_words(int B, int Nin) { return (digest_lds_words(B, Nin) + B * ((Nin + 31) / 32) + 3) & ~3; }

// One digest entry (entry 0 = the layer's `s` at entry, entry e = inputs[e-1]) by one workgroup of NTH threads: written to D (global memory:
// k_dc2015_prep, one workgroup per entry; or an LDS staging copy of the entry: the producer workgroups of k_dc2015_async, which publish it with
// write-through stores).  LDS: dc_prep_lds_bytes(B, Nin, NTH) at smem.  Returns the entry's flags (meta[33]; uniform, behind a barrier).
template <int NTH>
__device__ __forceinline__ int dc_prep_entry(const DcCtx &c, unsigned char *smem, int e, uint32_t *D) {
    const int B = c.B, Nin = c.Nin, NinW = c.NinW;
    uint32_t *sXw = (uint32_t *)smem;                               // [B][NinW]
    uint32_t *rowmask = sXw + B * NinW;                             // [Nin]
    int *misc = (int *)(rowmask + Nin);                             // [0] nact, [1] flags
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *src = (e == 0) ? c.sX0 : c.in + (size_t)(e - 1) * B * Nin;
    uint32_t *D_xl = D, *D_meta = D_xl + B * (LX / 2), *D_rm = D_meta + 40, *D_xw = D + c.OXW;   // (bit words last)
    uint16_t *D_ar = (uint16_t *)(D_rm + Nin), *D_rp = D_ar + 2 * ((Nin + 1) / 2);
    for (int k = tid; k < Nin; k += NTH) rowmask[k] = 0;
    if (tid < 2) misc[tid] = 0;
    __syncthreads();
    {
        const int total16 = (B * Nin) >> 4, hwps = Nin >> 4, HS = NinW * 2;
        uint16_t *sXh = (uint16_t *)sXw;
        uint32_t big = 0;
        for (int k16 = tid; k16 < total16; k16 += NTH) {
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
    uint32_t *D_gc = (uint32_t *)(D_l2 + B * LX);                    // [B] five GCB-bit group sizes
    uint16_t *lscr = (uint16_t *)(misc + 4) + wave * LX;             // this wave's scratch list
    for (int b = wave; b < B; b += NTH / 64) {
        const int nx = build_list(sXw + b * NinW, NinW, lane, lscr, LX);
        if (lane == 0) { D_meta[b] = (uint32_t)nx; if (nx > LXF) atomicOr((unsigned int *)&misc[1], 2u); if (nx > LX - 1) atomicOr((unsigned int *)&misc[1], 4u); }
        // (LDS operations of one wave execute in program order: the list is readable right away)
        const bool have = lane < LX && lane < nx;
        const int i = have ? (int)lscr[lane] : 0;
        if (lane < LX) ((uint16_t *)D_xl)[b * LX + lane] = (uint16_t)i;
        // the same events grouped by ATen row_sum lane (index mod 4; group 4 = the n % 4 leftover sources), ascending
        // inside a group: what a quad of threads walks for a column >= 32*floor(N/32)
        const bool in16 = have;                     // (every listed event: the consumers walk group sizes, not 16 slots)
        const int grp = (i >= ((Nin >> 2) << 2)) ? 4 : (i & 3);
        int start = 0, my = 0; uint32_t gc = 0;
        for (int k = 0; k < 5; ++k) {
            const uint64_t mk = __ballot(in16 && grp == k);
            const int ck = __popcll(mk);
            if (grp == k) my = start + __popcll(mk & ((1ull << lane) - 1ull));
            start += ck; gc |= (uint32_t)min(ck, (1 << GCB) - 1) << (GCB * k);
        }
        uint16_t *perm = lscr + (NTH / 64) * LX;        // second per-wave scratch: permute in LDS, store each slot once
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
                gq |= (uint32_t)min((int)__popcll(mk), (1 << GCB) - 1) << (GCB * k);
            }
            if (lane == 0) D_gc[B + b] = gq;
        }
    }
    for (int k = tid; k < B * NinW; k += NTH) D_xw[k] = sXw[k];
    for (int base = 0; base < Nin; base += NTH) {       // compact the rows with a spike in any sample
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
    const int flags = misc[1];
    __syncthreads();                                                // (misc is rewritten by the workgroup's next entry)
    return flags;
}
// X trace after every step: entry 0 = trace at run entry, entry e = trace after step e-1 (nodes.py:96-103).
// ADD: additive traces (x = x * decay + scale * s) / replacing ones (x = s ? scale : x * decay), branch-free either way.
template <bool ADD>
__device__ __forceinline__ float xtrace_next(float x, uint8_t s, float decay, float scale) {
    const float t = x * decay;
    if (ADD) return t + scale * (float)s;
    return s ? scale : t;
}

template <bool ADD>
__device__ __forceinline__ void xtrace_body(const DcCtx &c, int n, int k) {
    float x = c.xX[1][k];
    c.xtr[k] = x;
    int t = 0;
    // The spike loads do not depend on x: 32 are issued together, and the NEXT 32 before the current 32 trace values are
    // stored, so the loop body is "32 loads, 32 multiplies / selects / stores, one s_waitcnt vmcnt(32)".  (Measured: 20-21 us
    // at cfg2 with 8 or 32 loads per batch, pipelined or not -- 25 MB written by 392 waves; not on the critical path's
    // scale: the resident launch behind it takes 1.8 ms.)
    if (c.T >= 32) {
        uint8_t s[32], sn[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) s[u] = c.in[(size_t)u * n + k];
        for (; t + 32 <= c.T; t += 32) {
            const bool more = t + 64 <= c.T;
            if (more) {
#pragma unroll
                for (int u = 0; u < 32; ++u) sn[u] = c.in[(size_t)(t + 32 + u) * n + k];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) { x = xtrace_next<ADD>(x, s[u], c.x_decay, c.x_scale); c.xtr[(size_t)(t + u + 1) * n + k] = x; }
            if (more) {
#pragma unroll
                for (int u = 0; u < 32; ++u) s[u] = sn[u];
            }
        }
    }
    for (; t < c.T; ++t) { x = xtrace_next<ADD>(x, c.in[(size_t)t * n + k], c.x_decay, c.x_scale); c.xtr[(size_t)(t + 1) * n + k] = x; }
    // (the caller's trace tensor is NOT touched here: the resident kernel copies entry T into it in its epilogue, once
    //  the run is known to have succeeded -- a refused or timed-out run must leave every state tensor as it found it)
}


inline size_t dc_prep_lds_bytes(int B, int Nin, int nth) { return (size_t)(B * ((Nin + 31) / 32) + Nin + 4) * 4 + 2 * (nth / 64) * LX * 2; }


}  // namespace

// resident form (snn_dc2015_resident.hip)
size_t snn_dc2015_resident_lds(int B, int Nin, int N, int cw);
size_t snn_dc2015_spec_lds(int B, int Nin, int N);
int snn_dc2015_resident_cw(int N);
int snn_dc2015_resident_nt();
// (ordinary: a plain launch instead of a cooperative one -- co-residency then rests on snn_dc2015_resident_capacity and the in-order stream)
int snn_dc2015_resident_launch(const DcCtx &c, int cw, int nt, size_t lds_bytes, int lean, hipStream_t st, bool ordinary = false);
int snn_dc2015_resident_capacity(int cw, int nt, size_t lds_bytes);
// third-generation lean form (snn_dc2015_async.hip)
size_t snn_dc2015_async_lds(int B, int Nin, int N);
int snn_dc2015_async_capacity(size_t lds_bytes);
int snn_dc2015_async_launch(const DcCtx &c, size_t lds_bytes, hipStream_t st, bool ordinary = false);

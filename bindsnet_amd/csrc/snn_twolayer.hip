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
constexpr int META = 136;       // [0] list entries, [1] active rows, [2] flags (1: spike byte > 1, 2: list overflow), [3] longest per-sample list, [4..4+B] CSR offsets
constexpr int kDigestRegs = 16 * 1024;   // words of a digest entry the run kernel's prefetch registers hold (PF * NT, either workgroup size)

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
    uint32_t *dig; int DW, LCAP, o_ent, o_am, o_ar, o_ab, o_xw, o_ri;   // digest: words per entry, list capacity, word offsets (o_ri: u16 row -> index in the active-row list)
    int late_events;                                     // PostPre family, two event areas: the next entry's events asked for behind phase A (with the row tables) instead of at the top (SNN_TWO_LATE_EVENTS)
    int late_rows;                                       // MSTDP: the row tables asked for behind the first chunk of the dense current phase instead of at the top (SNN_TWO_LATE_ROWS)
    int ent2;                                            // two [meta | event list] areas in LDS (learning instances, where it fits)
    float inv_hwps;
    int prodw;                                           // floats of LDS for the staged products of the dense dot (0: off)
    int mstdp_rows;                                      // MSTDP in its row-per-thread forms (developer switch SNN_TWO_MSTDP_ROWS=0: off)
    int mstdp_burst;                                     // MSTDP burst update (developer switch SNN_TWO_MSTDP_BURST): 2 sample-major where it applies (default), 1 the row walk in its packed forms, 0 the row walk as round 5 left it
    int rowmajor;                                        // PostPre in its row-major form (developer switch SNN_TWO_ROWMAJOR=0: off)
    int use_xsl;                                         // stage the source traces of spiking columns in LDS (fits + Nin <= nt)
    int nt;                                              // threads per workgroup of the run kernel: 1024, or 512 (256 VGPRs per lane: nothing spills)
    // MSTDP (learning.py:1504-1574), factored eligibility: p_plus / p_minus traces, previous-step spike factors
    float *p_plus, *p_minus; uint8_t *s_src_prev, *s_tgt_prev;
    float *pall;                                         // [T+1][B][Nin] p_plus at entry / after every step
    float a_plus, a_minus, d_plus, d_minus, reward; const float *reward_vec;
    int has_norm; float norm; int norm_abs;              // post-run normalisation, done on the LDS tile in the epilogue
    long long *dbg;                                      // developer aid (SNN_TWO_TIMING=1): phase timestamps of workgroup 0
};

#define WMARK() do { if (c.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0) c.dbg[(size_t)8 * 4096 + (size_t)t * 16 + (threadIdx.x >> 6)] = (long long)wall_clock64(); } while (0)
#define SMARK(slot) do { if (c.dbg && blockIdx.x == 0 && threadIdx.x == 0) c.dbg[(size_t)8 * 4096 + (size_t)t * 16 + (slot)] = (long long)wall_clock64(); } while (0)
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
    uint16_t *D_ent = (uint16_t *)(D + c.o_ent), *D_ar = (uint16_t *)(D + c.o_ar), *D_ri = (uint16_t *)(D + c.o_ri);
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
    if (tid == 0) { int mx = 0; cntb[0] = 0; for (int b = 0; b < B; ++b) { mx = max(mx, cntb[b + 1]); cntb[b + 1] += cntb[b]; } misc[2] = mx; }
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
            D_ri[i] = (uint16_t)cp;                       // (rows without a spike keep whatever the workspace held: only active rows are looked up)
            for (int w = 0; w < mw; ++w) D_am[cp * mw + w] = rowmask[i * mw + w];
            atomicOr(&D_ab[i >> 5], 1u << (i & 31));
        }
    }
    __syncthreads();
    if (tid <= B) D[4 + tid] = (uint32_t)cntb[tid];
    if (tid == 0) { D[0] = (uint32_t)total; D[1] = (uint32_t)misc[0]; D[2] = (uint32_t)(misc[1] | (total > c.LCAP ? 2 : 0)); D[3] = (uint32_t)misc[2]; }
}

// ---------------------------------------------------------------------------------------------- the run
#define TWO_NS nt1024
#define TWO_NT 1024
#define TWO_PF 16
#include "snn_twolayer_run.inc"
#undef TWO_NS
#undef TWO_NT
#undef TWO_PF
#define TWO_NS nt512
#define TWO_NT 512
#define TWO_PF 32
#include "snn_twolayer_run.inc"
#undef TWO_NS
#undef TWO_NT
#undef TWO_PF

int digest_layout(TwoCtx &c) {
    c.o_ent = META;
    c.o_am = c.o_ent + c.LCAP / 2;
    c.o_ar = c.o_am + c.Nin * c.MW;
    c.o_ab = c.o_ar + (c.Nin + 1) / 2;
    c.o_xw = c.o_ab + c.NinW;
    c.o_ri = (c.o_xw + c.B * c.NinW + 3) & ~3;
    return (c.o_ri + (c.Nin + 1) / 2 + 3) & ~3;
}

size_t run_lds(const TwoCtx &c) {
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return (size_t)c.Nin * c.CW * 4 + META * 4 + al((size_t)c.LCAP * 2) + (size_t)c.Nin * 4 + al((size_t)c.Nin * 2) +
           (size_t)c.Nin * (c.MW - 1) * 4 +
           al((size_t)c.NinW * 4) + (c.rowmajor ? al((size_t)c.Nin * 2) : 0) + (size_t)c.BC * 8 * 4 + (size_t)16 * c.MW * 4 + (size_t)c.BC * 4 + (size_t)c.BC * 8 * 4 + (size_t)c.prodw * 4 +
           (size_t)(c.T + 2) * 8 +
           (c.use_xsl ? (size_t)c.Nin * c.CW * 4 : 0) + 16 +
           (c.ent2 ? META * 4 + al((size_t)c.LCAP * 2) : 0);
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
    c.mstdp_burst = getenv("SNN_TWO_MSTDP_BURST") ? atoi(getenv("SNN_TWO_MSTDP_BURST")) : 2;
    // list capacity: all events of a step in LDS (u16 each), up to 24 KiB
    c.LCAP = (int)((((size_t)B * Nin < 12288 ? (size_t)B * Nin : 12288) + 7) & ~(size_t)7);
    // widest column tile whose weight slice + digest fit LDS, with a tile thread per (sample, column)
    int cw = 8;
    while (cw > 1) {
        c.CW = cw;
        if (B * cw <= 1024 && run_lds(c) <= 140 * 1024) break;
        cw >>= 1;
    }
    if (const char *e = getenv("SNN_TWO_CW")) {          // (measurement switch: a narrower tile than the widest that fits)
        const int want = atoi(e);
        if ((want == 1 || want == 2 || want == 4) && want < cw) cw = want;
    }
    c.CW = cw;
    if (B * cw > 1024 || run_lds(c) > 140 * 1024) return false;
    // Workgroup size of the run kernel.  At 1024 threads every learning instantiation sits at the 128-VGPR cap with 59 .. 139 spilled registers;
    // at 512 threads (256 VGPRs per lane) nothing spills -- and the kernel is SLOWER all the same (round 6, same box, bit-exact both ways:
    // cfg5 MSTDP 28.3 against 24.7 us per timestep, cfg3 PostPre B=16 / 32 7.64 / 8.50 against 7.06 / 7.40, Hebbian 8.60 against 7.46): the
    // phases are issue-bound loops over rows / events that 1024 threads walk in half the trips, and the spilled registers are touched outside
    // them.  So 1024 it stays; SNN_TWO_NT=512 runs the other instantiation where the tile (B * CW pairs) fits 512 threads.
    c.nt = 1024;
    {
        const char *e = getenv("SNN_TWO_NT");
        if (e && atoi(e) == 512 && B * cw <= 512) c.nt = 512;
    }
    c.G = (N + cw - 1) / cw;
    if (c.T + 1 > 4096) return false;
    c.prodw = 0;
    if (C[0].kind == SNN_CONN_DENSE && B * cw <= 64) {      // (the largest staging area that fits: a chunk costs two barriers and a ramp whatever its length)
        const int sizes[9] = {16384, 12288, 8192, 6144, 5120, 4096, 3072, 2048, 0};
        for (int k = 0; k < 9; ++k) { c.prodw = sizes[k]; if (run_lds(c) <= 140 * 1024) break; }
        if (c.prodw / (B * cw) < 36) c.prodw = 0;           // (a chunk of at least 32 terms per pair)
        if (getenv("SNN_TWO_PRODW")) { c.prodw = atoi(getenv("SNN_TWO_PRODW")); if (run_lds(c) > 150 * 1024) c.prodw = 0; }
    }
    c.use_xsl = 0;
    if (Nin <= c.nt && (c.rule == SNN_RULE_POSTPRE || c.rule == SNN_RULE_HEBBIAN || c.rule == SNN_RULE_WDPOSTPRE) && !(c.rowmajor && ((size_t)Nin * N) % 32 == 0)) { c.use_xsl = 1; if (run_lds(c) > 140 * 1024) c.use_xsl = 0; }
    c.late_events = getenv("SNN_TWO_LATE_EVENTS") ? atoi(getenv("SNN_TWO_LATE_EVENTS")) : 1;   // (same box, cfg3 B=128: 83.9 against 82.9 k timesteps/s)
    c.late_rows = getenv("SNN_TWO_LATE_ROWS") ? atoi(getenv("SNN_TWO_LATE_ROWS")) : 1;     // (same box, cfg5: 71.1 against 67.8 k timesteps/s)
    c.ent2 = 0;
    if (c.rule != SNN_RULE_NONE && c.rule != SNN_RULE_MSTDP && c.MW > 1 && !(getenv("SNN_TWO_ENT2") && atoi(getenv("SNN_TWO_ENT2")) == 0)) { c.ent2 = 1; if (run_lds(c) > 140 * 1024) c.ent2 = 0; }
    // the per-step digest copy must fit the prefetch registers
    if (META + c.LCAP / 2 + Nin * c.MW + (Nin + 1) / 2 + c.NinW > kDigestRegs) return false;
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
#define TWO_VARIANTS(NS) \
            (const void *)NS::k_two_run<false, SNN_RULE_HEBBIAN, 1>, (const void *)NS::k_two_run<false, SNN_RULE_WDPOSTPRE, 1>, \
            (const void *)NS::k_two_run<false, SNN_RULE_HEBBIAN, 4>, (const void *)NS::k_two_run<false, SNN_RULE_WDPOSTPRE, 4>, \
            (const void *)NS::k_two_run<true, SNN_RULE_NONE, 1>, (const void *)NS::k_two_run<true, SNN_RULE_POSTPRE, 1>, \
            (const void *)NS::k_two_run<false, SNN_RULE_NONE, 1>, (const void *)NS::k_two_run<false, SNN_RULE_POSTPRE, 1>, \
            (const void *)NS::k_two_run<false, SNN_RULE_MSTDP, 1>, (const void *)NS::k_two_run<true, SNN_RULE_MSTDP, 1>, \
            (const void *)NS::k_two_run<true, SNN_RULE_NONE, 4>, (const void *)NS::k_two_run<true, SNN_RULE_POSTPRE, 4>, \
            (const void *)NS::k_two_run<false, SNN_RULE_NONE, 4>, (const void *)NS::k_two_run<false, SNN_RULE_POSTPRE, 4>, \
            (const void *)NS::k_two_run<false, SNN_RULE_MSTDP, 4>, (const void *)NS::k_two_run<true, SNN_RULE_MSTDP, 4>
        const void *variants[34] = {TWO_VARIANTS(nt1024), TWO_VARIANTS(nt512),
                                    (const void *)nt1024::k_two_run<false, SNN_RULE_POSTPRE, 1, 16>, (const void *)nt1024::k_two_run<false, SNN_RULE_POSTPRE, 1, 32>};
#undef TWO_VARIANTS
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
        const dim3 grid(c.G), blk(c.nt);
        const size_t lds = run_lds(c);
#define TWO_LAUNCH(NS, MWV) do { \
        if (c.cascade && c.rule == SNN_RULE_POSTPRE) hipLaunchKernelGGL((NS::k_two_run<true, SNN_RULE_POSTPRE, MWV>), grid, blk, lds, st, c); \
        else if (c.cascade && c.rule == SNN_RULE_MSTDP) hipLaunchKernelGGL((NS::k_two_run<true, SNN_RULE_MSTDP, MWV>), grid, blk, lds, st, c); \
        else if (c.cascade) hipLaunchKernelGGL((NS::k_two_run<true, SNN_RULE_NONE, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_POSTPRE) hipLaunchKernelGGL((NS::k_two_run<false, SNN_RULE_POSTPRE, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_MSTDP) hipLaunchKernelGGL((NS::k_two_run<false, SNN_RULE_MSTDP, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_HEBBIAN) hipLaunchKernelGGL((NS::k_two_run<false, SNN_RULE_HEBBIAN, MWV>), grid, blk, lds, st, c); \
        else if (c.rule == SNN_RULE_WDPOSTPRE) hipLaunchKernelGGL((NS::k_two_run<false, SNN_RULE_WDPOSTPRE, MWV>), grid, blk, lds, st, c); \
        else hipLaunchKernelGGL((NS::k_two_run<false, SNN_RULE_NONE, MWV>), grid, blk, lds, st, c); } while (0)
        // (same-box A/B, round 6: B = 16 / 32 PostPre 143.1 -> 145.8 k / 136.2 -> 137.6 k timesteps/s; B = 128 unchanged and MSTDP at B = 16 18 % SLOWER
        //  -- 210 spilled registers against 94 -- so those two instances are not built)
        static const bool bk_env = !(getenv("SNN_TWO_BCONST") && atoi(getenv("SNN_TWO_BCONST")) == 0);       // (measurement switch: 0 = the general instances)
        const bool bk = bk_env && c.nt == 1024 && !c.cascade;
        if (bk && c.rule == SNN_RULE_POSTPRE && c.MW == 1 && c.B == 16) hipLaunchKernelGGL((nt1024::k_two_run<false, SNN_RULE_POSTPRE, 1, 16>), grid, blk, lds, st, c);
        else if (bk && c.rule == SNN_RULE_POSTPRE && c.MW == 1 && c.B == 32) hipLaunchKernelGGL((nt1024::k_two_run<false, SNN_RULE_POSTPRE, 1, 32>), grid, blk, lds, st, c);
        else
        if (c.nt == 512) { if (c.MW == 1) TWO_LAUNCH(nt512, 1); else TWO_LAUNCH(nt512, 4); }
        else { if (c.MW == 1) TWO_LAUNCH(nt1024, 1); else TWO_LAUNCH(nt1024, 4); }
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
        if (mstdp && c.prodw)      // (the MSTDP instance has no phase A: thread 0 marks the dense current phase in slots 1 .. 8 instead)
            fprintf(stderr, "[twolayer timing] dense current phase, us from the step's phase A mark: [1] set up | first chunk [2] staged [3] barrier [4] summed [5] barrier | last chunk [6] staged [7] barrier [8] summed:");
        else
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

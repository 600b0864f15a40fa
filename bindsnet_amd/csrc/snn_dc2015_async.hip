// snn_dc2015_async.hip -- third-generation lean form of the resident DiehlAndCook2015 plan (k_dc2015_async).
//
// Same graph (bindsnet/models/models.py:156-244), same arithmetic, same order of every floating-point operation as the kernels in
// snn_dc2015_resident.hip (read that file's header first); what changed is WHO WAITS FOR WHOM.
//
// The first two generations run all workgroups in lock step: every timestep ends with an all-to-all spike exchange that everybody
// waits for (5.5 / 7.0 us per step at cfg2).  But of "finish step t-1, start step t" almost nothing needs the other workgroups'
// spikes of step t-1:
//   * membrane potential, refractory counter and theta of an Ae neuron follow its PRE-arbitration crossings (nodes.py:1088-1097: the
//     reset, the refractory period and theta are applied before torch.multinomial picks the one spike) -- local;
//   * the inhibitory current of step t comes from the Ai spikes of step t-1, which are the Ae WINNERS of step t-2 (Ai_j is driven by
//     Ae_j alone: own slice of the Ae -> Ai weights diagonal; an Ai neuron that does not follow its Ae partner ends the launch with
//     SNN_ERR_RETRY) -- known one whole step earlier;
//   * only a pair that crossed at step t-1 has to know whether it WON (Ae trace, the post-synaptic PostPre term of its column, its
//     Ae -> Ai current).
// So a workgroup publishes its crossings of step t and goes on with step t+1; the winners of step t are needed by everybody only for
// the membrane update of step t+2, and at once only by the few workgroups (~6 of 100 per step at cfg2) that crossed.
//
// Roles (one cooperative launch, one workgroup per CU, 512 threads = 8 waves: 256 VGPRs per lane, nothing spills):
//   workgroups 0 .. G-1   COMPUTE: 4 columns x 32 samples each, weights / state in LDS as before.  No generator, no polling waves,
//                         no software barriers: three s_barrier per step (probe: 88-140 cycles each against 312-448 for an LDS-counter
//                         barrier among a subset of the waves, tools/probe_latency.hip).  Per step: [own crossing at t-1: wait for the
//                         winners of t-1] trace | PostPre(t-1) in place | X currents of step t | membrane update (inhibition from the
//                         winners of t-2), publish the crossings of step t as ONE summary granule per tile wave (ring of four steps).
//   workgroup G           ARBITER: one wave polls all crossing granules of a step, decodes them, decides every sample's winner -- a
//                         sample with one crossing needs no draw; the others take the draw-comparison arbitration of the lean forms on
//                         the ONE generator copy of the launch (a second wave twists ahead into a 32-block ring; 0.375 us per block) --
//                         and publishes the winners as tagged 8-byte granules (3 {sample, column} entries each; ring of eight steps).
//                         It owns the generator position and writes the generator back.
//   workgroups G+1 ..     RASTER writers (when spike monitors are attached): read the winners granules, write whole [N]-byte raster
//                         rows of both layers; the arbiter does not overrun them (progress words).
// Hand-offs are the 8-byte {tag, data} granules of the earlier forms (one relaxed agent-scope store / load each; the data is the
// flag).  Ring depths: a compute workgroup publishes step t+4 only after the winners of t+2 exist, i.e. after the arbiter has read
// every granule of step t+2 >= t; the arbiter publishes step e+8 only after every raster writer has reported step e+2 done.
//
// State is written back only behind the arbiter's COMMIT granule ("step T"): every step published without an abort mark, every raster
// writer through its last step -- by then nobody can give up any more, so a launch that ends with SNN_ERR_RETRY / SNN_ERR_TIMEOUT has left
// every state tensor (and the generator) untouched, as before.
// Inputs the lean forms do not take (multi-valued spike bytes, > 63 events in a sample: k_dc2015_prep's tbad word), more than one
// entry spike per sample, off-diagonal Ae -> Ai weights: refused up front / on first sight with SNN_ERR_RETRY.
#include <limits.h>
#include <stdlib.h>
#include "snn_dc2015.hpp"
#include "snn_dc2015_tile.hpp"

namespace {

constexpr int ACW = 4;                 // columns per compute workgroup
constexpr int ANT = 512;               // threads per workgroup (all roles)
constexpr int kCrossRing = 4;          // steps of crossing granules in flight (ex / exs)
constexpr int kWinRing = 8;            // steps of winners granules in flight
constexpr int kWinGr = 11;             // winners granules per step: ceil(MAXB / 3)
constexpr int kArbRing = 32;           // generator blocks resident in the arbiter's LDS
constexpr unsigned kAPoll = 400000u;   // bounded polls (~0.5 s), then SNN_ERR_TIMEOUT
constexpr unsigned kAbortPay = 0xFFFFFFFEu;   // crossing-granule payload of a workgroup that gives up

__device__ __forceinline__ uint32_t win_tag(int e) { return (uint32_t)(e + 1) & 1023u; }

// The status word keeps the FIRST error of a launch: a later SNN_ERR_TIMEOUT of somebody who waited in vain for a workgroup that had already
// given up must not hide that workgroup's SNN_ERR_RETRY (the host picks the plan of the second attempt by it).
__device__ __forceinline__ void report(int *status, int code) {
    if (status) atomicCAS(status, 0, code);
}

// The kernel context for a RARELY taken path: the same struct through a pointer the compiler cannot see through, so that the fields
// read through it are loaded where they are used (scalar loads from the kernel-argument segment) instead of being kept -- spilled --
// in scalar registers across the whole step loop.
// (DcCtx is the kernel's only argument: it sits at offset 0 of the kernel-argument segment; taking &c instead made the compiler keep a
//  568-byte private copy.)
__device__ __forceinline__ const DcCtx &cold(const DcCtx &) {
    const __attribute__((address_space(4))) DcCtx *p = (const __attribute__((address_space(4))) DcCtx *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const DcCtx *)p;
}

// ---- LDS layout of a compute workgroup (bytes) -----------------------------------------------------------------------------------
constexpr int AT = MAXB * ACW;         // tile threads
constexpr size_t OC_CTL = 0, OC_XNU0 = 128, OC_XNU0S = OC_XNU0 + 2 * AT * 4, OC_CURX = OC_XNU0S + AT * 4, OC_CURXW = OC_CURX + 2 * AT * 4,
                 OC_SPFIN = OC_CURXW + AT * 4, OC_THC = OC_SPFIN + AT * 4, OC_COLM = OC_THC + 32, OC_COLX = OC_COLM + 32, OC_COLRES = OC_COLX + 32,
                 OC_XWINV = OC_COLRES + 16, OC_XCNT = OC_XWINV + 16 + 16, OC_W0 = OC_XCNT + 2 * MAXB * 4, OC_C0 = OC_W0 + 2 * MAXB * 4, OC_WT = OC_C0 + 2 * MAXB * 4;
static_assert(OC_WT % 16 == 0 && OC_XNU0 % 16 == 0 && OC_XNU0S % 16 == 0, "16-byte aligned float4 arrays");
// ... then wtile [Nin][4] | wieT [N][4] | weiT [N][4] | two digests | wbak [Nin][4] | wwin [Nin][4]
size_t async_compute_lds(int B, int Nin, int N) {
    const int DGS = (digest_lds_words(B, Nin) + 63) & ~63;
    return OC_WT + (size_t)3 * Nin * ACW * 4 + (size_t)2 * N * ACW * 4 + (size_t)2 * DGS * 4;
}
// ---- ... of the arbiter: ctl[32] | cntc[32] | colc[32] | winlist[32] | keys[32] u64 | entry winners [2][32] | crs [B*NW] | mt ring
constexpr size_t OA_CTL = 0, OA_CNT = 128, OA_COL = 256, OA_WL = 384, OA_KEY = 512, OA_CRS = 1280;
size_t async_arbiter_lds(int B, int N) {
    const int BW = (B * ((N + 31) / 32) + 3) & ~3;
    return OA_CRS + (size_t)BW * 4 + (size_t)kArbRing * 624 * 4;
}

// The X -> Ae part of the Ae current of (sample pb, column pq) from the X spikes in the digest `dg` and the weights ws[Nin][4], by
// lane pL of the pair's four threads (k_dc2015_spec's x_current, statement for statement): the value, valid in lane pL == 0.
// DUAL: the same sum over a second weight array ws2 (the won branch of a crossing column) in the same pass -> out2: the event
// lists are read once, only the weight reads and the additions double.
template <bool DUAL>
__device__ __forceinline__ float x_current4(const float *ws, const float *ws2, float &out2, const uint32_t *dg, int B, int Nin, int pb, int pq,
                                            int pL, bool tailcol) {
    constexpr int CW = ACW;
    constexpr uint32_t GM = (1u << GCB) - 1u;
    const uint16_t *lstX = (const uint16_t *)dg;
    const uint32_t *rowmask = dg + B * (LX / 2) + 40;
    const uint16_t *arows = (const uint16_t *)(rowmask + Nin);
    const uint16_t *lst2 = arows + 4 * ((Nin + 1) / 2);
    const uint32_t *gcnt = (const uint32_t *)(lst2 + B * LX);
    const uint32_t *gqn = gcnt + B;
    if (tailcol) {
        const uint32_t gc = gcnt[pb];
        const int st = (pL > 0 ? (int)(gc & GM) : 0) + (pL > 1 ? (int)((gc >> GCB) & GM) : 0) + (pL > 2 ? (int)((gc >> (2 * GCB)) & GM) : 0);
        const int nL = (int)((gc >> (GCB * pL)) & GM);
        const uint16_t *l2 = lst2 + pb * LX;
        const int n4 = Nin >> 2;
        int ix[8]; float wx[8], wy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ix[u] = min((int)l2[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) { wx[u] = ws[ix[u] * CW + pq]; wy[u] = DUAL ? ws2[ix[u] * CW + pq] : 0.f; }
        CascadeFlat a, a2; a.init(); a2.init();
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < nL) { a.add(ix[u] >> 2, wx[u] * 1.0f, n4); if (DUAL) a2.add(ix[u] >> 2, wy[u] * 1.0f, n4); }
        for (int u = 8; u < nL; ++u) {
            const int i = (int)l2[st + u];
            a.add(i >> 2, ws[i * CW + pq] * 1.0f, n4);
            if (DUAL) a2.add(i >> 2, ws2[i * CW + pq] * 1.0f, n4);
        }
        float v = a.finish(n4), vb = DUAL ? a2.finish(n4) : 0.f;
        if (pL == 0) {
            const int s4 = (int)(gc & GM) + (int)((gc >> GCB) & GM) + (int)((gc >> (2 * GCB)) & GM) + (int)((gc >> (3 * GCB)) & GM);
            const int n5 = (int)((gc >> (4 * GCB)) & GM);
            for (int u = 0; u < n5; ++u) {
                const int i = (int)l2[s4 + u];
                v += ws[i * CW + pq] * 1.0f;
                if (DUAL) vb += ws2[i * CW + pq] * 1.0f;
            }
        }
        const float v1 = __shfl_down(v, 1, 4), v2 = __shfl_down(v, 2, 4), v3 = __shfl_down(v, 3, 4);
        const float e1 = ((v + v1) + v2) + v3;
        if (DUAL) {
            const float b1 = __shfl_down(vb, 1, 4), b2 = __shfl_down(vb, 2, 4), b3 = __shfl_down(vb, 3, 4);
            out2 = 0.0f + (((vb + b1) + b2) + b3);
        }
        return 0.0f + e1;
    }
    const uint32_t gq = gqn[pb];
    const int st = (pL > 0 ? (int)(gq & GM) : 0) + (pL > 1 ? (int)((gq >> GCB) & GM) : 0) + (pL > 2 ? (int)((gq >> (2 * GCB)) & GM) : 0);
    const int nL = (int)((gq >> (GCB * pL)) & GM);
    const uint16_t *lx = lstX + pb * LX;
    int ix[8]; float wx[8], wy[8];
    // (no clamp to Nin - 1: the slots behind a sample's events hold 0 -- k_dc2015_prep fills the whole row -- and the slot index
    //  stays inside the row)
#pragma unroll
    for (int u = 0; u < 8; ++u) ix[u] = (int)lx[min(st + u, LX - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { wx[u] = ws[ix[u] * CW + pq]; wy[u] = DUAL ? ws2[ix[u] * CW + pq] : 0.f; }
    // This lane's events all lie in ONE 256-position group of the cascade, so CascadeFlat's third level never closes on them and its
    // result here is a1 + a0: the same additions in the same order with the (value-neutral: 0 + 0) group bookkeeping left out --
    // block cb closes into a1 when the next event's 16-position block differs, then the event is added to a0.
    float a0 = 0.f, a1 = 0.f, c0_ = 0.f, c1_ = 0.f;
    int cb = -1;
    const int nfull = Nin >> 4;
    auto add = [&](int pos, float term, float term2) __attribute__((always_inline)) {
        int blk = pos >> 4;
        blk = blk < nfull ? blk : nfull;
        const bool nb = blk != cb;
        const float s1 = a1 + a0;
        a1 = nb ? s1 : a1;
        a0 = nb ? 0.f : a0;
        a0 += term;
        if (DUAL) {
            const float t1 = c1_ + c0_;
            c1_ = nb ? t1 : c1_;
            c0_ = nb ? 0.f : c0_;
            c0_ += term2;
        }
        cb = blk;
    };
#pragma unroll
    for (int u = 0; u < 8; ++u) if (u < nL) add(ix[u], wx[u] * 1.0f, wy[u] * 1.0f);
    for (int u = 8; u < nL; ++u) {
        const int ii2 = (int)lx[st + u];
        add(ii2, ws[ii2 * CW + pq] * 1.0f, DUAL ? ws2[ii2 * CW + pq] * 1.0f : 0.f);
    }
    const int GL = (Nin >> 4) >> 4;
    auto combine = [&](float G) __attribute__((always_inline)) -> float {
        const float G1 = __shfl_down(G, 1, 4), G2 = __shfl_down(G, 2, 4), G3 = __shfl_down(G, 3, 4);
        const float Gs[4] = {G, G1, G2, G3};
        float A2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < GL) A2 = A2 + Gs[k];
        float Gl = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k == GL) Gl = Gs[k];
        const float res = ((0.0f + Gl) + A2) + 0.0f;
        return 0.0f + res;
    };
    if (DUAL) out2 = combine(c1_ + c0_);
    return combine(a1 + a0);
}

// The X -> Ae currents of sample pb for ALL FOUR columns of the workgroup by lane pL of the sample's four threads (round 6): x_current4's statements with the
// row's four weights read as one float4 and the event bookkeeping (which 16-position block, does it close) shared by the four columns --
// per column the same additions in the same order.  The values, valid in lane pL == 0.
// lane l of a quad reads lane l + k of the same quad (k = 1, 2, 3; meaningful in lane 0): one DPP move instead of a ds_bpermute round trip
template <int K>
__device__ __forceinline__ float quad_down(float v) {
    constexpr int ctrl = K == 1 ? 0x39 : (K == 2 ? 0x0E : 0x03);          // quad_perm [1,2,3,0] / [2,3,0,0] / [3,0,0,0]
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xF, 0xF, true));
}

__device__ __forceinline__ float4 x_current_f4(const float *ws, const uint32_t *dg, int B, int Nin, int pb, int pL, bool tailcol) {
    constexpr uint32_t GM = (1u << GCB) - 1u;
    const uint16_t *lstX = (const uint16_t *)dg;
    const uint32_t *rowmask = dg + B * (LX / 2) + 40;
    const uint16_t *arows = (const uint16_t *)(rowmask + Nin);
    const uint16_t *lst2 = arows + 4 * ((Nin + 1) / 2);
    const uint32_t *gcnt = (const uint32_t *)(lst2 + B * LX);
    const uint32_t *gqn = gcnt + B;
    float r[4];
    if (tailcol) {
        const uint32_t gc = gcnt[pb];
        const int st = (pL > 0 ? (int)(gc & GM) : 0) + (pL > 1 ? (int)((gc >> GCB) & GM) : 0) + (pL > 2 ? (int)((gc >> (2 * GCB)) & GM) : 0);
        const int nL = (int)((gc >> (GCB * pL)) & GM);
        const uint16_t *l2 = lst2 + pb * LX;
        const int n4 = Nin >> 2, nfull = n4 >> 4;
        int ix[8]; float4 wx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ix[u] = min((int)l2[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) wx[u] = *(const float4 *)(ws + ix[u] * 4);
        // CascadeFlat per column (snn_order.hpp), the block / group tests shared
        float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        int cb = -1;
        auto add = [&](int pos, const float4 &w) __attribute__((always_inline)) {
            int blk = pos >> 4;
            blk = blk < nfull ? blk : nfull;
            const bool nb = blk != cb, ng = (blk >> 4) != (cb >> 4);
            const float t[4] = {w.x * 1.0f, w.y * 1.0f, w.z * 1.0f, w.w * 1.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = a1[q] + a0[q];
                a1[q] = nb ? s1 : a1[q];
                a0[q] = nb ? 0.f : a0[q];
                const float s2 = a2[q] + a1[q];
                a2[q] = ng ? s2 : a2[q];
                a1[q] = ng ? 0.f : a1[q];
                a0[q] += t[q];
            }
            cb = blk;
        };
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < nL) add(ix[u] >> 2, wx[u]);
        for (int u = 8; u < nL; ++u) {
            const int i = (int)l2[st + u];
            add(i >> 2, *(const float4 *)(ws + i * 4));
        }
        const bool closeb = cb != nfull, closeg = closeb && (nfull >> 4) != (cb >> 4);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float b0 = a0[q], b1 = a1[q], b2 = a2[q];
            if (closeb) { b1 += b0; b0 = 0.f; if (closeg) { b2 += b1; b1 = 0.f; } }
            v[q] = ((b0 + b1) + b2) + 0.0f;
        }
        if (pL == 0) {
            const int s4 = (int)(gc & GM) + (int)((gc >> GCB) & GM) + (int)((gc >> (2 * GCB)) & GM) + (int)((gc >> (3 * GCB)) & GM);
            const int n5 = (int)((gc >> (4 * GCB)) & GM);
            for (int u = 0; u < n5; ++u) {
                const int i = (int)l2[s4 + u];
                const float4 w = *(const float4 *)(ws + i * 4);
                v[0] += w.x * 1.0f; v[1] += w.y * 1.0f; v[2] += w.z * 1.0f; v[3] += w.w * 1.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v1 = __shfl_down(v[q], 1, 4), v2 = __shfl_down(v[q], 2, 4), v3 = __shfl_down(v[q], 3, 4);
            r[q] = 0.0f + (((v[q] + v1) + v2) + v3);
        }
        return make_float4(r[0], r[1], r[2], r[3]);
    }
    const uint32_t gq = gqn[pb];
    const int st = (pL > 0 ? (int)(gq & GM) : 0) + (pL > 1 ? (int)((gq >> GCB) & GM) : 0) + (pL > 2 ? (int)((gq >> (2 * GCB)) & GM) : 0);
    const int nL = (int)((gq >> (GCB * pL)) & GM);
    const uint16_t *lx = lstX + pb * LX;
    int ix[8]; float4 wx[8];
    // slots behind the lane's events repeat its LAST event's position with a zero term: the same block, so nothing closes, and a0 + 0 = a0 --
    // no lane sits out a slot, the wave runs the eight steps without a branch
    const int last = st + max(nL, 1) - 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) ix[u] = (int)lx[min(st + u, last)];
#pragma unroll
    for (int u = 0; u < 8; ++u) wx[u] = *(const float4 *)(ws + ix[u] * 4);
    // this lane's events lie in ONE 256-position group (x_current4): a1 + a0 per column.  A block change is applied through 0/1 factors
    // inside fmas whose products are exact (x * 1, x * 0): a1 = a0 * diff + a1, a0 = a0 * same + term -- the single rounding of the plain adds
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    int cb = -1;
    const int nfull = Nin >> 4;
    auto add = [&](int pos, const float4 &w, float live) __attribute__((always_inline)) {
        int blk = pos >> 4;
        blk = blk < nfull ? blk : nfull;
        const float same = blk == cb ? 1.f : 0.f, diff = 1.f - same;
        const float t[4] = {w.x * live, w.y * live, w.z * live, w.w * live};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a1[q] = __builtin_fmaf(a0[q], diff, a1[q]);
            a0[q] = __builtin_fmaf(a0[q], same, t[q]);
        }
        cb = blk;
    };
#pragma unroll
    for (int u = 0; u < 8; ++u) add(ix[u], wx[u], u < nL ? 1.0f : 0.0f);
    for (int u = 8; u < nL; ++u) {
        const int i2 = (int)lx[st + u];
        add(i2, *(const float4 *)(ws + i2 * 4), 1.0f);
    }
    const int GL = (Nin >> 4) >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float G = a1[q] + a0[q];
        const float G1 = quad_down<1>(G), G2 = quad_down<2>(G), G3 = quad_down<3>(G);
        const float Gs[4] = {G, G1, G2, G3};
        float A2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < GL) A2 = A2 + Gs[k];
        float Gl = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k == GL) Gl = Gs[k];
        const float res = ((0.0f + Gl) + A2) + 0.0f;
        r[q] = 0.0f + res;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// Entry spikes of both layers (the step before the run): per sample the column of its Ae spike -> w0[b] (the "winners of step -1":
// they drive Ai at step 0 and are what inhibits at step 1) and of its Ai spike -> w0[MAXB + b] ("step -2": what inhibits at step 0);
// -1 = none.  cnt0 counts them: more than one spike in a sample of either layer is outside the lean forms.  All threads of the
// workgroup call it; a barrier behind it.
__device__ __forceinline__ void scan_entry(const DcCtx &c, int *w0, int *cnt0, int tid) {
    const int B = c.B, N = c.N;
    if (tid < 2 * MAXB) { w0[tid] = -1; cnt0[tid] = 0; }
    __syncthreads();
    for (int k = tid; k < B * N; k += ANT) {
        if (c.sE[k]) { const int b = k / N; atomicAdd(&cnt0[b], 1); w0[b] = k - b * N; }
        if (c.sI[k]) { const int b = k / N; atomicAdd(&cnt0[MAXB + b], 1); w0[MAXB + b] = k - b * N; }
    }
    __syncthreads();
}

// Winner column of sample `myb` at step e (-1: none) from the arbiter's granules.  e = -1 / -2: the entry spikes (scan_entry).  bad: set on an
// abort mark or when the poll gives up.  Round 6: the step's kWinGr granules are indexed BY SAMPLE -- granule k holds the entries of samples
// 3k .. 3k+2 (0xFFFF: no winner) --, so a lane loads the one granule of its sample and takes its field: a shift and a compare where the list
// form (round 4: entries in the order of the crossing samples) walked up to six entries per lane in front of every membrane stage -- the
// stretch every crossing chain runs through.  The caller may have loaded the granule earlier (pre: valid if the tag is).
constexpr int kWinPollSleep = 1;       // s_sleep between two polls of the winners granules (other values measured: profiles/r04_async_sensitivity.txt)
struct WinPre { unsigned long long g0; bool have; };
__device__ __forceinline__ WinPre win_prefetch(const DcCtx &c, int e, int myb) {
    WinPre p; p.g0 = granule_load(c.wing + (size_t)(e & (kWinRing - 1)) * kWinGr + myb / 3); p.have = true;
    return p;
}
__device__ __forceinline__ int sample_winner(const DcCtx &c, const int *w0, int e, int myb, bool &bad, WinPre pre = WinPre{0ull, false}) {
    if (e < 0) return w0[(e == -1 ? 0 : MAXB) + myb];
    const unsigned long long *gr = c.wing + (size_t)(e & (kWinRing - 1)) * kWinGr + myb / 3;
    const uint32_t tag = win_tag(e);
    unsigned long long x0 = pre.have ? pre.g0 : granule_load(gr);
    for (unsigned spins = 0; __any((uint32_t)(x0 >> 54) != tag); ++spins) {
        if (spins > kAPoll) { bad = true; report(cold(c).status, SNN_ERR_TIMEOUT); return -1; }
        __builtin_amdgcn_s_sleep(kWinPollSleep);
        x0 = granule_load(gr);
    }
    if (__any((int)((x0 >> 48) & 63u) == 63)) { bad = true; return -1; }       // (an abort mark sits in every granule of the step)
    const uint32_t w = (uint32_t)(x0 >> (16 * (myb % 3))) & 0xFFFFu;
    return w == 0xFFFFu ? -1 : (int)(w & 0x7FFu);
}

// developer aid (SNN_DC_TIMING=<workgroup>): 100 MHz wall-clock marks of one compute workgroup per step, [T+1][24]; behind
// them [T+1][256][4] per workgroup: [1] published, [2] own crossings; slot 255: the arbiter ([0] all granules seen, [1] winners out)
#define AMARKW(k, th) do { if constexpr (TIMING) { if (c.dbg && g == c.dbg_wg && tid == (th)) c.dbg[(size_t)t * 24 + (k)] = (long long)wall_clock64(); } } while (0)
// wave 0's marks are kept in registers and written once per iteration, behind barrier B: a store per mark sat in front of the next
// vmcnt wait (the winners granules) and was measured as part of it
// (SNN_DC_TIMING=-1, "lite": dbg_wg < 0 -- no marks at all, only every workgroup's publish time and crossing counts: the least the instance can disturb)
#define AMARK(k) do { if constexpr (TIMING) { if (wave == 0 && c.dbg_wg >= 0) mk[k] = (long long)wall_clock64(); } } while (0)
// (flushed behind the winners decode of the NEXT iteration: marks 0, 9, 10, 7 are that iteration's by then, the others the previous one's)
// wave 2 (a PostPre wave: the won branch's rows, no resolution) keeps five marks of its own, slots 19..23, written at the top of the next iteration
#define AMARK2(k) do { if constexpr (TIMING) { if (wave == 2 && c.dbg_wg >= 0) mk[k] = (long long)wall_clock64(); } } while (0)
#define AMARK2_FLUSH() do { if constexpr (TIMING) { if (c.dbg && g == c.dbg_wg && tid == 128 && t >= 1) { _Pragma("unroll") for (int k_ = 19; k_ < 24; ++k_) c.dbg[(size_t)(t - 1) * 24 + k_] = mk[k_]; } } } while (0)
#define AMARK_FLUSH() do { if constexpr (TIMING) { if (c.dbg && g == c.dbg_wg && tid == 0) { _Pragma("unroll") for (int k_ = 0; k_ < 19; ++k_) { \
        const bool early_ = k_ == 0 || k_ == 9 || k_ == 10 || k_ == 7 || k_ == 11 || k_ == 18; \
        if (k_ != 3 && k_ != 17 && (early_ || t >= 1)) c.dbg[(size_t)(early_ ? t : t - 1) * 24 + k_] = mk[k_]; } } } } while (0)

// A loop-invariant float parameter into a VECTOR register: kernel arguments are uniform, so the compiler keeps them in scalar
// registers -- of which the compute loop needs far more than the 102 a wave has: the first versions re-read 226 spilled scalars per
// iteration with v_readlane.  These values are operands of vector instructions anyway.
__device__ __forceinline__ float vgpr(float x) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ snn_lif_params vgpr_params(snn_lif_params p) {
    p.decay = vgpr(p.decay); p.rest = vgpr(p.rest); p.reset = vgpr(p.reset); p.thresh = vgpr(p.thresh); p.refrac = vgpr(p.refrac);
    p.dt = vgpr(p.dt); p.lbound = vgpr(p.lbound); p.trace_decay = vgpr(p.trace_decay); p.trace_scale = vgpr(p.trace_scale);
    return p;
}
struct PPar { float nu0, nu1, dt, wmin, wmax; int use_dt, has_min, has_max; };   // PostPre's parameters (floats in vector registers)

// The pre-synaptic PostPre term of one row (its four columns) when no own pair has a FINAL spike: spec_rows4's body
// (snn_dc2015_tile.hpp), statement for statement.  m: samples whose source spiked; xnu0[b][4] = x_tgt * nu0.
__device__ __forceinline__ float4 postpre_row_nowin(const PPar &c, float4 w4, uint32_t m, const float *xnu0) {
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
    return make_float4(w[0], w[1], w[2], w[3]);
}

// One element of PostPre in its (row, column) form (k_dc2015_spec's postpre_elem = stdp_rows_lds's update; lean form: no tail
// elements, 0/1 spikes): row i, column q, pre-synaptic samples m, post-synaptic samples cm; sample bst's x_tgt*nu0 replaced by xw
// when bst >= 0; have_x: one post-synaptic sample, its X trace passed in.
__device__ __forceinline__ float postpre_elem(const PPar &c, int B, int Nin, const float *xnu0, float w, int i, int q, uint32_t m, uint32_t cm,
                                              int bst, float xw, const float *xs, bool have_x, float xval) {
    constexpr int CW = ACW;
    if (c.nu0 != 0.f) {
        float uu = 0.f;
        if (m) {
            CascadeT acc; acc.init(false);
            while (m) {
                const int b = __ffs(m) - 1; m &= m - 1;
                acc.add(b, 1.0f * (b == bst ? xw : xnu0[b * CW + q]), B);
            }
            uu = acc.finish(B);
        }
        if (c.use_dt) uu = uu * c.dt;
        w = w - uu;
    }
    if (c.nu1 != 0.f) {
        float uu = 0.f;
        if (cm) {
            CascadeT acc; acc.init(false);
            while (cm) {
                const int b = __ffs(cm) - 1; cm &= cm - 1;
                acc.add(b, (have_x ? xval : xs[b * Nin + i]) * (1.0f * c.nu1), B);
            }
            uu = acc.finish(B);
        }
        if (c.use_dt) uu = uu * c.dt;
        w = w + uu;
    }
    if (c.has_min && w < c.wmin) w = c.wmin;
    if (c.has_max && w > c.wmax) w = c.wmax;
    return w;
}

// ---- producer workgroups (c.NP > 0): the input-only pre-passes inside the launch ---------------------------------------------------
// Producer p digests entries p, p + NP, ... (dc_prep_entry, snn_dc2015.hpp: k_dc2015_prep's body) and walks the X traces of chunks p,
// p + NP, ... of 256 (sample, source) pairs (xtrace_body: k_dc2015_xtrace's body).  Hand-off to the compute workgroups -- which sit on
// other XCDs, behind other L2s -- by RELEASE / ACQUIRE at agent scope on dready[] (the compiler's cache write-back / invalidate; the
// tagged-granule trick of the step loop does not carry bulk data): a producer's barrier orders its threads' stores before thread 0's
// release; a consumer acquires BEFORE its first load of an entry / of the traces, and only until it has once seen everything finished
// (a few iterations: all producers are done after ~50 us of a ~900 us launch) -- the step loop pays one uniform branch afterwards.
constexpr int kXChunk = 256;
__device__ __forceinline__ int x_chunks(const DcCtx &c) { return c.x_traces ? (c.B * c.Nin + kXChunk - 1) / kXChunk : 0; }

__device__ __forceinline__ void async_producer(const DcCtx &c, unsigned char *smem, int p) {
    const int T = c.T, tid = threadIdx.x;
    for (int e = p; e <= T; e += c.NP) {
        const int flags = dc_prep_entry<ANT>(c, smem, e, c.dig + (size_t)e * c.DW);        // (ends with a barrier)
        if (tid == 0) {
            if (flags & 5) atomicOr(c.tbad, 1);
            __hip_atomic_store(&c.dready[e], (flags & 5) ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&c.dready[T + 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int n = c.B * c.Nin, nch = x_chunks(c);
    for (int ch = p; ch < nch; ch += c.NP) {
        const int k = ch * kXChunk + tid;
        if (tid < kXChunk && k < n) { if (c.x_additive) xtrace_body<true>(c, n, k); else xtrace_body<false>(c, n, k); }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&c.dready[T + 2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (c.rasX) {      // a spike monitor on the Input layer records a copy of the input (monitors.py:94-111): nobody in the launch reads it
        const size_t n16 = ((size_t)T * c.B * c.Nin) >> 4;                  // (Nin % 16 == 0; both pointers 16-byte aligned: checked by the host)
        const uint4 *src = (const uint4 *)c.in;
        uint4 *dst = (uint4 *)c.rasX;
        for (size_t k = (size_t)p * ANT + tid; k < n16; k += (size_t)c.NP * ANT) dst[k] = src[k];
    }
}

// Consumer side.  need_entry: digest entry e may be loaded (1), is one the lean forms do not take (2), or never came (-1: bounded poll).
// `all` is set once every entry has been seen finished (the caller stops asking).  Every lane of the wave runs the same loads.
__device__ __forceinline__ int need_entry(const DcCtx &c, int e, bool &all) {
    // (relaxed polls, ONE acquire fence behind the value that lets the caller go on: an acquire load per poll would invalidate the caches and
    //  drain the wave's outstanding loads every time it is asked -- once per iteration while the producers are still at work)
    const int T = c.T;
    int r;
    if (__hip_atomic_load(&c.dready[T + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == T + 1) {
        all = true;
        r = 0;
    } else {
        for (unsigned spins = 0;; ++spins) {
            r = __hip_atomic_load(&c.dready[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r) break;
            if (spins > kAPoll) return -1;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (all) r = __hip_atomic_load(c.tbad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 2 : 1;
    return r;
}
// need_xtr: the X traces of EVERY step are there (1) or never came (-1).
__device__ __forceinline__ int need_xtr(const DcCtx &c, bool &all) {
    const int nch = x_chunks(c);
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(&c.dready[c.T + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nch) break;
        if (spins > kAPoll) return -1;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    all = true;
    return 1;
}

// ===================================================================================================================== compute
// Threads: tile threads 0..127 <-> (sample tid / 4, column tid % 4): the Ae neuron of the pair, its state in registers; threads
// 384..511 <-> the same pairs: the Ai neuron (its only input is the pair's own final Ae spike); threads 128..383 + 384..511: PostPre.
// Iteration t (two s_barrier: M in the middle, B at the end):
//   B .. M          tile waves: membrane update of step t (X currents prepared one iteration earlier -- the won branch's for a
//                   column that won at t-1 --, inhibition from the winners of t-2), publish the crossings of step t, then the Ae
//                   trace of step t and x_tgt*nu0 of step t+1 as they are WITHOUT a final spike (the trace just decays)
//                   waves 2..5: PostPre of step t on the own slice under "no own final spike at t" (a column that won at t-1
//                   starts from its won branch; the old rows are kept in wbak); waves 6..7: Ai membrane update of step t
//   M .. B          all waves: X currents of step t+1 from the new weights; a workgroup that crossed at step t prepares the won
//                   branch of every crossing column (whole column from the old weights + its X currents) while the arbiter works --
//                   a column with more than one crossing sample waits for the winners and is redone exactly;
//                   then the tile waves: a wave that crossed at step t waits for that step's winners; a winner redoes its Ae trace
//                   and x_tgt*nu0 and marks its column (colmask); every pair leaves its final spike of step t for its Ai thread
// FT (round 6): the tile is FULL -- B == MAXB samples and every workgroup's four columns exist (N % 4 == 0), BASELINE cfg2's shape: every tile / Ai
// thread owns a neuron and B is a constant of the instance: the per-sample loops of the resolution (840 -> 168 instructions), the `mine` tests and
// their exec-mask bookkeeping fall away, 57 scalar registers fewer are spilled.  Same-box A/B, bit-exact on the 114 D&C tests: 894.5 -> 841 us per
// launch at K=20, 850 -> 805 at K=200 (the source count as a constant as well -- Nin = 784 -- was 1.2 % SLOWER: longer unrolled code, 609 spilled scalars).
template <bool TIMING, bool FT>
__device__ __forceinline__ void async_compute(const DcCtx &c, unsigned char *smem) {
    constexpr int CW = ACW, TT = AT, NT = ANT, NTW = TT / 64, SPW = 64 / CW, SPG = 16 / CW, NBC = NT - TT, TI0 = NT - TT;
    const int B = FT ? MAXB : c.B, Nin = c.Nin, N = c.N, T = c.T;
    int *ctl = (int *)(smem + OC_CTL);                       // [0] abort seen (any wave), [1] commit seen
    float *xnu0 = (float *)(smem + OC_XNU0);                 // [2][B][CW] x_tgt * nu0 of a step without an own final spike, by step parity
    float *xnu0s = (float *)(smem + OC_XNU0S);               // [B][CW] ... with the winners of a slow column put in
    float *curX = (float *)(smem + OC_CURX);                 // [2][B][CW] X -> Ae part of the Ae current, by step parity
    float *curXwin = (float *)(smem + OC_CURXW);             // [B][CW] ... of column q in its won branch
    int *spfin = (int *)(smem + OC_SPFIN);                   // [TT] final Ae spike of the pair at step t-1 (for its Ai thread)
    int *thc = (int *)(smem + OC_THC);                       // [2][CW] crossings per own column by step parity (theta)
    uint32_t *colmask = (uint32_t *)(smem + OC_COLM);        // [2][CW] samples with a FINAL spike per own column at step t-1, by the parity of t
    uint32_t *colx = (uint32_t *)(smem + OC_COLX);           // [2][CW] samples with a crossing per own column, by step parity
    uint32_t *colres = (uint32_t *)(smem + OC_COLRES);       // [CW] slow columns: their winners of this step
    float *xwinv = (float *)(smem + OC_XWINV);               // [CW] x_tgt*nu0 of the crossing pair of column q if it wins
    int *xcnt = (int *)(smem + OC_XCNT);                     // [NTW][MAXB] a crossing tile wave's own count of the step's crossings per sample
    int *w0 = (int *)(smem + OC_W0), *cnt0 = (int *)(smem + OC_C0);
    float *wtile = (float *)(smem + OC_WT);                  // [Nin][CW] learned weights
    float *wieT = wtile + (size_t)Nin * CW;
    float *weiT = wieT + (size_t)N * CW;
    const int DGS = (c.DGW + 63) & ~63;
    uint32_t *dgbuf = (uint32_t *)(weiT + (size_t)N * CW);
    float *wbak = (float *)(dgbuf + 2 * DGS);                // [Nin][CW] rows PostPre touched: their weights before
    float *wwin = wbak + (size_t)Nin * CW;                   // [Nin][CW] column q: the whole column with its final spikes of this step

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = (int)blockIdx.x, c0 = g * CW;
    const bool is_ai = tid >= TI0;                            // Ai thread of pair tid - TI0
    const int ptile = is_ai ? tid - TI0 : tid;                // pair of a tile / Ai thread
    const int jj = ptile % CW, bl = ptile / CW, j = c0 + jj;
    const bool colv = FT ? true : j < N, tailcol = c0 >= (N / 32) * 32;
    const bool mine = FT ? tid < TT : (tid < TT && bl < B && colv);             // owns the Ae neuron of a pair
    const bool mine_i = FT ? is_ai : (is_ai && bl < B && colv);                 // owns the Ai neuron of a pair
    const unsigned kst = (unsigned)(bl * N + j);
    const int NGS = c.G * NTW;
    // loop-invariant parameters: floats in vector registers (see vgpr())
    const snn_lif_params pE = vgpr_params(c.pE.lif), pI = vgpr_params(c.pI);
    const float theta_plus = vgpr(c.pE.theta_plus), theta_decay = vgpr(c.pE.theta_decay);
    const PPar pp = {vgpr(c.nu0), vgpr(c.nu1), vgpr(c.dt), vgpr(c.wmin), vgpr(c.wmax), c.use_dt, c.has_min, c.has_max};
    const bool e_learning = c.pE.learning != 0;

    if (tid < 32) ctl[tid] = 0;
    if (tid < 2 * CW) { thc[tid] = 0; colx[tid] = 0; colmask[tid] = 0; colres[tid & 3] = 0; xwinv[tid & 3] = 0.f; }
    scan_entry(c, w0, cnt0, tid);
    bool offdiag = false, multi0 = false;
    for (int k = tid; k < Nin * CW; k += NT) {
        const int i = k / CW, q = k % CW;
        wtile[k] = (c0 + q < N) ? c.Wxe[i * N + c0 + q] : 0.f;
        wbak[k] = 0.f; wwin[k] = 0.f;
    }
    for (int k = tid; k < N * CW; k += NT) {
        const int i = k / CW, q = k % CW;
        wieT[k] = (c0 + q < N) ? c.Wie[i * N + c0 + q] : 0.f;
        const float we = (c0 + q < N) ? c.Wei[i * N + c0 + q] : 0.f;
        weiT[k] = we;
        offdiag = offdiag || (i != c0 + q && we != 0.f);
    }
    if (tid < 2 * MAXB) multi0 = cnt0[tid] > 1;
    if (offdiag || multi0) ctl[0] = 1;                        // (benign race: everybody writes 1)
    // The digest of a step travels global memory -> registers (at the top of an iteration) -> LDS (at its end): an LDS-DMA fetch
    // (global_load_lds) makes the compiler drain vmcnt in front of EVERY later LDS read of the issuing wave -- it cannot tell which
    // LDS bytes the transfer writes --, which put the transfer's whole latency in front of the step (0.6 us per iteration measured)
    // (three named 16-byte registers per thread, loads unconditional with clamped indices: an array captured by a lambda went to scratch)
    const int nchunk = c.DGW >> 2;                                        // 16-byte pieces of a digest entry: <= 3 * 512 (Nin <= 1024)
    uint4 dg0 = make_uint4(0, 0, 0, 0), dg1 = dg0, dg2 = dg0;
#define DIGEST_LOAD(e) do { const uint4 *src_ = (const uint4 *)(c.dig + (size_t)(e) * c.DW); dg0 = src_[min(tid, nchunk - 1)]; \
                            dg1 = src_[min(tid + NT, nchunk - 1)]; dg2 = src_[min(tid + 2 * NT, nchunk - 1)]; } while (0)
#define DIGEST_STORE(e) do { uint4 *dst_ = (uint4 *)(dgbuf + ((e) & 1) * DGS); if (tid < nchunk) dst_[tid] = dg0; \
                             if (tid + NT < nchunk) dst_[tid + NT] = dg1; if (tid + 2 * NT < nchunk) dst_[tid + 2 * NT] = dg2; } while (0)
    // (producer workgroups: entries 0 and 1 are waited for here, behind the weight loads; the later ones at the top of the iteration that asks for them)
    // `pend`: bit 0 = digest entries, bit 1 = the X traces may still be on their way (wave-uniform; kept in a VECTOR register like the float
    // parameters above -- the step loop has no scalar register to spare -- and read with v_readfirstlane where it is asked)
    int pend = 0;
    int dig01 = 1;
    if (c.NP != 0) {
        bool dig_all = false;
        const int r0 = need_entry(c, 0, dig_all), r1 = (T >= 1 && !dig_all) ? need_entry(c, 1, dig_all) : 1;
        dig01 = r0 != 1 ? r0 : r1;
        if (dig01 != 1) { ctl[0] = 1; if (tid == 0) report(c.status, dig01 < 0 ? SNN_ERR_TIMEOUT : SNN_ERR_RETRY); }
        const int p0 = (dig_all ? 0 : 1) | (c.x_traces ? 2 : 0);
        asm volatile("v_mov_b32 %0, %1" : "=v"(pend) : "s"(p0));
    }
#define PENDING(bit) (__builtin_amdgcn_readfirstlane(pend) & (bit))
    if (dig01 >= 0) DIGEST_LOAD(0);
    DIGEST_STORE(0);
    if (T >= 1) { if (dig01 >= 0) DIGEST_LOAD(1); DIGEST_STORE(1); }
    // Inside the loop the digest store would sit in the tail of EVERY iteration -- the stretch that both the ordinary iteration and the
    // crossing chain run through (0.135 us measured).  Its buffer (entry t's) is free during the whole of iteration t, so the six non-tile
    // waves carry the whole entry (four 16-byte pieces per thread: DIGEST_LOAD_E at the top of the iteration) and store it behind their
    // PostPre / Ai work, in front of barrier M; the tile waves neither load nor store it (round 5, first GPU call: bit-exact, 945 vs 950 us
    // per launch on one box, twice; profiles/r05_variants_first_call.txt).
    uint4 dg3 = make_uint4(0, 0, 0, 0);
#define DIGEST_LOAD_E(e) do { const uint4 *src_ = (const uint4 *)(c.dig + (size_t)(e) * c.DW); const int d_ = tid - TT; dg0 = src_[min(d_, nchunk - 1)]; \
                              dg1 = src_[min(d_ + NBC, nchunk - 1)]; dg2 = src_[min(d_ + 2 * NBC, nchunk - 1)]; dg3 = src_[min(d_ + 3 * NBC, nchunk - 1)]; } while (0)
#define DIGEST_STORE_E(e) do { uint4 *dst_ = (uint4 *)(dgbuf + ((e) & 1) * DGS); const int d_ = tid - TT; if (d_ < nchunk) dst_[d_] = dg0; \
                               if (d_ + NBC < nchunk) dst_[d_ + NBC] = dg1; if (d_ + 2 * NBC < nchunk) dst_[d_ + 2 * NBC] = dg2; \
                               if (d_ + 3 * NBC < nchunk) dst_[d_ + 3 * NBC] = dg3; } while (0)
    // state of the own neuron in registers: tile thread -> Ae (v, refractory counter, theta, trace), Ai thread -> Ai (v, counter, trace)
    float r_v = 0.f, r_r = 0.f, r_th = 0.f, x_cur = 0.f, x_before = 0.f;
    bool last_s = false;                                      // last final spike of the own neuron (Ae: redone by a winner; Ai)
    if (mine) {
        r_v = c.vE[kst]; r_r = c.rE[kst]; r_th = c.theta[j];
        x_cur = pE.traces ? c.xE[kst] : 0.f;
        last_s = c.sE[kst] != 0;
    }
    if (mine_i) {
        r_v = c.vI[kst]; r_r = c.rI[kst];
        x_cur = pI.traces ? c.xI[kst] : 0.f;
        last_s = c.sI[kst] != 0;
    }
    if (tid < TT) {
        // x_tgt*nu0 of step 0 without an own final spike: the entry trace, decayed once
        float xn = 0.f;
        if (bl < B && colv && pE.traces) xn = trace_next(x_cur, 0, pE.trace_decay, pE.trace_scale, pE.traces_additive);
        xnu0[tid] = xn * pp.nu0; xnu0[TT + tid] = 0.f; xnu0s[tid] = 0.f;
        curX[tid] = 0.f; curX[TT + tid] = 0.f; curXwin[tid] = 0.f;
        spfin[tid] = (mine && last_s) ? 1 : 0;                // the pair's final Ae spike of the step before the run, for its Ai thread
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool bad = ctl[0] != 0;                                   // this wave has seen a reason to give up (uniform per wave)
    if (bad && tid == 0) report(c.status, SNN_ERR_RETRY);
    // X -> Ae currents of step 0 (from the layer's spikes at entry, digest entry 0)
    for (int qt = tid; qt < B * CW * 4; qt += NT) {
        const int pb = qt / (CW * 4), pq = (qt >> 2) % CW, pL = qt & 3;
        float unused = 0.f;
        const float v = x_current4<false>(wtile, wtile, unused, dgbuf, B, Nin, pb, pq, pL, tailcol);
        if (pL == 0 && c0 + pq < N) curX[pb * CW + pq] = v;
    }
    lds_barrier();                                            // (digest buffer 0 is refilled at the end of the first iteration)
    bool sp_prev = false;                                     // tile thread: final Ae spike of this pair at the previous step
    bool crossed_prev = false;                                // ... and its crossing
    unsigned long long prevE = 0ull;                          // tile waves: crossing ballot of the previous step
    int published = 0;                                        // steps this (tile) wave has published
    WinPre pre_w = WinPre{0ull, false};                       // tile waves: the sample's winners granule asked for ahead of its use
    const bool learn_pp = c.learning && c.rule == SNN_RULE_POSTPRE;

    if constexpr (TIMING) {      // where this workgroup runs: XCC_ID, HW_ID (wave / simd / cu / sh / se) -> the row behind the last step
        if (c.dbg && tid == 0) {
            c.dbg[(size_t)24 * (T + 1) + ((size_t)T * 256 + g) * 4 + 2] = (long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF);
            c.dbg[(size_t)24 * (T + 1) + ((size_t)T * 256 + g) * 4 + 3] = (long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFFF);
        }
    }
    long long mk[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (TIMING only)
    (void)mk;
    bool prev_anyx = true;                                    // the previous iteration had a crossing in this workgroup (uniform)
    // The step loop is instantiated once per ROLE of a wave (round 6): 0 = tile waves 0..1, 1 = waves 2..3 (PostPre + the X-current pass), 2 = waves
    // 4..5 (PostPre + the won branch's pass), 3 = waves 6..7 (Ai + PostPre).  A wave's role never changes, but as ONE loop with `if (wave < ...)`
    // around the roles' pieces every wave carried every role's uniform values through the loop (350-400 spilled scalar registers, a v_readlane
    // in front of each use) and stepped over the other roles' code.  Every copy executes the same barriers.
    auto step_loop = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool R_TILE = ROLE == 0, R_F4 = ROLE == 1, R_W4 = ROLE == 2, R_AI = ROLE == 3, R_NT = ROLE != 0;
    (void)R_W4; (void)R_AI; (void)R_F4;
    for (int t = 0; t <= T; ++t) {
        const bool phaseB = t < T;
        const int par = t & 1;
        const uint32_t *dgn = dgbuf + (par ^ 1) * DGS;                    // digest entry t+1: the X spikes of step t
        const int *meta = (const int *)(dgn + B * (LX / 2));
        const uint32_t *rowmask = dgn + B * (LX / 2) + 40;
        const uint16_t *arows = (const uint16_t *)(rowmask + Nin);
        const bool do_stdp = phaseB && learn_pp;
        const bool full = t == 0;                                         // the first update of a run clamps every element
        AMARK(0);
        AMARK2_FLUSH();
        if constexpr (TIMING) { if (c.dbg && tid == 0 && c.dbg_wg >= 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 0] = (long long)wall_clock64(); }
        // (the winners granules of step t-2, which the membrane stage wants, were asked for at the end of the previous iteration: pre_w)
        // digest entry t+2 -> registers (into LDS at the end of the iteration).  The tile waves issue theirs behind the publish: loads
        // return in order, so waiting for the winners granule in the membrane stage would wait for these (first touch: HBM) as well
        if (t + 2 <= T && R_NT) {
            if (PENDING(1)) {                                             // (producer workgroups: the first iterations only)
                bool all = false;
                const int r = need_entry(cold(c), t + 2, all);
                if (r != 1) { ctl[0] = 1; report(cold(c).status, r < 0 ? SNN_ERR_TIMEOUT : SNN_ERR_RETRY); }
                if (all) pend &= ~1;
            }
            DIGEST_LOAD_E(t + 2);
        }
        // (tile copy: the X current and last step's crossing count of the own pair are asked for together with the abort word -- one LDS round trip in
        //  front of the membrane stage instead of three.  Before the loop had a copy per role this made the kernel 2.7 % SLOWER; now -1 %: 776 -> 768 us)
        float cx_plain = 0.f;
        int thc_prev = 0;
        if (R_TILE) { cx_plain = curX[par * TT + tid]; thc_prev = thc[(par ^ 1) * CW + jj]; }
        if (ctl[0]) { bad = true; break; }                                // (written in front of barrier B)
        AMARK(9);
        uint32_t wonm = 0;                                                // own columns that won at step t-1: their won branch is what happened
        if (learn_pp && prev_anyx) {                                      // (only an iteration behind a crossing can find a bit there: four LDS reads less in front of the
                                                                          //  membrane stage otherwise -- round 6, same-box A/B: 912 -> 896 us per launch)
#pragma unroll
            for (int q = 0; q < CW; ++q) wonm |= (__builtin_amdgcn_readfirstlane((int)colmask[par * CW + q]) != 0 ? 1u : 0u) << q;
        }
        AMARK(10);
        if (!phaseB) {                                                    // behind the last step: only its winners' columns are left to commit
            if (wonm)
                for (int i = tid; i < Nin; i += NT) {
                    float4 v = *(const float4 *)(wtile + i * 4);
                    const float4 ww = *(const float4 *)(wwin + i * 4);
                    if (wonm & 1u) v.x = ww.x;
                    if (wonm & 2u) v.y = ww.y;
                    if (wonm & 4u) v.z = ww.z;
                    if (wonm & 8u) v.w = ww.w;
                    *(float4 *)(wtile + i * 4) = v;
                }
            break;
        }
        if (R_TILE) {

            // ---- Ae membrane update of step t, publish its crossings
            float cx = 0.f;
            if (mine) cx = ((wonm >> jj) & 1u) ? curXwin[bl * CW + jj] : cx_plain;                 // (bl * CW + jj = tid)
            if constexpr (TIMING) { if (c.dbg && g == c.dbg_wg && tid == 0) c.dbg[(size_t)t * 24 + 17] = !pre_w.have ? 2 : ((uint32_t)(pre_w.g0 >> 54) != win_tag(t - 2) ? 1 : 0); }
            const int jI = bad ? -1 : sample_winner(c, w0, t - 2, min(bl, B - 1), bad, pre_w);   // the Ai spike of step t-1 in this sample = the Ae winner of step t-2
            AMARK(7);
            AMARK_FLUSH();
            bool spE = false;
            if (mine) {
                const float e2 = jI >= 0 ? wieT[min(jI, N - 1) * CW + jj] * 1.0f + 0.0f : 0.0f;
                const float curE = cx + e2;                                // (zeros + X->Ae) + Ai->Ae   (network.py:225-248)
                if (e_learning && t >= 1) r_th = r_th + theta_plus * (float)thc_prev;
                if (e_learning) r_th = r_th * theta_decay;
                spE = dc_update(r_v, r_r, curE, pE.thresh + r_th, pE);
                if (spE) atomicAdd(&thc[par * CW + jj], 1);
            }
            const uint64_t mE = __ballot(spE);
            AMARK(12);
            const int slot = t & (kCrossRing - 1);
            uint32_t pay;
            if (bad) pay = kAbortPay;
            else {
                const int nev = __popcll(mE);
                if (nev <= 3) {
                    pay = (uint32_t)nev << 30;
                    int sh = 0;
                    for (uint64_t m = mE; m; m &= m - 1) { pay |= (uint32_t)(__ffsll((unsigned long long)m) - 1) << sh; sh += 8; }
                } else {
                    const int sidx = lane / CW, b = wave * SPW + sidx;
                    const uint32_t v = (uint32_t)((mE >> (sidx * CW)) & 0xFFFFull);
                    if ((lane % CW) == 0 && (sidx % SPG) == 0 && b < B)
                        granule_store(cold(c).ex + (size_t)slot * (cold(c).G * cold(c).KB) + g * cold(c).KB + b / SPG, ((unsigned long long)(uint32_t)(t + 1) << 32) | v);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pay = 0xC0FFFFFFu;
                }
            }

            if (lane == 0) granule_store(c.exs + (size_t)slot * NGS + g * NTW + wave, ((unsigned long long)(uint32_t)(t + 1) << 32) | pay);

            AMARK(8);
            if constexpr (TIMING) { if (c.dbg && tid == 0) { c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 1] = (long long)wall_clock64(); c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 2] = (long long)__popcll(mE); } }
            if constexpr (TIMING) { if (c.dbg && tid == 64 && c.dbg_wg < 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 3] = 0x100 + (long long)__popcll(mE); }   // (lite: tile wave 1's crossings)
            published = t + 1;
            prevE = mE; crossed_prev = spE;
            // Ae trace of step t as it is without a final spike (nodes.py:96-103), x_tgt*nu0 of step t+1 likewise (the front of the
            // next iteration redoes both for a winner of step t)
            if (bl < B) {
                float xn = 0.f;
                if (colv && pE.traces) {
                    x_before = x_cur;
                    x_cur = trace_next(x_before, 0, pE.trace_decay, pE.trace_scale, pE.traces_additive);
                    xn = trace_next(x_cur, 0, pE.trace_decay, pE.trace_scale, pE.traces_additive);
                }
                xnu0[(par ^ 1) * TT + tid] = xn * pp.nu0;
            }
            if (mine) last_s = false;                                      // (a winner of step t is put back in front of A of the next iteration)
            if (spE) {
                atomicOr(&colx[par * CW + jj], 1u << bl);
                if (pE.traces) xwinv[jj] = trace_next(x_before, 1, pE.trace_decay, pE.trace_scale, pE.traces_additive) * pp.nu0;
            }
            if (bad) ctl[0] = 1;
            if (mine && c.rasVE) (cold(c).rasVE + (size_t)t * B * N)[kst] = r_v;
        } else if (R_AI) {
            // ---- Ai membrane update of step t: its input is the pair's own final Ae spike of step t-1 (own slice of the Ae -> Ai
            //      weights diagonal); it must fire exactly when that spike was there -- what everybody's inhibition assumes

            if (mine_i) {
                const bool spA = spfin[ptile] != 0;
                const float e3 = spA ? weiT[j * CW + jj] * 1.0f + 0.0f : 0.0f;
                float ci = 0.0f + e3;                                      // zeros + Ae->Ai
                if (r_r > 0.f) ci = 0.f;
                const bool spIn = lif_update(r_v, r_r, ci, pI);
                last_s = spIn;
                if (pI.traces) x_cur = trace_next(x_cur, spIn, pI.trace_decay, pI.trace_scale, pI.traces_additive);
                if (spIn != spA) {
                    ctl[0] = 1;
                    report(cold(c).status, SNN_ERR_RETRY);
                }
                if (c.rasVI) (cold(c).rasVI + (size_t)t * B * N)[kst] = r_v;
            }
        }
        if (R_NT && do_stdp) {

            // ---- PostPre of step t under "no own final spike at step t" (learning.py / MCC_learning.py:224-302), one thread per
            //      listed row, in place; a column that won at step t-1 enters with its won branch; the row as it was goes to wbak
            //      (the Ai waves join behind their own update: rows beyond the 256 of waves 2..5)
            const int nact = full ? Nin : __builtin_amdgcn_readfirstlane(meta[32]);
            const int ptid = tid - TT;
            const float *xn0 = xnu0 + par * TT;
            for (int k = ptid; k < nact; k += NBC) {
                const int i = full ? k : (int)arows[k];
                const uint32_t m = rowmask[i];
                float4 v = *(const float4 *)(wtile + i * 4);
                if (wonm) {
                    const float4 ww = *(const float4 *)(wwin + i * 4);
                    if (wonm & 1u) v.x = ww.x;
                    if (wonm & 2u) v.y = ww.y;
                    if (wonm & 4u) v.z = ww.z;
                    if (wonm & 8u) v.w = ww.w;
                }
                *(float4 *)(wbak + i * 4) = v;
                *(float4 *)(wtile + i * 4) = postpre_row_nowin(pp, v, m, xn0);
            }
            if (wonm && !full)                                            // ... and the rows this step does not touch take the won column as it is
                for (int i = ptid; i < Nin; i += NBC) {
                    if (rowmask[i] != 0) continue;
                    float4 v = *(const float4 *)(wtile + i * 4);
                    const float4 ww = *(const float4 *)(wwin + i * 4);
                    if (wonm & 1u) v.x = ww.x;
                    if (wonm & 2u) v.y = ww.y;
                    if (wonm & 4u) v.z = ww.z;
                    if (wonm & 8u) v.w = ww.w;
                    *(float4 *)(wtile + i * 4) = v;
                }
            AMARKW(3, TT);
        }
        if (t + 2 <= T && R_NT) DIGEST_STORE_E(t + 2);             // (its buffer, entry t's, was last read before barrier B of the previous iteration)
        AMARK(13);
        lds_barrier();                                                    // ---- M
        AMARK(4);

        if constexpr (TIMING) { if (c.dbg && tid == 0 && c.dbg_wg >= 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 3] = (long long)wall_clock64(); }
        if (tid < CW) { colmask[par * CW + tid] = 0; colx[(par ^ 1) * CW + tid] = 0; thc[(par ^ 1) * CW + tid] = 0; }
        // the winners of step t-1, which the membrane stage of the NEXT iteration wants: asked for now (two loads in flight while the X
        // currents are computed; nothing waits for them before that stage -- asked for at the end of the iteration they sat in front
        // of the digest's LDS stores, which wait for every outstanding load: 0.4 us per iteration, more on some workgroups)
        if (R_TILE) { pre_w.have = false; if (t >= 1 && t + 1 < T) pre_w = win_prefetch(c, t - 1, min(bl, B - 1)); }
        // ---- X -> Ae currents of step t+1 under "nobody of this workgroup won step t" (from wtile, final since PostPre): waves 2..3 start on
        //      them at once -- the pass is the longest piece between the barriers M and B, and it needs nothing of what the other waves read
        //      first (the crossing state, the row count); in a workgroup that crossed, the six other waves do the touched rows meanwhile
        constexpr bool f4w = R_F4;
        if (f4w) {
            const int ptid = tid - TT, pb = ptid >> 2, pL = ptid & 3;
            const float4 v = x_current_f4(wtile, dgn, B, Nin, min(pb, B - 1), pL, tailcol);
            if (pL == 0 && pb < B) *(float4 *)(curX + (par ^ 1) * TT + pb * CW) = v;
        }
        // ---- a workgroup that crossed at step t prepares the WON BRANCH of its crossing columns while the arbiter works: each such column
        //      as it is with its final spike(s) of step t, every row from the old weights.  First the rows this step's X spikes touch --
        //      the only rows the X currents of step t+1 read --, then the currents of both branches in ONE pass, then (waves 2..7,
        //      while the tile waves look after the resolution) the other rows.  A column with several crossing samples has no single
        //      "it won" outcome: it waits for the winners of this step and is done exactly.
        uint32_t xq[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) xq[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)colx[par * CW + q]);
        const bool crossed_wg = do_stdp && (xq[0] | xq[1] | xq[2] | xq[3]) != 0u;
        prev_anyx = (xq[0] | xq[1] | xq[2] | xq[3]) != 0u;
        if (crossed_wg && PENDING(2)) {                                   // (producer workgroups: a crossing in the launch's first ~50 us)
            bool all = false;
            if (need_xtr(cold(c), all) != 1) { ctl[0] = 1; report(cold(c).status, SNN_ERR_TIMEOUT); }
            if (all) pend &= ~2;
        }
        const float *xsrc = crossed_wg ? cold(c).xtr + (size_t)(t + 1) * B * Nin : nullptr;   // X trace after step t
        const float *xn0 = xnu0 + par * TT;
        uint32_t cmq[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) cmq[q] = crossed_wg ? xq[q] : 0u;
        const int nact = !phaseB ? 0 : (full ? Nin : __builtin_amdgcn_readfirstlane(meta[32]));
        // one element of the won branch: row i of column q
        AMARK(14);
        AMARK2(19);
        auto won_elem = [&](int i, int q) __attribute__((always_inline)) {
            const bool single = __popc(xq[q]) == 1;
            const int bst = single ? __ffs(cmq[q]) - 1 : -1;
            const uint32_t m = rowmask[i];
            const float wold = (full || m != 0) ? wbak[i * CW + q] : wtile[i * CW + q];
            wwin[i * CW + q] = single ? postpre_elem(pp, B, Nin, xn0, wold, i, q, m, cmq[q], bst, xwinv[q], xsrc, true, xsrc[bst * Nin + i])
                                      : postpre_elem(pp, B, Nin, xnu0s, wold, i, q, m, cmq[q], -1, 0.f, xsrc, false, 0.f);
        };
        if (crossed_wg) {

            const bool slow = __popc(xq[0]) > 1 || __popc(xq[1]) > 1 || __popc(xq[2]) > 1 || __popc(xq[3]) > 1;
            if (slow) {
                if (wave == 0) {
                    bool b2 = bad;
                    const int jw = b2 ? -1 : sample_winner(c, w0, t, min(lane, B - 1), b2);
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        const uint64_t mm = __ballot(lane < B && jw == c0 + q && jw >= 0);
                        if (lane == 0) colres[q] = (uint32_t)mm;
                    }
                    if (b2) { bad = true; ctl[0] = 1; }
                }
                lds_barrier();                                            // ---- S1
#pragma unroll
                for (int q = 0; q < CW; ++q) if (__popc(xq[q]) > 1) cmq[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)colres[q]);
                if (R_TILE) {
                    float v = xn0[tid];
                    const uint32_t xj = jj == 0 ? xq[0] : (jj == 1 ? xq[1] : (jj == 2 ? xq[2] : xq[3]));
                    const uint32_t cj = jj == 0 ? cmq[0] : (jj == 1 ? cmq[1] : (jj == 2 ? cmq[2] : cmq[3]));
                    if (bl < B && colv && __popc(xj) > 1 && ((cj >> bl) & 1u) && pE.traces)
                        v = trace_next(x_before, 1, pE.trace_decay, pE.trace_scale, pE.traces_additive) * pp.nu0;
                    xnu0s[tid] = v;
                }
                lds_barrier();                                            // ---- S2
            }
            if (!f4w)                                                     // the rows the X currents of step t+1 read (waves 2..3 are on their pass)
                for (int k = R_TILE ? tid : tid - 128; k < nact; k += NT - 128) {
                    const int i = full ? k : (int)arows[k];
#pragma unroll
                    for (int q = 0; q < CW; ++q) if (cmq[q] && c0 + q < N) won_elem(i, q);
                }
            AMARK2(20);
            lds_barrier();                                                // ---- P
        }
        AMARK(15);
        // ---- X -> Ae currents of step t+1: four threads per (sample, column) pair; "nobody of this workgroup won step t" from wtile and,
        //      for a workgroup that crossed, the won branch of its crossing columns from wwin in the same pass
        //      Round 6: the float4 pass (x_current_f4: four lanes per SAMPLE, the row's four weights as one float4, the block bookkeeping shared
        //      by the columns, DPP instead of LDS shuffles) on waves 2..3 -- and for a workgroup that crossed on waves 4..5 from wwin -- instead of
        //      four lanes per (sample, column) pair on all eight waves: fewer instructions on two waves than the old pass had on every wave, and
        //      the tile waves go straight to their resolution (same-box A/B, bit-exact: 947 -> 900 us per launch, profiles/NOTES_r06.md).
        if (crossed_wg && R_W4) {
            const int ptid = tid - TT - 128, pb = ptid >> 2, pL = ptid & 3;
            const float4 v = x_current_f4(wwin, dgn, B, Nin, min(pb, B - 1), pL, tailcol);     // (its other columns are of no interest)
            if (pL == 0 && pb < B) {
                if (cmq[0] && c0 + 0 < N) curXwin[pb * CW + 0] = v.x;
                if (cmq[1] && c0 + 1 < N) curXwin[pb * CW + 1] = v.y;
                if (cmq[2] && c0 + 2 < N) curXwin[pb * CW + 2] = v.z;
                if (cmq[3] && c0 + 3 < N) curXwin[pb * CW + 3] = v.w;
            }
        }
        AMARK(5);
        AMARK2(21);

        if (crossed_wg && R_NT && !full) {
            // the rows this step's X spikes do not touch (read again only by the next iteration's PostPre): by the non-tile waves
            for (int i = tid - TT; i < Nin; i += NBC) {
                if (rowmask[i] != 0) continue;
#pragma unroll
                for (int q = 0; q < CW; ++q) if (cmq[q] && c0 + q < N) won_elem(i, q);
            }
        }
        AMARK(6);
        AMARK2(22);
        // ---- tile waves: which of the own crossings of step t won -- only a wave that had one waits for the arbiter.  A winner redoes
        //      its Ae trace of step t and x_tgt*nu0 of step t+1 and marks its column; every pair leaves its final spike for its Ai thread
        if (R_TILE) {
            sp_prev = false;
            if (prevE != 0ull && !bad) {
                // A pair that is the ONLY crossing of its sample at step t has won: that needs no draw and no arbiter -- the wave looks
                // at the step's crossing granules itself (lane l takes granules l, l + 64, ...; per-sample counts in LDS), one hop
                // instead of two.  Only a sample with several crossings waits for the arbiter's draw comparison.
                int *xc = xcnt + wave * MAXB;

                if (lane < MAXB) xc[lane] = 0;
                {
                    constexpr int PG = 8;                                 // granules per lane: NGS <= 512
                    const unsigned long long *sums = cold(c).exs + (size_t)(t & (kCrossRing - 1)) * NGS;
                    unsigned long long xs[PG];
                    uint32_t need = 0;
#pragma unroll
                    for (int u = 0; u < PG; ++u) { xs[u] = 0ull; if (lane + 64 * u < NGS) need |= 1u << u; }
                    for (unsigned spins = 0;; ++spins) {
#pragma unroll
                        for (int u = 0; u < PG; ++u) if ((need >> u) & 1u) xs[u] = granule_load(sums + lane + 64 * u);
#pragma unroll
                        for (int u = 0; u < PG; ++u) if (((need >> u) & 1u) && (uint32_t)(xs[u] >> 32) == (uint32_t)(t + 1)) need &= ~(1u << u);
                        if (!__any(need != 0u)) break;
                        if (spins > kAPoll) { bad = true; report(cold(c).status, SNN_ERR_TIMEOUT); break; }
                    }
                    bool ab = false;
#pragma unroll
                    for (int u = 0; u < PG; ++u) {
                        const int gi = lane + 64 * u;
                        if (bad || gi >= NGS) continue;
                        const uint32_t pay = (uint32_t)xs[u];
                        if (!pay) continue;
                        if (pay == kAbortPay) { ab = true; continue; }
                        const int w = gi % NTW;
                        if ((pay & 0xFFu) == 0xFFu) {                    // that tile wave sent bit granules (> 3 crossings): any of its samples may have one
                            for (int bs = 0; bs < SPW; ++bs) if (w * SPW + bs < B) atomicAdd(&xc[w * SPW + bs], 2);
                        } else {
                            const int ne = (int)(pay >> 30);
                            for (int e2 = 0; e2 < ne; ++e2) {
                                const int bsm = w * SPW + (int)((pay >> (8 * e2)) & 0x3Fu) / CW;
                                if (bsm < B) atomicAdd(&xc[bsm], 1);
                            }
                        }
                    }
                    if (__any(ab)) bad = true;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int mycnt = (crossed_prev && bl < B) ? xc[bl] : 0;
                int jw = (crossed_prev && mycnt == 1) ? j : -1;
                if (!bad && __any(crossed_prev && mycnt > 1)) {
                    const int ja = sample_winner(c, w0, t, min(bl, B - 1), bad);
                    if (mycnt > 1) jw = ja;
                }
                const bool sp = !bad && crossed_prev && jw == j && bl < B && colv;
                if (sp) {
                    if (pE.traces) {
                        x_cur = trace_next(x_before, 1, pE.trace_decay, pE.trace_scale, pE.traces_additive);
                        if (t + 1 < T) xnu0[(par ^ 1) * TT + tid] = trace_next(x_cur, 0, pE.trace_decay, pE.trace_scale, pE.traces_additive) * pp.nu0;
                    }
                    atomicOr(&colmask[(par ^ 1) * CW + jj], 1u << bl);
                    last_s = true; sp_prev = true;
                }
            }
            spfin[tid] = sp_prev ? 1 : 0;
            if (bad) ctl[0] = 1;
        }
        AMARK(1);

        AMARK(16);
        AMARK2(23);
        lds_barrier();                                                    // ---- B
        AMARK(2);
    }
    };
    if (wave < NTW) step_loop(std::integral_constant<int, 0>{});
    else if (wave < NTW + 2) step_loop(std::integral_constant<int, 1>{});
    else if (wave < NTW + 4) step_loop(std::integral_constant<int, 2>{});
    else step_loop(std::integral_constant<int, 3>{});
    // ---- a tile wave that gives up says so in the granule of the first step it has not published: the arbiter passes it on.  A wave that has
    //      published all T steps sends a FINAL REPORT in the granule of "step T": nothing, or the abort mark -- a reason to give up that shows in
    //      the LAST iteration (an Ai neuron that does not follow its partner at step T-1, a poll of that step's winners that ran out) comes
    //      after the wave's last publish, and the arbiter must not commit over it: every other workgroup would write its state back while this
    //      one returns, and the host would repeat the input on a half-advanced network.  (`bad` is final here: the top of iteration T read the
    //      abort word behind the last barrier B, and nothing writes it afterwards.)
    // (producer workgroups: the epilogue copies the last X traces -- they must be there BEFORE the final report, the last point where a
    //  wave can still give up; by now they have been for ~95 % of the launch)
    if (PENDING(2) && !bad) {
        bool all = false;
        if (need_xtr(cold(c), all) != 1) { bad = true; report(cold(c).status, SNN_ERR_TIMEOUT); }
    }
    if (wave < NTW && lane == 0) {
        const int pstep = (bad && published <= T - 1) ? published : T;
        granule_store(c.exs + (size_t)(pstep & (kCrossRing - 1)) * NGS + g * NTW + wave, ((unsigned long long)(uint32_t)(pstep + 1) << 32) | (bad ? kAbortPay : 0u));
    }
    // ---- commit: the arbiter's commit granule ("step T") has arrived -> every step was published without an abort mark and the raster
    //      writers are through: nobody can give up any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0 && !bad) {
        bool b2 = false;
        (void)sample_winner(c, w0, T, 0, b2);
        if (!b2 && lane == 0) ctl[1] = 1;
    }
    __syncthreads();
    if (ctl[0] != 0 || ctl[1] == 0) return;
    if (c.x_traces) {
        const float *src = c.xtr + (size_t)T * B * Nin;
        for (int k = g * NT + tid; k < B * Nin; k += c.G * NT) c.xX[1][k] = src[k];
    }
    if (mine) {
        float th = r_th;
        if (e_learning) th = th + theta_plus * (float)thc[((T - 1) & 1) * CW + jj];
        c.vE[kst] = r_v; c.rE[kst] = r_r;
        if (bl == 0) c.theta[j] = th;
        if (pE.traces) c.xE[kst] = x_cur;
        c.sE[kst] = last_s;
    }
    if (mine_i) {
        c.vI[kst] = r_v; c.rI[kst] = r_r;
        if (pI.traces) c.xI[kst] = x_cur;
        c.sI[kst] = last_s;
    }
    if (c.has_norm) {          // topology_features.py:250-266 on the own columns, ATen's column-sum order (k_dc2015_spec's epilogue)
        float *bsum = (float *)dgbuf;
        float *sc = xnu0;
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
            if (tid < CW * 4) {
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
}


// ===================================================================================================================== arbiter
// One sample's one_spike arbitration by one wave: the lean forms' draw comparison (k_dc2015_spec's arbitrate_sample) on the ring
// of generator blocks.  tb / off: ring-absolute block and offset of the step's first word; r = the sample's rank among the crossing
// ones.  Returns the winning column (wave-uniform).
__device__ __forceinline__ int arb_sample(const DcCtx &c, const uint32_t *mt, int tb, int off, int r, const uint32_t *crsrow,
                                          unsigned long long *key, int lane) {
    constexpr int RMK = kArbRing - 1;
    const int N = c.N, NW = c.NW;
    const uint32_t bits = lane < NW ? crsrow[lane] : 0u;
    unsigned long long k1 = ~0ull, k2 = ~0ull;
    for (uint32_t bb = bits; bb; bb &= bb - 1) {
        const int jx = lane * 32 + __ffs(bb) - 1;
        const int w0 = off + 2 * (r * N + jx), w1 = w0 + 1;
        const int m0 = w0 / 624, m1 = w1 / 624;
        const uint32_t hi = mt_temper(mt[((tb + m0) & RMK) * 624 + w0 - 624 * m0]);
        const uint32_t lo = mt_temper(mt[((tb + m1) & RMK) * 624 + w1 - 624 * m1]);
        const unsigned long long m = (((unsigned long long)hi << 32) | lo) & ((1ull << 53) - 1ull);
        const unsigned long long kk = (m << 10) | (unsigned long long)jx;
        if (kk < k1) { k2 = k1; k1 = kk; } else if (kk < k2) k2 = kk;
    }
    if (lane == 0) *key = ~0ull;
    if (bits) atomicMin(key, k1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long kmin = *(volatile unsigned long long *)key;
    const unsigned long long mmin = kmin >> 10, zone = mmin + (mmin >> c.zone_shift) + 1ull;
    const bool close = (k1 != ~0ull && k1 != kmin && (k1 >> 10) <= zone) || (k2 != ~0ull && (k2 >> 10) <= zone);
    int win = (int)(kmin & 1023ull);
    if (__any(close)) {
        if (lane == 0) *key = 0ull;
        for (uint32_t bb = bits; bb; bb &= bb - 1) {
            const int jx = lane * 32 + __ffs(bb) - 1;
            const int w0 = off + 2 * (r * N + jx), w1 = w0 + 1;
            const int m0 = w0 / 624, m1 = w1 / 624;
            const float q = exp1_from_words(mt_temper(mt[((tb + m0) & RMK) * 624 + w0 - 624 * m0]),
                                            mt_temper(mt[((tb + m1) & RMK) * 624 + w1 - 624 * m1]));
            const float val = 1.0f / q;
            atomicMax(key, ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        win = (int)(0xFFFFFFFFu - (uint32_t)(*(volatile unsigned long long *)key & 0xFFFFFFFFull));
    }
    return win;
}

template <bool TIMING>
__device__ __forceinline__ void async_arbiter(const DcCtx &c, unsigned char *smem) {
    constexpr int CW = ACW, NTW = AT / 64, SPW = 64 / CW, SPG = 16 / CW, RB = kArbRing, RMK = RB - 1;
    const int B = c.B, N = c.N, NW = c.NW, T = c.T, G = c.G;
    int *ctl = (int *)(smem + OA_CTL);                       // [0] generator head (last block produced), [1] tail (first block still needed), [2] stop
    int *cntc = (int *)(smem + OA_CNT), *colc = (int *)(smem + OA_COL), *winlist = (int *)(smem + OA_WL);
    unsigned long long *keys = (unsigned long long *)(smem + OA_KEY);
    uint32_t *crs = (uint32_t *)(smem + OA_CRS);
    const int BW = B * NW, BWp = (BW + 3) & ~3;
    uint32_t *mt = crs + BWp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KB = c.KB, NG = G * KB, NGS = G * NTW;

    if (tid < 32) { ctl[tid] = 0; cntc[tid] = 0; colc[tid] = 0; }
    for (int k = tid; k < BW; k += ANT) crs[k] = 0;
    for (int k = tid; k < 624; k += ANT) mt[k] = c.rng[0]->mt[k];
    __syncthreads();
    if (wave == 1) {
        // ---- generator: block k (ring slot k & RMK) from block k-1, as far ahead as the ring allows
        for (int k = 1;; ++k) {
            bool stop = false;
            for (;;) {
                if (k - __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < RB) break;
                if (__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop || __hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            asm volatile("" ::: "memory");
            mt_twist_block_wave(mt + ((k - 1) & RMK) * 624, mt + (k & RMK) * 624, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&ctl[0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else if (wave == 0) {
        int pos = __builtin_amdgcn_readfirstlane(c.rng[0]->pos);         // offset of the next word in block tb (0 .. 624)
        int tb = 0;
        long long consumed;
        {
            const long long c0_ = c.rng[0]->consumed;
            consumed = ((long long)__builtin_amdgcn_readfirstlane((int)(c0_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)c0_);
        }
        bool failed = false;
        auto wait_block = [&](int need) __attribute__((always_inline)) -> bool {        // generator head >= need
            for (unsigned spins = 0; __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++spins) {
                if (spins > 40000000u) return false;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            return true;
        };
        // an abort mark for step e AND the two behind it: a compute workgroup reads at most two steps past the one everybody
        // is waiting for (its own resolution of e+1 in front of the inhibition of e), and their ring slots are free (the raster
        // writers are done with step e-6 when step e is published)
        auto publish_abort = [&](int e0) __attribute__((always_inline)) {
            if (lane < 3 * kWinGr) {                                      // (every granule of the three steps: a reader looks at its sample's)
                const int de = lane / kWinGr, k = lane - de * kWinGr;
                granule_store(c.wing + (size_t)((e0 + de) & (kWinRing - 1)) * kWinGr + k,
                              ((unsigned long long)win_tag(e0 + de) << 54) | (63ull << 48) | 0xFFFFFFFFFFFFull);
            }
        };
        // the raster writers must be done with the step whose ring slot step e0 (and an abort mark's two steps behind it) takes: progress >= e0 - 5
        auto raster_wait = [&](int e0, int rp) __attribute__((always_inline)) -> bool {
            bool late = false;
            if (c.NRW > 0 && e0 >= kWinRing - 2 && lane < c.NRW) {
                for (unsigned spins = 0; rp < e0 - (kWinRing - 2) + 1; ++spins) {
                    if (spins > kAPoll) { late = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    rp = __hip_atomic_load(&c.rprog[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return __any(late);
        };
        int e = 0;
        for (; e < T && !failed; ++e) {
            const int slot = e & (kCrossRing - 1);
            const unsigned long long *sums = c.exs + (size_t)slot * NGS;
            const unsigned long long *exr = c.ex + (size_t)slot * NG;
            bool abortseen = false;

            // ---- every crossing granule of step e: lane l takes granules l, l + 64, ... (all its loads in flight at once, the missing
            //      ones asked for again), then decodes them into bit words / per-sample count / a crossing column
            int rp = INT_MAX;                                             // raster writers' progress, read early (used at publish time)
            if (c.NRW > 0 && lane < c.NRW) rp = __hip_atomic_load(&c.rprog[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            constexpr int PG = 8;                                         // granules per lane: NGS <= 512
            unsigned long long xs[PG];
            uint32_t need = 0;
#pragma unroll
            for (int u = 0; u < PG; ++u) { xs[u] = 0ull; if (lane + 64 * u < NGS) need |= 1u << u; }
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
                for (int u = 0; u < PG; ++u) if ((need >> u) & 1u) xs[u] = granule_load(sums + lane + 64 * u);
#pragma unroll
                for (int u = 0; u < PG; ++u) if (((need >> u) & 1u) && (uint32_t)(xs[u] >> 32) == (uint32_t)(e + 1)) need &= ~(1u << u);
                if (!__any(need != 0u)) break;
                if (spins > kAPoll) { failed = true; break; }
            }
            auto event = [&](int bsm, int jx) __attribute__((always_inline)) {
                if (bsm >= B || jx >= N) return;
                atomicOr((unsigned int *)&crs[bsm * NW + (jx >> 5)], 1u << (jx & 31));
                atomicAdd(&cntc[bsm], 1);
                colc[bsm] = jx;
            };
#pragma unroll
            for (int u = 0; u < PG; ++u) {
                const int gi = lane + 64 * u;
                if (failed || gi >= NGS) continue;
                const uint32_t pay = (uint32_t)xs[u];
                if (!pay) continue;
                if (pay == kAbortPay) { abortseen = true; continue; }
                const int gsrc = gi / NTW, w = gi - gsrc * NTW;
                if ((pay & 0xFFu) == 0xFFu) {                            // more than three crossings in that tile wave: its bit granules
                    for (int q = 0; q < SPW / SPG; ++q) {
                        const int k = w * (SPW / SPG) + q;
                        if (k >= KB) break;
                        unsigned long long d = 0;
                        for (unsigned sp2 = 0;; ++sp2) {
                            d = granule_load(exr + gsrc * KB + k);
                            if ((uint32_t)(d >> 32) == (uint32_t)(e + 1)) break;
                            if (sp2 > kAPoll) { failed = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (failed) break;
                        uint32_t be = (uint32_t)d & 0xFFFFu;
                        while (be) { const int p_ = __ffs(be) - 1; be &= be - 1; event(k * SPG + p_ / CW, gsrc * CW + p_ % CW); }
                    }
                } else {
                    const int ne = (int)(pay >> 30);
                    for (int e2 = 0; e2 < ne; ++e2) {
                        const uint32_t ev = (pay >> (8 * e2)) & 0xFFu;
                        const int p_ = (int)(ev & 0x3Fu);
                        event(w * SPW + p_ / CW, gsrc * CW + p_ % CW);
                    }
                }
            }
            failed = __any(failed);
            abortseen = __any(abortseen);
            if constexpr (TIMING) { if (c.dbg && lane == 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)e * 256 + 255) * 4 + 0] = (long long)wall_clock64(); }
            if (failed) { if (lane == 0) report(c.status, SNN_ERR_TIMEOUT); }
            if (failed || abortseen) {
                // pass the abort on: every reader of this step's (and any later) winners sees the mark -- once the raster writers have left
                // the three ring slots it takes (a writer still polling one of them would wait out its bounded poll: 0.5 s)
                (void)raster_wait(e, rp);
                publish_abort(e);
                failed = true;
                break;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- winners: lane b <-> sample b
            const int myc = lane < B ? cntc[lane] : 0;
            int mywin = (lane < B && myc == 1) ? colc[lane] : -1;
            const uint32_t anym = (uint32_t)__ballot(myc > 0);
            const uint32_t multim = (uint32_t)__ballot(myc > 1);
            const int arb_rows = __popc(anym);
            for (uint32_t rem = multim; rem; rem &= rem - 1) {
                const int bsm = __ffs(rem) - 1;
                const int r = __popc(anym & ((1u << bsm) - 1u));
                // blocks in front of this sample's words are done with; its own words must be resident
                const int first = tb + (pos + 2 * r * N) / 624, lastb = tb + (pos + 2 * (r + 1) * N - 1) / 624;
                if (lane == 0) __hip_atomic_store(&ctl[1], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!wait_block(lastb)) { failed = true; break; }
                const int wcol = arb_sample(c, mt, tb, pos, r, crs + bsm * NW, &keys[bsm], lane);
                if (lane == bsm) mywin = wcol;
            }
            if (failed) {
                if (lane == 0) report(c.status, SNN_ERR_TIMEOUT);
                publish_abort(e);
                break;
            }
            // ---- pack: the entry of sample b = (b << 11) | its winner's column, 0xFFFF without one
            if (lane < 32) winlist[lane] = (lane < B && myc > 0) ? ((lane << 11) | (mywin & 0x7FF)) : 0xFFFF;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the raster writers must be done with the step whose ring slot this one takes (progress read at the start of the step)
            if (raster_wait(e, rp)) {
                if (lane == 0) report(c.status, SNN_ERR_TIMEOUT);
                publish_abort(e);
                failed = true;
                break;
            }
            // (granule k <-> samples 3k .. 3k+2; every granule of the slot carries the step's tag and the number of winners)
            if (lane < kWinGr) {
                unsigned long long pl = 0;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int idx = 3 * lane + u;                         // sample
                    const unsigned long long wv = idx < 32 ? (unsigned long long)(uint32_t)winlist[idx] : 0xFFFFull;
                    pl |= wv << (16 * u);
                }
                granule_store(c.wing + (size_t)(e & (kWinRing - 1)) * kWinGr + lane,
                              ((unsigned long long)win_tag(e) << 54) | ((unsigned long long)arb_rows << 48) | pl);
            }
            if constexpr (TIMING) { if (c.dbg && lane == 0) { c.dbg[(size_t)24 * (T + 1) + ((size_t)e * 256 + 255) * 4 + 1] = (long long)wall_clock64(); c.dbg[(size_t)24 * (T + 1) + ((size_t)e * 256 + 255) * 4 + 2] = arb_rows + 100 * __popc(multim); } }
            // ---- the generator moves on by 2 N words per crossing sample (nodes.py:1100-1105: one multinomial row each); clean up
            if (arb_rows) {
                const int endw = pos + 2 * arb_rows * N;                 // offset (from block tb) behind the step's last word
                const int adv = (endw - 1) / 624;
                tb += adv; pos = endw - 624 * adv;                        // pos in 1 .. 624, as the reference leaves it
                consumed += (long long)arb_rows * N;
                if (lane == 0) __hip_atomic_store(&ctl[1], tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (int k = lane; k < BW; k += 64) crs[k] = 0;
                if (lane < 32) cntc[lane] = 0;
            }
        }
        // ---- COMMIT.  Every step has been published without an abort mark, so no compute workgroup can give up any more; when the raster
        //      writers have reported their last step too (none of them ran into a bounded poll) and the generator block the launch ends
        //      in is there, the arbiter publishes the commit granule -- "step T": the tag, no winners -- and only behind it do the
        //      compute workgroups write their state back and the arbiter the generator.  Whatever fails before that leaves an abort mark
        //      instead: nobody writes anything.
        if (!failed && e == T) {
            // the compute workgroups' FINAL REPORTS ("step T" crossing granules): an abort mark among them -> no commit
            {
                const unsigned long long *sums = c.exs + (size_t)(T & (kCrossRing - 1)) * NGS;
                bool ab = false, to = false;
                for (int gi = lane; gi < NGS; gi += 64) {
                    unsigned long long x = granule_load(sums + gi);
                    for (unsigned spins = 0; (uint32_t)(x >> 32) != (uint32_t)(T + 1); ++spins) {
                        if (spins > kAPoll) { to = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                        x = granule_load(sums + gi);
                    }
                    if (to) break;
                    if ((uint32_t)x == kAbortPay) ab = true;
                }
                if (__any(to)) { if (lane == 0) report(c.status, SNN_ERR_TIMEOUT); failed = true; }
                else if (__any(ab)) failed = true;
                if (failed) { (void)raster_wait(T, 0); publish_abort(T); }
            }
        }
        if (!failed && e == T) {
            bool late = false;
            if (c.NRW > 0 && lane < c.NRW) {
                for (unsigned spins = 0; __hip_atomic_load(&c.rprog[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T; ++spins) {
                    if (spins > kAPoll) { late = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (__any(late) || !wait_block(tb)) {
                if (lane == 0) report(c.status, SNN_ERR_TIMEOUT);
                publish_abort(T);
            } else {
                if (lane == 0)
                    granule_store(c.wing + (size_t)(T & (kWinRing - 1)) * kWinGr, ((unsigned long long)win_tag(T) << 54) | 0xFFFFFFFFFFFFull);
                // the generator as the reference leaves it (the one copy of the launch)
                snn_rng_state *wr = c.rng[0];
                for (int k = lane; k < 624; k += 64) wr->mt[k] = mt[(tb & RMK) * 624 + k];
                if (lane == 0) { wr->pos = pos; wr->consumed = consumed; }
            }
        }
        if (lane == 0) __hip_atomic_store(&ctl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ===================================================================================================================== raster
// Raster rows of both layers (monitors.py:94-111: one [B, N] byte slice per step): row r of the 2 B rows of a step by raster
// workgroup r mod NRW, wave by wave.  Ae row (step e, sample b): the winner's byte; Ai row (step e, sample b): the winner of step
// e-1 (the arbiter's winners ARE the Ai spikes one step later; compute workgroups end the launch where that fails).
__device__ __forceinline__ void async_raster(const DcCtx &c, unsigned char *smem, int rid) {
    const int B = c.B, N = c.N, T = c.T;
    int *w0 = (int *)smem, *cnt0 = w0 + 2 * MAXB;            // entry winners
    int *wcur = cnt0 + 2 * MAXB;                              // [2][MAXB] winners by step parity
    int *flag = wcur + 2 * MAXB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NWV = ANT / 64;
    scan_entry(c, w0, cnt0, tid);
    if (tid < MAXB) { wcur[MAXB + tid] = w0[tid]; wcur[tid] = -1; }   // "step -1" sits in the odd slot
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    for (int e = 0; e < T; ++e) {
        const int par = e & 1;
        if (wave == 0) {
            bool bad = false;
            const int wv = sample_winner(c, w0, e, min(lane, B - 1), bad);
            if (lane < MAXB) wcur[par * MAXB + lane] = lane < B ? wv : -1;
            if (bad && lane == 0) flag[0] = 1;
        }
        lds_barrier();
        if (flag[0]) break;
        for (int r = rid + c.NRW * wave; r < 2 * B; r += c.NRW * NWV) {
            const int b = r < B ? r : r - B;
            uint8_t *ras = r < B ? c.rasE : c.rasI;
            const int jw = r < B ? wcur[par * MAXB + b] : wcur[(par ^ 1) * MAXB + b];
            if (ras) { uint8_t *row = ras + ((size_t)e * B + b) * N; for (int jx = lane; jx < N; jx += 64) row[jx] = (uint8_t)(jx == jw); }
        }
        lds_barrier();
        if (tid == 0) __hip_atomic_store(&c.rprog[rid], e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool TIMING, bool FT>
__global__ __launch_bounds__(ANT) void k_dc2015_async(const DcCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int blk = (int)blockIdx.x;
    if (blk == c.stall_wg) return;
    if (c.NP == 0 && *c.tbad != 0) {  // an input the lean forms do not take (pre-pass launches: known up front): refused before anything has happened
        if (blk == 0 && threadIdx.x == 0) report(c.status, SNN_ERR_RETRY);
        return;
    }
    if (blk < c.G) async_compute<TIMING, FT>(c, smem);
    else if (blk == c.G) async_arbiter<TIMING>(c, smem);
    else if (blk < c.G + 1 + c.NRW) async_raster(c, smem, blk - c.G - 1);
    else async_producer(c, smem, blk - c.G - 1 - c.NRW);
}

}  // namespace

size_t snn_dc2015_async_lds(int B, int Nin, int N) {
    const size_t a = async_compute_lds(B, Nin, N), b = async_arbiter_lds(B, N), p = dc_prep_lds_bytes(B, Nin, ANT);
    return a > b ? (a > p ? a : p) : (b > p ? b : p);
}

static bool async_attr_once() {
    static int state = 0;
    if (!state) state = (snn_check(hipFuncSetAttribute((const void *)k_dc2015_async<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) ||
                         snn_check(hipFuncSetAttribute((const void *)k_dc2015_async<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) ||
                         snn_check(hipFuncSetAttribute((const void *)k_dc2015_async<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) ? -1 : 1;
    return state == 1;
}

int snn_dc2015_async_capacity(size_t lds) {
    if (!async_attr_once()) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    // (asked on every run: the answer is kept per device and LDS size -- two attribute queries and an occupancy query otherwise)
    struct Known { int dev; size_t lds; int cap; };
    static thread_local Known known[4] = {{-1, 0, 0}, {-1, 0, 0}, {-1, 0, 0}, {-1, 0, 0}};
    static thread_local int next = 0;
    const char *fake = getenv("SNN_DC_FAKE_CUS");            // (test switch, read per run: never kept)
    const int fake_cus = fake ? atoi(fake) : 0;
    if (!fake_cus) for (const Known &k : known) if (k.dev == dev && k.lds == lds) return k.cap;
    int cus = 0, coop = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess) coop = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_dc2015_async<false, false>, ANT, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (fake_cus) cus = fake_cus;
    const int cap = coop ? cus * per_cu : 0;
    if (!fake_cus) { known[next] = Known{dev, lds, cap}; next = (next + 1) & 3; }
    return cap;
}

// grid = G compute workgroups + the arbiter + c.NRW raster writers + c.NP producers, all co-resident (cooperative launch)
int snn_dc2015_async_launch(const DcCtx &c, size_t lds, hipStream_t st, bool ordinary) {
    if (!async_attr_once()) return SNN_ERR_LAUNCH;
    static const bool coop_env = !(getenv("SNN_DC_COOP") && atoi(getenv("SNN_DC_COOP")) == 0);
    const bool coop = coop_env && !ordinary;
    DcCtx arg = c;
    void *args[1] = {(void *)&arg};
    const unsigned grid = (unsigned)(c.G + 1 + c.NRW + c.NP);
    static const bool ft_env = !(getenv("SNN_DC_FULLTILE") && atoi(getenv("SNN_DC_FULLTILE")) == 0);      // (measurement switch: 0 = the general instance)
    const bool ft = ft_env && c.B == MAXB && c.N % ACW == 0;
    const void *fn = c.dbg ? (const void *)k_dc2015_async<true, false>                                     // (the timing marks are compiled out of the ordinary instances)
                           : (ft ? (const void *)k_dc2015_async<false, true> : (const void *)k_dc2015_async<false, false>);
    if (!coop) return snn_check(hipLaunchKernel(fn, dim3(grid), dim3(ANT), args, lds, st));
    const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(grid), dim3(ANT), args, (unsigned)lds, st);
    if (e == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return SNN_ERR_UNSUPPORTED; }
    return snn_check(e);
}

// snn_dc2015.hip -- fused plan "dc2015-fused": ONE kernel launch per timestep for the
// DiehlAndCook2015 graph (Input -X->Ae (PostPre)-> DiehlAndCookNodes <-> LIFNodes), i.e. the loop body
// of bindsnet/network/network.py:380-461 for the wiring of bindsnet/models/models.py:156-244.
//
// Decomposition.  Workgroup g owns 8 consecutive target columns (neurons c0..c0+7 of BOTH Ae and
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
//
// This file also holds the once-per-run input digest (k_dc2015_prep), the per-step X trace (k_dc2015_xtrace) and the host
// side of BOTH D&C plans; the resident whole-run kernel (the default) lives in snn_dc2015_resident.hip.
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "snn_dc2015.hpp"

unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                                const snn_run_desc *R);
bool snn_prof_begin(int t, hipStream_t st);
bool snn_prof_active();
void snn_prof_end(hipStream_t st);

namespace {

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

// Once per run: digest the X spikes of every step (entry 0 = the layer's `s` at entry, entry e = inputs[e-1]), fully parallel over steps and
// off the per-timestep critical path.  One workgroup per entry (dc_prep_entry, snn_dc2015.hpp; the third-generation lean form runs the same
// body on producer workgroups INSIDE its launch).
__global__ __launch_bounds__(NT) void k_dc2015_prep(const DcCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = blockIdx.x;
    const int flags = dc_prep_entry<NT>(c, smem, e, c.dig + (size_t)e * c.DW);
    if (threadIdx.x == 0 && c.tbad && (flags & 5)) atomicOr(c.tbad, 1);      // an entry the lean resident forms give up on
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
    uint32_t r_dg[4];                                                  // DGW <= 4 * NT (host check)
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int k = tid + u * NT; r_dg[u] = k < c.DGW ? Dg[k] : 0u; }
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
    for (int u = 0; u < 4; ++u) { const int k = tid + u * NT; if (k < c.DGW) dg[k] = r_dg[u]; }
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


// (xtrace_next / xtrace_body: snn_dc2015.hpp -- the producer workgroups of k_dc2015_async run the same body)
__global__ __launch_bounds__(256) void k_dc2015_xtrace(const DcCtx c) {
    const int n = c.B * c.Nin;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    if (c.x_additive) xtrace_body<true>(c, n, k); else xtrace_body<false>(c, n, k);
}

size_t lds_bytes(int B, int Nin, int N) {
    const int NW = (N + 31) / 32;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return al((size_t)digest_lds_words(B, Nin) * 4) + 3 * al((size_t)B * NW * 4) +
           (size_t)MAXB * CW * 4 + 8 * 624 * 4 + NCAND * 4 + MAXB * 8 + 2 * MAXB * LR * 2 + 2 * MAXB * 4 + 2 * 32 * 4 + 32 +
           (size_t)Nin * CW * 4 + 2 * MAXB * CW * 4;
}


}  // namespace

// Device scratch the fused plan needs (bytes); 0 if the graph does not match the plan.
static size_t fused_workspace(int B, int Nin, int N) {
    const int NW = (N + 31) / 32;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return 4 * al((size_t)B * NW * 4) + al((size_t)B * Nin * 4) + al(sizeof(snn_rng_state));
}

// lean forms: summary granules [2][G][tile waves <= 4] with G <= (N + 1) / 2; the third generation keeps FOUR steps of both kinds of
// granules in flight (rings of 4), plus its winners granules [8][11], 16 progress words of the raster writers and the tbad word
static thread_local int g_last_form = -1;   // resident form of this thread's last D&C run (like snn_plan_name()): 0 general, 1 / 2 / 3 lean generations, -1 per-step
static thread_local unsigned long long g_ws_key_in = 0;   // snn_run_desc.host_state[0] as snn_net_run found it (it zeroes the word before any plan runs)
void snn_dc2015_ws_key_in(unsigned long long key) { g_ws_key_in = key; }
constexpr int kAsyncDefault = 1;      // third-generation lean form on by default?  (SNN_DC_ASYNC overrides)
static size_t resident_summary_bytes(int N) { return (size_t)4 * ((N + 1) / 2) * 4 * 8; }
static size_t resident_gran_bytes(int B, int N) { return (size_t)4 * ((N + 1) / 2) * ((B + 1) / 2) * 8; }   // >= 4 * G * KB * 8 for every tile width
constexpr size_t kAsyncCtlBytes = 8 * 11 * 8 + 16 * 4 + 64;

// ... plus two more copies of the granule areas: where the gated second attempt of a pipelined caller exchanges (alternating, so that the
// one it does not use can be cleared for the next run)
static size_t resident_second_bytes(int B, int N) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(resident_gran_bytes(B, N)) + al(resident_summary_bytes(N));
}
static size_t resident_extra(int B, int Nin, int N, int T) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(resident_gran_bytes(B, N)) + al(resident_summary_bytes(N)) + al(kAsyncCtlBytes) + al((size_t)(T + 1 + 16) * 4) + al((size_t)(T + 1) * B * Nin * 4) +
           2 * resident_second_bytes(B, N);
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
    if (digest_lds_words(R->B, L[0].n) > 4 * NT) return false;
    return true;
}

extern "C" int snn_dc2015_last_form(void) { return g_last_form; }

unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R);
unsigned long long snn_convpp_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R);
extern "C" unsigned long long snn_net_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                          const snn_run_desc *R) {
    if (!L || !R || !C) return 0;
    if (nL == 2 && nC == 1) {
        const unsigned long long w = snn_twolayer_workspace_bytes(L, nL, C, nC, R);
        return w ? w : snn_convpp_workspace_bytes(L, nL, C, nC, R);          // (Input -> Conv2d PostPre -> LIF: snn_convlif.hip)
    }
    if (nL != 3 || nC != 3) return 0;
    return fused_workspace_total(R->B, L[0].n, L[1].n, R->T);
}

void snn_set_plan_name(const char *name);
uint8_t *snn_input_raster_request(int layer);      // snn_run.hip: a spike monitor on an Input layer, if the plan wants to serve it itself
void snn_input_raster_done(int layer);
unsigned long long snn_twolayer_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC,
                                                const snn_run_desc *R);
int g_graph_stats[3] = {0, 0, 0};   // runs enqueued as plain launches / captured / replayed

extern "C" void snn_graph_stats(int *h_plain, int *h_captured, int *h_replayed) {
    if (h_plain) *h_plain = g_graph_stats[0];
    if (h_captured) *h_captured = g_graph_stats[1];
    if (h_replayed) *h_replayed = g_graph_stats[2];
}

int snn_try_fused_dc2015(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                         hipStream_t st, int resident, int allow_lean, int *handled, unsigned *normalized) {
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
        c.exs = (unsigned long long *)(p + al2(resident_gran_bytes(B, N)));
        unsigned char *actl = (unsigned char *)c.exs + al2(resident_summary_bytes(N));      // third generation: winners granules, raster progress, tbad
        c.wing = (unsigned long long *)actl;
        c.rprog = (int *)(actl + 8 * 11 * 8);
        c.tbad = nullptr;                                                                  // (set below when that form is chosen)
        c.dready = (int *)(actl + al2(kAsyncCtlBytes));                                    // third generation with producers: [T+1] entry flags, [T+1] / [T+2] counters
        c.xtr = (float *)((unsigned char *)c.dready + al2((size_t)(R->T + 1 + 16) * 4));
        c.status = R->status;
        c.rows4 = !(getenv("SNN_DC_ROWS4") && atoi(getenv("SNN_DC_ROWS4")) == 0);
        c.spec_flags = getenv("SNN_DC_SPECFLAGS") ? atoi(getenv("SNN_DC_SPECFLAGS")) : 0;
        c.stall_wg = getenv("SNN_DC_TEST_STALL") ? atoi(getenv("SNN_DC_TEST_STALL")) : -1;
        c.zone_shift = 19;
        if (getenv("SNN_DC_TEST_ZONE")) { const int z = atoi(getenv("SNN_DC_TEST_ZONE")); if (z >= 0 && z <= 19) c.zone_shift = z; }
        c.has_norm = C[0].has_norm; c.norm = C[0].norm; c.norm_abs = C[0].norm_abs;
    }
    // The resident plan hands spikes between workgroups INSIDE one launch, so every workgroup of the grid must be
    // running at once: the grid is sized against what THIS device holds (CU count x occupancy of the chosen variant,
    // snn_dc2015_resident_capacity), the tile is widened while the grid does not fit, and the launch itself is a
    // cooperative one (the runtime refuses what it cannot make co-resident).  Whatever does not fit takes the
    // one-launch-per-timestep form.  The first PostPre pass (clamp everything) needs the X trace at t = 1.
    if (resident >= 0 && getenv("SNN_DC_RESIDENT")) resident = atoi(getenv("SNN_DC_RESIDENT"));
    if (resident < 0) resident = 0;                                 // (-1: the cooperative launch was just refused)
    int rcw = snn_dc2015_resident_cw(N);
    int rnt0 = snn_dc2015_resident_nt();
    if (!R->status) resident = 0;                                   // nowhere to report a failed hand-off
    if (resident) {
        auto cap = [&](int cw) { return snn_dc2015_resident_capacity(cw, (rnt0 == 512 && cw <= 4) ? 512 : 1024, snn_dc2015_resident_lds(B, Nin, N, cw)); };
        while (rcw < 8 && ((N + rcw - 1) / rcw > cap(rcw) || snn_dc2015_resident_lds(B, Nin, N, rcw) > 150 * 1024)) rcw *= 2;
        if ((N + rcw - 1) / rcw > cap(rcw)) resident = 0;
    }
    const int rG = (N + rcw - 1) / rcw, rKB = (B + 16 / rcw - 1) / (16 / rcw);
    if ((c.rule == SNN_RULE_POSTPRE && !c.x_traces) || snn_dc2015_resident_lds(B, Nin, N, rcw) > 150 * 1024 ||
        resident_extra(B, Nin, N, R->T) > kResidentMaxExtra) resident = 0;
    int rnt = rnt0;
    if (rnt == 512 && (rcw > 4 || B * c.NW > 512)) rnt = 1024;       // single-pass stages of the 512-thread variant
    // lean form of the resident kernel (snn_dc2015_resident.hip): the common case compiled on its own.  Its receive stage
    // relies on the barrier that closes the currents stage (row_sum workgroups, or X currents on the spare threads).
    static const bool lean_on = !(getenv("SNN_DC_LEAN") && atoi(getenv("SNN_DC_LEAN")) == 0);
    const bool lean1 = resident && allow_lean && lean_on && rcw == 4 && rnt == 1024 && L[1].p.one_spike &&
                       ((size_t)Nin * N) % 32 == 0 && Nin <= 1024 && 1024 - MAXB * 4 >= B * 4 * 4;
    // ... and its second generation (k_dc2015_spec, "speculate, then repair"): the default wherever the lean form applies
    // and the extra weight copy fits (developer switch SNN_DC_SPEC=0: first generation)
    static const bool spec_on = !(getenv("SNN_DC_SPEC") && atoi(getenv("SNN_DC_SPEC")) == 0);
    int lean = !lean1 ? 0 : (spec_on && snn_dc2015_spec_lds(B, Nin, N) <= 150 * 1024 ? 2 : 1);
    // ... and the third generation (k_dc2015_async, snn_dc2015_async.hip: compute workgroups that do not wait for each other, one
    // arbiter workgroup, raster writers): wherever the lean form applies, its grid (G + 1 + raster writers) is co-resident and the
    // columns fit the winners' 11-bit field.  SNN_DC_ASYNC=0 / 1 forces it off / on.
    {
        const int async_env = getenv("SNN_DC_ASYNC") ? atoi(getenv("SNN_DC_ASYNC")) : kAsyncDefault;   // (read per run: the tests switch it)
        if (lean && async_env && N <= 1024 && B <= MAXB) {
            const size_t alds = snn_dc2015_async_lds(B, Nin, N);
            const int nrw = (c.rasE || c.rasI) ? 4 : 0;
            const int cap3 = alds <= 150 * 1024 ? snn_dc2015_async_capacity(alds) : 0;
            if (alds <= 150 * 1024 && rG + 1 + nrw <= cap3) {
                lean = 3;
                c.NRW = nrw;
                c.tbad = (int *)((unsigned char *)c.rprog + 16 * 4);
                // the CUs the grid leaves idle run the input-only pre-passes INSIDE the launch (producer workgroups: digest entries, then the
                // X-trace walk; per-entry ready flags): SNN_DC_PRODUCERS=0 / a device too small for >= 16 of them -> the two launches in front
                const char *pe = getenv("SNN_DC_PRODUCERS");
                const int spare = cap3 - (rG + 1 + nrw), want = pe ? atoi(pe) : 128;
                c.NP = (want > 0 && spare >= 16 && dc_prep_lds_bytes(B, Nin, 512) <= alds) ? (spare < want ? spare : want) : 0;
                uint8_t *rq = c.NP > 0 ? snn_input_raster_request(0) : nullptr;
                if (rq && !(((uintptr_t)rq | (uintptr_t)c.in) & 15)) c.rasX = rq;           // the Input layer's raster: copied by the producers
            }
        }
    }
    if (resident) { c.G = rG; c.KB = rKB; }
    static long long *dbg = nullptr;
    static int dbg_T = 0;
    if (getenv("SNN_DC_TIMING")) {
        // [T+1][24] marks of the chosen workgroup, then [T+1][256][4] per-workgroup marks (second-generation lean kernel)
        const size_t dbg_words = (size_t)(24 + 256 * 4) * (R->T + 1);
        if (!dbg || dbg_T < R->T + 1) { if (dbg) (void)hipFree(dbg); (void)hipMalloc(&dbg, sizeof(long long) * dbg_words); dbg_T = R->T + 1; }
        (void)hipMemsetAsync(dbg, 0, sizeof(long long) * dbg_words, st);
        for (int t = 0; t <= R->T; ++t) (void)hipMemsetAsync(dbg + (size_t)t * 24 + 20, 0x7F, sizeof(long long), st);
        c.dbg = dbg; c.dbg_wg = atoi(getenv("SNN_DC_TIMING")); if (c.dbg_wg != -1 && (c.dbg_wg < 0 || c.dbg_wg >= c.G)) c.dbg_wg = c.G - 1;   // (-1: the third generation's "lite" dump, no workgroup carries marks)
    }
    const size_t lds = lds_bytes(B, Nin, N);
    static bool lds_attr = false;
    if (!lds_attr) {   // the kernel may use more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        if (snn_check(hipFuncSetAttribute((const void *)k_dc2015_step, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) return SNN_ERR_LAUNCH;
        lds_attr = true;
    }
    // One run = memset of the exchange words (pad bytes for columns >= N are never written by a workgroup),
    // T+1 launches, and up to two small copies (final X trace sits in xX[(T-1)&1], final generator in rng[T&1]).
    hipStream_t qs = st;           // stream the run is enqueued on (a private one when graphs are in play)
    auto enqueue = [&](bool with_events) -> int {
        int rc0;
        if (resident) {
            // memset of the exchange granules (epochs restart at 1 every run), input-only pre-passes, ONE launch
            const size_t exbytes = (size_t)((unsigned char *)c.xtr - (unsigned char *)c.ex);      // (includes the tbad word)
            // A pipelined caller (status2 + host_state): the gated second attempt behind run k clears this area -- and the copy of the
            // general form's granule areas that run k+1's second attempt will use -- for run k+1, so a section's steady state starts no run
            // with a memset.  host_state[0]: a key of (workspace, layout) while that holds; [1]: runs enqueued that way (which copy is next).
            const size_t secbytes = resident_second_bytes(B, N);
            unsigned char *sec0 = (unsigned char *)c.xtr + al((size_t)(R->T + 1) * B * Nin * 4);
            const bool chain = lean && R->status2 && R->host_state;
            // the key names the layout explicitly -- workspace pointer, B, Nin, N, T, kernel form, bytes -- each field mixed in through a full
            // 64-bit finaliser (no field can cancel another: round 5 XORed shifted fields together); never 0, which means "unknown".
            // snn_net_run has already taken the caller's word away (g_ws_key_in: what it held); it is put back below when this run leaves the area clean.
            auto mix = [](unsigned long long h, unsigned long long v) {
                h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
                h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
                return h;
            };
            unsigned long long key = 0x5EEDull;
            for (unsigned long long v : {(unsigned long long)(uintptr_t)c.ex, (unsigned long long)B, (unsigned long long)Nin, (unsigned long long)N, (unsigned long long)R->T,
                                         (unsigned long long)lean, (unsigned long long)exbytes, (unsigned long long)secbytes})
                key = mix(key, v);
            key |= 1ull;
            const bool clean = chain && g_ws_key_in == key;
            if (!clean) {
                if ((rc0 = snn_check(hipMemsetAsync(c.ex, 0, exbytes, qs)))) return rc0;
                if (chain && (rc0 = snn_check(hipMemsetAsync(sec0, 0, 2 * secbytes, qs)))) return rc0;
            }
            // input-only pre-passes: launches of their own -- unless the third generation runs them on producer workgroups INSIDE its
            // launch (c.NP > 0).  (Both in ONE launch was measured: 38 us against 10 + 22 -- the X-trace walk wants 256-thread workgroups.)
            if (!(lean == 3 && c.NP > 0)) {
                hipLaunchKernelGGL(k_dc2015_prep, dim3(R->T + 1), dim3(NT), dc_prep_lds_bytes(B, Nin, NT), qs, c);
                if (c.x_traces) hipLaunchKernelGGL(k_dc2015_xtrace, dim3((B * Nin + 255) / 256), dim3(256), 0, qs, c);
            }
            const bool prof = with_events && snn_prof_begin(0, qs);
            // A pipelined caller's launches are ORDINARY ones: a cooperative launch costs ~20 us of dispatch latency each (it goes through a
            // queue of its own: 0.989 -> 0.946 ms per step with both launches of a run ordinary, profiles/NOTES_r05.md).  What the cooperative
            // launch guarantees beyond the capacity check made above -- that no kernel of ANOTHER stream of this process holds CUs meanwhile --
            // is part of what a section asks of its caller anyway (a grid that is not co-resident in time ends with SNN_ERR_TIMEOUT, which
            // a section reports as an error); synchronous runs keep the cooperative launch.  SNN_DC_GATED_COOP / SNN_DC_SECTION_COOP=1: measurement switches.
            static const bool section_coop = getenv("SNN_DC_SECTION_COOP") && atoi(getenv("SNN_DC_SECTION_COOP")) != 0;
            const bool ordinary = chain && !section_coop;
            int rcl = lean == 3 ? snn_dc2015_async_launch(c, snn_dc2015_async_lds(B, Nin, N), qs, ordinary)
                                : snn_dc2015_resident_launch(c, rcw, rnt, lean == 2 ? snn_dc2015_spec_lds(B, Nin, N) : snn_dc2015_resident_lds(B, Nin, N, rcw), lean, qs);
            if (prof) snn_prof_end(qs);
            if (rcl == SNN_OK && lean && R->status2) {
                // A pipelined caller does not read the status word back between two runs, so the second attempt of an input the lean form
                // gives up on (SNN_ERR_RETRY: nothing written) is enqueued right behind the first: the general resident kernel on the same
                // digest and X traces, gated on the first attempt's status word, reporting into *status2.  Where the first attempt went
                // through, its workgroups return at once.
                DcCtx c2 = c;
                c2.tbad = nullptr; c2.NRW = 0; c2.NP = 0; c2.rasX = nullptr; c2.status = R->status2; c2.gate = R->status;
                if (chain) {
                    const int par = (int)(R->host_state[1] & 1ull);
                    unsigned char *mine = sec0 + (size_t)par * secbytes, *other = sec0 + (size_t)(par ^ 1) * secbytes;
                    c2.ex = (unsigned long long *)mine;
                    c2.exs = (unsigned long long *)(mine + al(resident_gran_bytes(B, N)));
                    c2.zeroA = (uint4 *)c.ex; c2.zeroA_n16 = (unsigned)(exbytes >> 4);
                    c2.zeroG = (uint4 *)other; c2.zeroG_n16 = (unsigned)(secbytes >> 4);
                } else if ((rc0 = snn_check(hipMemsetAsync(c.ex, 0, exbytes, qs)))) return rc0;
                static const bool gated_coop = getenv("SNN_DC_GATED_COOP") && atoi(getenv("SNN_DC_GATED_COOP")) != 0;
                rcl = snn_dc2015_resident_launch(c2, rcw, rnt, snn_dc2015_resident_lds(B, Nin, N, rcw), 0, qs, !gated_coop && !section_coop);
                if (rcl == SNN_OK && chain) { R->host_state[0] = key; R->host_state[1] += 1; }
            }
            if (rcl == SNN_OK && lean == 3 && c.rasX) snn_input_raster_done(0);
            return rcl;                                     // SNN_ERR_UNSUPPORTED: the runtime refused the cooperative grid
        }
        rc0 = snn_check(hipMemsetAsync(ws, 0, 4 * wb, qs));
        if (rc0) return rc0;
        hipLaunchKernelGGL(k_dc2015_prep, dim3(R->T + 1), dim3(NT), dc_prep_lds_bytes(B, Nin, NT), qs, c);
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
        if (rc == SNN_ERR_UNSUPPORTED && resident)          // grid refused by the runtime: the same run, one launch per timestep
            return snn_try_fused_dc2015(L, nL, C, nC, R, st, -1, 0, handled, normalized);
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
    g_last_form = resident ? lean : -1;
    if (c.dbg && resident && lean == 3) {   // developer aid, third-generation lean kernel
        (void)hipStreamSynchronize(st);
        const int T_ = R->T;
        std::vector<long long> h((size_t)24 * (T_ + 1)), hw((size_t)(T_ + 1) * 256 * 4);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw.data(), dbg + (size_t)24 * (T_ + 1), hw.size() * 8, hipMemcpyDeviceToHost);
        if (const char *dump = getenv("SNN_DC_TIMING_DUMP")) {       // raw marks for offline analysis: [T+1][24] then [T+1][256][4] int64
            if (FILE *f = fopen(dump, "wb")) { fwrite(h.data(), 8, h.size(), f); fwrite(hw.data(), 8, hw.size(), f); fclose(f); }
        }
        double a[24] = {0}, step = 0; int n = 0, na[24] = {0};
        for (int t = 3; t + 1 < T_; ++t, ++n) {
            const long long *r = &h[(size_t)t * 24];
            for (int k = 1; k < 24; ++k) if (r[k] > r[0]) { a[k] += (double)(r[k] - r[0]) / 100.0; na[k]++; }       // (a mark of the crossing path is there only in a crossing iteration)
            step += (double)(h[(size_t)(t + 1) * 24] - r[0]) / 100.0;
        }
        for (int k = 1; k < 24; ++k) a[k] = na[k] ? a[k] / na[k] * n : 0.0;
        fprintf(stderr, "[dc2015 async, us from iteration start, compute workgroup %d] tile waves: winners(t-2) read %.2f, published %.2f | other waves: PostPre done %.2f | "
                        "barrier M %.2f | X currents %.2f | won branch %.2f | resolution of step t %.2f | barrier B %.2f || iteration %.2f us\n",
                c.dbg_wg, a[7] / n, a[8] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n, a[1] / n, a[2] / n, step / n);
        // per step: first / last publish over the compute workgroups, arbiter: all granules seen, winners out
        double spread = 0, seen = 0, out = 0, period = 0, lastx = 0; int m = 0, nx = 0; long long prev_last = 0;
        std::vector<int> lastcnt(c.G, 0);
        for (int t = 3; t + 1 < T_; ++t) {
            long long p0 = hw[((size_t)t * 256) * 4 + 1], p1 = p0; int gl = 0;
            for (int gq = 1; gq < c.G; ++gq) { const long long v = hw[((size_t)t * 256 + gq) * 4 + 1]; if (v < p0) p0 = v; if (v > p1) { p1 = v; gl = gq; } }
            const long long s0 = hw[((size_t)t * 256 + 255) * 4], s1 = hw[((size_t)t * 256 + 255) * 4 + 1];
            spread += (double)(p1 - p0) / 100.0; seen += (double)(s0 - p1) / 100.0; out += (double)(s1 - s0) / 100.0;
            if (prev_last) period += (double)(p1 - prev_last) / 100.0;
            prev_last = p1; ++m; lastcnt[gl]++;
            if (t >= 1 && hw[((size_t)(t - 1) * 256 + gl) * 4 + 2] > 0) { lastx += 1; }
            (void)nx;
        }
        {
            std::vector<std::pair<int, int>> top;
            for (int gq = 0; gq < c.G; ++gq) top.push_back({lastcnt[gq], gq});
            std::sort(top.begin(), top.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first > y.first; });
            fprintf(stderr, "[dc2015 async] workgroups most often the last publisher of a step (of %d steps):", m);
            for (int k = 0; k < 8 && k < (int)top.size(); ++k) fprintf(stderr, " wg%d x%d", top[k].second, top[k].first);
            // how long the last publisher's own iteration was (its publish to its next publish), crossing or not
            double itc = 0, itn = 0; int nc = 0, nn2 = 0;
            for (int t = 4; t + 1 < T_; ++t) {
                long long p1 = 0; int gl = 0;
                for (int gq = 0; gq < c.G; ++gq) { const long long v = hw[((size_t)t * 256 + gq) * 4 + 1]; if (v > p1) { p1 = v; gl = gq; } }
                const double own = (double)(p1 - hw[((size_t)(t - 1) * 256 + gl) * 4 + 1]) / 100.0;
                if (hw[((size_t)(t - 1) * 256 + gl) * 4 + 2] > 0) { itc += own; ++nc; } else { itn += own; ++nn2; }
            }
            fprintf(stderr, " | the last publisher's own step took %.2f us when it had crossed the step before (%d), %.2f us otherwise (%d)\n", nc ? itc / nc : 0.0, nc, nn2 ? itn / nn2 : 0.0, nn2);
        }
        fprintf(stderr, "[dc2015 async per step] last publish - first publish %.2f us | arbiter: last publish -> all granules seen %.2f us, -> winners out +%.2f us | "
                        "period of the last publisher %.2f us | the last publisher had crossed the step before in %.0f %% of the steps\n",
                spread / m, seen / m, out / m, period / (m - 1), 100.0 * lastx / m);
    } else
    if (c.dbg && resident && lean == 2) {   // developer aid, second-generation lean kernel: its own marks (us since the iteration's start)
        (void)hipStreamSynchronize(st);
        std::vector<long long> h((size_t)24 * (R->T + 1));
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double a[8] = {0}, step = 0; int n = 0, nwin = 0; double awin = 0;
        for (int t = 2; t < R->T; ++t, ++n) {
            const long long *r = &h[(size_t)t * 24];
            for (int k = 1; k < 8; ++k) a[k] += (double)(r[k] - r[0]) / 100.0;
            step += (double)(h[(size_t)(t + 1) * 24] - r[0]) / 100.0;
            if (r[12]) { awin += (double)(r[6] - r[5]) / 100.0; ++nwin; }
        }
        {   // per workgroup: barrier R passed / published, relative to the earliest R of the iteration
            const int G = c.G;
            std::vector<long long> hw((size_t)(R->T + 1) * 256 * 4);
            (void)hipMemcpy(hw.data(), dbg + (size_t)24 * (R->T + 1), hw.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> lagR(G, 0.0), chain(G, 0.0), lagP(G, 0.0); int nn = 0;
            for (int t = 2; t + 1 < R->T; ++t, ++nn) {
                long long r0 = hw[((size_t)t * 256) * 4], p0 = hw[((size_t)t * 256) * 4 + 1];
                for (int gq = 1; gq < G; ++gq) { r0 = std::min(r0, hw[((size_t)t * 256 + gq) * 4]); p0 = std::min(p0, hw[((size_t)t * 256 + gq) * 4 + 1]); }
                for (int gq = 0; gq < G; ++gq) {
                    const long long *w = &hw[((size_t)t * 256 + gq) * 4];
                    lagR[gq] += (double)(w[0] - r0) / 100.0; chain[gq] += (double)(w[1] - w[0]) / 100.0; lagP[gq] += (double)(w[1] - p0) / 100.0;
                }
            }
            {   // arrival of each wave of the chosen workgroup at barrier R (us since its iteration start)
                double aw[16] = {0};
                for (int t = 2; t + 1 < R->T; ++t) for (int w = 0; w < 16; ++w) aw[w] += (double)(hw[((size_t)t * 256 + 200 + w) * 4 + 3] - h[(size_t)t * 24]) / 100.0;
                fprintf(stderr, "[dc2015 spec wave arrival at R, workgroup %d, us]", c.dbg_wg);
                for (int w = 0; w < 16; ++w) fprintf(stderr, " w%d %.2f", w, aw[w] / (R->T - 3));
                fprintf(stderr, "\n");
            }
            {   // the workgroups that pass R latest on average, and where they run
                std::vector<int> ord(G); for (int gq = 0; gq < G; ++gq) ord[gq] = gq;
                std::sort(ord.begin(), ord.end(), [&](int x, int y) { return lagR[x] > lagR[y]; });
                fprintf(stderr, "[dc2015 spec per workgroup] mean R lag, top 6:");
                for (int k = 0; k < 6 && k < G; ++k) { const long long id = hw[((size_t)0 * 256 + ord[k]) * 4 + 3]; fprintf(stderr, " wg%d %.2f us (xcc %lld se %lld cu %lld)", ord[k], lagR[ord[k]] / nn, id >> 16, (id >> 13) & 7, (id >> 8) & 15); }
                fprintf(stderr, " | median wg%d %.2f us\n", ord[G / 2], lagR[ord[G / 2]] / nn);
            }
            double mR = 0, mC = 0, mP = 0;
            for (int gq = 0; gq < G; ++gq) { mR += lagR[gq] / nn / G; mC += chain[gq] / nn / G; mP += lagP[gq] / nn / G; }
            fprintf(stderr, "[dc2015 spec per workgroup, us] mean over workgroups: R after the earliest R %.2f | R -> publish %.2f | publish after the earliest publish %.2f\n", mR, mC, mP);
            // the steps' LAST publisher: who, and how late
            std::vector<int> lastcnt(G, 0); double late = 0, lateR = 0, lateC = 0, medC = 0; int hist[8] = {0}; int kinds[3] = {0, 0, 0}, nprep = 0; double klate[3] = {0, 0, 0};
            for (int t = 2; t + 1 < R->T; ++t) {
                int arg = 0; long long pm = 0, p0 = hw[((size_t)t * 256) * 4 + 1], r0 = hw[((size_t)t * 256) * 4];
                std::vector<double> ch(G);
                for (int gq = 0; gq < G; ++gq) { const long long *w = &hw[((size_t)t * 256 + gq) * 4]; if (w[1] > pm) { pm = w[1]; arg = gq; } p0 = std::min(p0, w[1]); r0 = std::min(r0, w[0]); ch[gq] = (double)(w[1] - w[0]) / 100.0; }
                lastcnt[arg]++; late += (double)(pm - p0) / 100.0;
                { const long long kind = hw[((size_t)t * 256 + arg) * 4 + 2]; kinds[(kind & 1) ? 1 : ((kind & 2) ? 2 : 0)]++; klate[(kind & 1) ? 1 : ((kind & 2) ? 2 : 0)] += (double)(hw[((size_t)t * 256 + arg) * 4] - r0) / 100.0; }
                for (int gq = 0; gq < G; ++gq) if (hw[((size_t)t * 256 + gq) * 4 + 2] & 1) nprep++;
                lateR += (double)(hw[((size_t)t * 256 + arg) * 4] - r0) / 100.0; lateC += ch[arg];
                std::nth_element(ch.begin(), ch.begin() + G / 2, ch.end()); medC += ch[G / 2];
                hist[std::min(7, (int)((double)(pm - p0) / 100.0))]++;
            }
            fprintf(stderr, "[dc2015 spec per workgroup] the LAST publisher of a step: its R %.2f us after the earliest R, its R -> publish %.2f us (median workgroup %.2f); lateness histogram (us, 0..7+):", lateR / nn, lateC / nn, medC / nn);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %d", hist[k]);
            fprintf(stderr, "\n[dc2015 spec per workgroup] window of the last publisher: plain %d (R lag %.2f) | won branch prepared %d (R lag %.2f) | slow columns %d (R lag %.2f); workgroup-steps with a prepared branch per step %.2f",
                    kinds[0], kinds[0] ? klate[0] / kinds[0] : 0.0, kinds[1], kinds[1] ? klate[1] / kinds[1] : 0.0, kinds[2], kinds[2] ? klate[2] / kinds[2] : 0.0, (double)nprep / nn);
            fprintf(stderr, "\n[dc2015 spec per workgroup] last publisher is on average %.2f us behind the first; workgroups most often last:", late / nn);
            for (int k = 0; k < 6; ++k) { int arg = 0; for (int gq = 0; gq < G; ++gq) if (lastcnt[gq] > lastcnt[arg]) arg = gq; fprintf(stderr, " wg%d x%d (chain %.2f)", arg, lastcnt[arg], chain[arg] / nn); lastcnt[arg] = -1; }
            fprintf(stderr, "\n");
        }
        {   // arbitration segment by kind of step: no crossing | ring-resident (fast) | overflow granules / more blocks than the ring
            double s3[3] = {0, 0, 0}; int n3[3] = {0, 0, 0}; double rows = 0;
            for (int t = 2; t < R->T; ++t) {
                const long long *r = &h[(size_t)t * 24];
                const int kind = r[10] == 0 ? 0 : (r[11] >= 100 ? 2 : 1);
                s3[kind] += (double)(r[5] - r[4]) / 100.0; n3[kind]++; rows += (double)r[10];
            }
            fprintf(stderr, "[dc2015 spec arbitration] no crossing: %d steps %.2f us | ring-resident: %d steps %.2f us | heavy (overflow / > 7 blocks): %d steps %.2f us | crossing samples per step %.2f\n",
                    n3[0], n3[0] ? s3[0] / n3[0] : 0.0, n3[1], n3[1] ? s3[1] / n3[1] : 0.0, n3[2], n3[2] ? s3[2] / n3[2] : 0.0, rows / n);
        }
        fprintf(stderr, "[dc2015 spec, us from iteration start, workgroup %d] spec PostPre+X currents done %.2f | poll+decode done %.2f | rng run-ahead done %.2f | "
                        "behind barrier R %.2f | arbitration + finals %.2f | own-winner repair %.2f | membrane + publish %.2f || iteration %.2f us; "
                        "%d of %d iterations with a repair (avg %.2f us)\n", c.dbg_wg, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n, a[7] / n, step / n, nwin, n,
                nwin ? awin / nwin : 0.0);
    } else
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
    snn_set_plan_name(resident ? (lean ? "dc2015-resident-lean" : "dc2015-resident") : "dc2015-fused");
    *handled = 1;
    return SNN_OK;
}

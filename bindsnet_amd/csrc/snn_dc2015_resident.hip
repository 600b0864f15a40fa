// snn_dc2015_resident.hip -- plan "dc2015-resident": the DiehlAndCook2015 graph (bindsnet/models/models.py:156-244) for a
// whole network.run() (bindsnet/network/network.py:380-461) in ONE launch.  Decomposition and arithmetic are those of
// the one-launch-per-timestep kernel in snn_dc2015.hip (read its header comment first); this file holds what is specific
// to staying resident: weights / state on chip, the tagged-granule spike exchange, the templated kernel and its launcher.
#include <stdlib.h>
#include "snn_dc2015.hpp"
#include "snn_dc2015_tile.hpp"

namespace {

// =====================================================================================================
// Resident plan "dc2015-resident": the SAME decomposition and arithmetic as k_dc2015_step, but ONE launch for
// the whole run.  What used to cross the kernel boundary now stays put or moves through tagged granules:
//   * the [Nin x CW] slice of the learned weights and the [N x CW] slices of both recurrent matrices live in LDS for
//     all T steps; membrane state / traces / theta of a (sample, column) pair live in an LDS slot of its tile thread;
//   * the X trace of every step is precomputed (k_dc2015_xtrace) -- it depends on the inputs alone; the digest of the
//     next step streams into a second LDS buffer (global_load_lds) while the current one is in use;
//   * each workgroup keeps its own copy of the generator (all copies advance identically);
//   * the per-step spike exchange uses 8-byte {epoch, bits} granules written with ONE relaxed agent-scope
//     (write-through) store and polled with relaxed agent-scope loads: the data is the flag, no fence
//     (cdna_hip_programming.md Guideline 16, form R2).  Granule k of workgroup g carries 16/CW consecutive samples:
//     their CW crossing bits each in the low half, their CW Ai-spike bits each in the high half (two 16-bit slices of
//     the publishing wave's ballots).  Two buffers by epoch parity: a workgroup overwrites a buffer only after every
//     other workgroup has published the epoch in between, which it does only after consuming the overwritten one.
// Per step: receive (flags the samples with a crossing) | Ai event lists + scores of the crossings (one barrier closes
// both) | winners | Ae trace | PostPre on the LDS slice | currents (X part: four spare threads per pair, recurrent part:
// the tile thread) | membrane, publish -- while the spare threads write the step's raster rows.
// All G <= 256 workgroups are co-resident (one workgroup per CU); polls are bounded (device status word).
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
                                                         int wb, int wj, int BW, int tid, int nthreads, int RMK = 7) {
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
                    q = exp1_from_words(mt_temper(mt[((mb + m0) & RMK) * 624 + w0 - 624 * m0]),
                                        mt_temper(mt[((mb + m1) & RMK) * 624 + w1 - 624 * m1])); have = true;
                } else if (m0 >= lo && m0 <= hi) {             // pair straddles the resident range: park the high word
                    parked = mt_temper(mt[((mb + m0) & RMK) * 624 + w0 - 624 * m0]);
                } else if (m1 >= lo && m1 <= hi) {
                    q = exp1_from_words(parked, mt_temper(mt[((mb + m1) & RMK) * 624 + w1 - 624 * m1])); have = true;
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
            mt_twist_block_wave(mt + ((mb + hi) & RMK) * 624, mt + ((mb + hi + 1) & RMK) * 624, lane);
            if (hi + 2 <= ntw) mt_twist_block_wave(mt + ((mb + hi + 1) & RMK) * 624, mt + ((mb + hi + 2) & RMK) * 624, lane);
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
    return 4 * kBitWords * 4 + MAXB * cw * 4 + 8 * 624 * 4 + MAXB * 8 + 2 * MAXB * LR * 2 + 2 * MAXB * 4 + 2 * 32 * 4 + 32 +
           2 * MAXB * cw * 4 + 7 * MAXB * cw * 4 + MAXB * LR * 2 + MAXB * 4;
}     // bounded spin: ~0.5 s, then the run is flagged SNN_ERR_TIMEOUT

// CWR = columns per workgroup (8, 4 or 2): the PostPre stage is ALU-throughput bound inside a CU, so narrower tiles on
// more CUs shorten it, while the stages every workgroup repeats (receive, lists, arbitration) stay as they are.
// NTR = threads per workgroup: 1024, or 512 (twice the registers per thread: no spills; needs B*NW <= 512 and CW <= 4).
// LEAN = the common case compiled on its own (one_spike on, every input spike 0/1, no event-list overflow, Nin*N a
// multiple of 32): cold branches are pruned at compile time, the spike exchange is ONE compact summary granule per tile
// wave (up to three {type, sample, column} events inline; more -> the full bit granules of the general form, read on
// demand), and ONE wave receives, decodes AND scores the one_spike candidates while the others wait at the barrier, so
// the separate list-building and scoring stages (and their barrier) disappear.  A step it does not handle (multi-valued
// spike bytes, a sample with more than four inhibitory spikes, > 63 input events in a sample) is detected identically
// by every workgroup from the exchanged data: all of them leave the loop together with status SNN_ERR_RETRY, nothing
// having been written back, and the host repeats the input on the general form.
template <int CWR, int NTR, bool LEAN>
__global__ __launch_bounds__(NTR) void k_dc2015_run(const DcCtx c) {
    constexpr int CW = CWR, TT = MAXB * CWR, NT = NTR;     // (shadow the per-step kernel's constants)
    constexpr int NTW = TT / 64, SPW = 64 / CW;            // tile waves; samples per tile wave
    constexpr int SPG = 16 / CW;                           // exchange: samples per granule (CW crossing bits + CW Ai-spike bits each)
    constexpr int WPB = 8 / CW;                            //           workgroups sharing one byte of a sample's bit string
    constexpr uint32_t FM = (1u << CW) - 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (c.gate) {
        // (a pipelined caller's second attempt: whether it is needed or not, the exchange areas of the caller's NEXT run are cleared here)
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (unsigned k = blockIdx.x * NTR + threadIdx.x; k < c.zeroA_n16; k += gridDim.x * NTR) c.zeroA[k] = z;
        for (unsigned k = blockIdx.x * NTR + threadIdx.x; k < c.zeroG_n16; k += gridDim.x * NTR) c.zeroG[k] = z;
        if (__hip_atomic_load(c.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != SNN_ERR_RETRY) return;          // a second attempt nobody needs
    }
    const int B = c.B, Nin = c.Nin, N = c.N, NW = c.NW, NinW = c.NinW, T = c.T;
    // ---- LDS carve-up.  Everything of fixed size sits at a compile-time offset (addresses fold into the
    //      instructions' immediate offsets instead of occupying registers); the four size-dependent arrays follow.
    constexpr size_t O_CRS = 0, O_FINE = O_CRS + kBitWords * 4, O_SPI = O_FINE + kBitWords * 4, O_XNU0 = O_SPI + 2 * kBitWords * 4,
                     O_MT = O_XNU0 + MAXB * CW * 4, O_KEYS = O_MT + 8 * 624 * 4,
                     O_LSTI = O_KEYS + MAXB * 8, O_LSTE = O_LSTI + MAXB * LR * 2, O_CNTI = O_LSTE + MAXB * LR * 2,
                     O_CNTE = O_CNTI + MAXB * 4, O_CNT = O_CNTE + MAXB * 4, O_COLM = O_CNT + 32 * 4, O_MISC = O_COLM + 32 * 4,
                     O_LSTIB = O_MISC + 32, O_CNTIB = O_LSTIB + MAXB * LR * 2,
                     O_CURB = O_CNTIB + MAXB * 4, O_ST = O_CURB + 2 * MAXB * CW * 4, O_WT = O_ST + 7 * MAXB * CW * 4;
    static_assert(O_WT == resident_fixed_lds(CW) && O_WT % 16 == 0, "fixed LDS part");
    static_assert(O_XNU0 % 16 == 0, "x_tgt*nu0 is read as float4");
    uint32_t *crs = (uint32_t *)(smem + O_CRS);            // [B][NW] Ae crossings of step t-1 (B * NW <= NT)
    uint32_t *finE = (uint32_t *)(smem + O_FINE);          // ... final Ae spikes
    uint32_t *spI2 = (uint32_t *)(smem + O_SPI);           // ... Ai spikes, two buffers by step parity (the raster rows of
                                                           //     one step are written while the next receive may already run)
    float *xnu0 = (float *)(smem + O_XNU0);
    uint32_t *mt = (uint32_t *)(smem + O_MT);              // generator ring: block base+m in slot (mb + m) & 7
    unsigned long long *keys = (unsigned long long *)(smem + O_KEYS);
    uint16_t *lstI0 = (uint16_t *)(smem + O_LSTI), *lstI1 = (uint16_t *)(smem + O_LSTIB);   // (second copy: lean form only)
    uint16_t *lstE = (uint16_t *)(smem + O_LSTE);
    int *cntI0 = (int *)(smem + O_CNTI), *cntI1 = (int *)(smem + O_CNTIB);
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
    if (g == c.stall_wg) return;                           // test hook: a workgroup that never takes part
    const int jj = tid % CW, bl = tid / CW;
    const int j = c0 + jj;
    const bool colv = j < N;
    const bool tailcol = c0 >= (N / 32) * 32;
    const int BW = B * NW;
    const bool mine = tid < TT && bl < B && colv;
    const unsigned kst = (unsigned)(bl * N + j);
    const int wb = (int)(((float)tid + 0.5f) * c.inv_NW), wj = tid - wb * NW;   // exchange word (sample wb, word wj)
    const int KB = c.KB, NG = c.G * KB;                    // granules per epoch
    const int NGS = c.G * NTW;                             // lean form: summary granules per epoch
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
    if (tid < MAXB) { keys[tid] = 0ull; cntI0[tid] = 0; cntI1[tid] = 0; }
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
        uint16_t *lstI = (LEAN && (t & 1)) ? lstI1 : lstI0;               // Ai event lists of step t-1 (lean: by step parity)
        int *cntI = (LEAN && (t & 1)) ? cntI1 : cntI0;
        // ------------------------------------------------------------------ receive step t-1
        const bool use_rng = phaseA && (LEAN || c.pE.one_spike);
        uint32_t anym = 0;
        bool heavy = !LEAN;                                               // lean: this step needs the all-thread scoring stage
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
        uint32_t pay = 0;                                                  // lean: this thread's summary granule of epoch t
        if (LEAN && phaseA) {
            if (tid < NGS) {
                // ---- one thread per summary granule (G * NTW of them): poll it, decode its events into the bit words /
                //      Ai event lists / crossing-sample mask in LDS; the candidates are scored behind the barrier
                const unsigned long long *sums = c.exs + (size_t)(t & 1) * NGS;
                const unsigned long long *exr = c.ex + (size_t)(t & 1) * NG;
                unsigned long long x;
                for (unsigned spins = 0;; ++spins) {
                    x = granule_load(sums + tid);
                    if ((uint32_t)(x >> 32) == (uint32_t)t || failed) break;
                    if (spins > kPollLimit) { failed = true; if (c.status) __hip_atomic_store(c.status, (int)SNN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                pay = (uint32_t)x;
                if (pay) {
                    uint32_t my_any = 0;
                    auto event = [&](int bsm, int jx, bool inh) {       // one crossing / inhibitory spike of step t-1
                        if (bsm >= B || jx >= N) return;
                        if (!inh) { atomicOr((unsigned int *)&crs[bsm * NW + (jx >> 5)], 1u << (jx & 31)); my_any |= 1u << bsm; }
                        else {
                            atomicOr((unsigned int *)&spI[bsm * NW + (jx >> 5)], 1u << (jx & 31));
                            const int slot = atomicAdd(&cntI[bsm], 1);
                            if (slot < LR) lstI[bsm * LR + slot] = (uint16_t)jx;
                            if (slot >= 4) atomicOr((unsigned int *)&misc[2], 2u);   // more than the four-entry fast path takes
                        }
                    };
                    const int gsrc = tid / NTW, w = tid - gsrc * NTW;
                    if ((pay & 0xFFu) == 0xFFu) {                        // overflow: that wave's full bit granules
                        misc[6] = 1;
                        for (int q = 0; q < SPW / SPG; ++q) {
                            const int k = w * (SPW / SPG) + q;
                            if (k >= KB) break;
                            unsigned long long d;
                            for (unsigned sp2 = 0;; ++sp2) {
                                d = granule_load(exr + gsrc * KB + k);
                                if ((uint32_t)(d >> 32) == (uint32_t)t || failed) break;
                                if (sp2 > kPollLimit) { failed = true; if (c.status) __hip_atomic_store(c.status, (int)SNN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            uint32_t be = (uint32_t)d & 0xFFFFu, bi = ((uint32_t)d >> 16) & 0xFFFFu;
                            while (be) { const int p_ = __ffs(be) - 1; be &= be - 1; event(k * SPG + p_ / CW, gsrc * CW + p_ % CW, false); }
                            while (bi) { const int p_ = __ffs(bi) - 1; bi &= bi - 1; event(k * SPG + p_ / CW, gsrc * CW + p_ % CW, true); }
                        }
                    } else {
                        const int ne = (int)(pay >> 30);
                        for (int e = 0; e < ne; ++e) {
                            const uint32_t ev = (pay >> (8 * e)) & 0xFFu;
                            const int p_ = (int)(ev & 0x3Fu);
                            event(w * SPW + p_ / CW, gsrc * CW + p_ % CW, (ev & 0x40u) != 0);
                        }
                    }
                    if (my_any) atomicOr((unsigned int *)&misc[3], my_any);
                }
            }
        } else if (phaseA) {
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
        const uint8_t *sbytes = LEAN ? nullptr : ((mflags & 1) ? sprev_g : nullptr);   // lean: such a step ends the launch (below)
        if (tid == 0 && (mflags & (LEAN ? 4 : 2))) atomicOr((unsigned int *)&misc[2], 2u);   // lean: lists are walked by group size, any length up to LX - 1
        if (LEAN) {
            if (phaseA) { anym = (uint32_t)__builtin_amdgcn_readfirstlane(misc[3]); heavy = __builtin_amdgcn_readfirstlane(misc[6]) != 0; }
            else heavy = true;                                            // t == 0: lists from the layers' spike bytes
            if (phaseB) {   // what the NEXT decode accumulates into: its previous readers ended before the barrier above
                for (int k = tid; k < BW; k += NT) spI2[((t + 1) & 1) * kBitWords + k] = 0;
                if (tid < MAXB) ((t & 1) ? cntI0 : cntI1)[tid] = 0;
            }
        }
        const bool do_stdp = phaseA && c.learning && c.rule == SNN_RULE_POSTPRE;
        const bool stdp_full = t == 1;
        const int nact = stdp_full ? Nin : __builtin_amdgcn_readfirstlane(meta[32]);
        // ---- per sample (one wave each, in turns): event list of its Ai spikes; does it have an Ae crossing?
        //      (lean form: the receiving wave has built them already, except at t == 0)
        if (LEAN && phaseA) {
        } else if (NW <= 32) {                                    // two samples per wave, one per half
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
        int arb_rows = 0, arb_E = 0, arb_ntw = 0;
        if (use_rng) {
            if (!LEAN) anym = (uint32_t)__builtin_amdgcn_readfirstlane(misc[3]);
            arb_rows = __popc(anym);
            arb_E = rng_pos + 2 * arb_rows * N;
            arb_ntw = arb_rows ? (arb_E - 1) / 624 : 0;
            if (LEAN && arb_ntw > 7) heavy = true;         // more generator blocks than the ring holds: the block-by-block walk
            if (LEAN && !heavy && arb_rows) {
                // ---- one wave per sample with a crossing (rank r = its position among them), lane = bit word of the sample.
                // The winner is argmax over the candidates j of fl32(1 / q_j), q_j = fl32(-log1p(-u_j)), u_j = m_j * 2^-53 the
                // j-th draw, ties to the lowest j (nodes.py:1097-1105).  u -> fl32(1/q) is monotone non-increasing, and for
                // m2 > m1 + (m1 >> 19) strictly decreasing: -log1p(-u) grows by at least the relative step of u (its
                // derivative 1/(1-u) >= f(u)/u), i.e. by more than 2^-19, which the three roundings on the way (f64 log1p,
                // f32 cast, f32 division: < 2^-22 together) cannot close.  So the candidate with the smallest 53-bit draw wins
                // outright unless another one lies inside that margin -- only then (probability ~ candidates * 2^-19) are
                // the logarithms evaluated.  No log1p on the common path.
                int r = 0;
                for (uint32_t rem = anym; rem; rem &= rem - 1, ++r) {
                    if ((r % NWV) != wave) continue;
                    const int bsm = __ffs(rem) - 1;
                    const uint32_t bits = lane < NW ? crs[bsm * NW + lane] : 0u;
                    unsigned long long k1 = ~0ull, k2 = ~0ull;            // this lane's two smallest keys (m << 10 | j)
                    for (uint32_t bb = bits; bb; bb &= bb - 1) {
                        const int jx = lane * 32 + __ffs(bb) - 1;
                        const int w0 = rng_pos + 2 * (r * N + jx), w1 = w0 + 1;
                        const int m0 = w0 / 624, m1 = w1 / 624;
                        const uint32_t hi = mt_temper(mt[((mb + m0) & 7) * 624 + w0 - 624 * m0]);
                        const uint32_t lo = mt_temper(mt[((mb + m1) & 7) * 624 + w1 - 624 * m1]);
                        const unsigned long long m = (((unsigned long long)hi << 32) | lo) & ((1ull << 53) - 1ull);
                        const unsigned long long key = (m << 10) | (unsigned long long)jx;
                        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                    }
                    if (lane == 0) keys[bsm] = ~0ull;                      // (LDS operations of one wave execute in order)
                    if (bits) atomicMin(&keys[bsm], k1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long kmin = *(volatile unsigned long long *)&keys[bsm];
                    const unsigned long long mmin = kmin >> 10, zone = mmin + (mmin >> c.zone_shift) + 1ull;   // zone_shift = 19 (test hook: smaller)
                    const bool close = (k1 != ~0ull && k1 != kmin && (k1 >> 10) <= zone) || (k2 != ~0ull && (k2 >> 10) <= zone);
                    int win = (int)(kmin & 1023ull);
                    if (__any(close)) {                                    // rare: exact evaluation of every candidate
                        if (lane == 0) keys[bsm] = 0ull;
                        for (uint32_t bb = bits; bb; bb &= bb - 1) {
                            const int jx = lane * 32 + __ffs(bb) - 1;
                            const int w0 = rng_pos + 2 * (r * N + jx), w1 = w0 + 1;
                            const int m0 = w0 / 624, m1 = w1 / 624;
                            const float q = exp1_from_words(mt_temper(mt[((mb + m0) & 7) * 624 + w0 - 624 * m0]),
                                                            mt_temper(mt[((mb + m1) & 7) * 624 + w1 - 624 * m1]));
                            const float val = 1.0f / q;
                            atomicMax(&keys[bsm], ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx));
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        win = (int)(0xFFFFFFFFu - (uint32_t)(*(volatile unsigned long long *)&keys[bsm] & 0xFFFFFFFFull));
                    }
                    if (lane == 0) keys[bsm] = (unsigned long long)(0xFFFFFFFFu - (uint32_t)win);   // (what the stages below decode)
                }
            }
            if (heavy && arb_ntw <= 7 && tid < BW) {            // every block the step consumes is resident (ring run ahead at the top)
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
        if (heavy || arb_rows > 0) lds_barrier();          // (lean, step without a crossing: nothing happened since the receive barrier)
        if (tid == 35) { misc[3] = 0; misc[6] = 0; }       // crossing-sample mask / overflow mark: every thread has read them; next set by the next receive
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
            if (LEAN) for (int k = tid; k < BW; k += NT) crs[k] = 0;     // (every reader of this step's crossings is done)
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
                        if constexpr (LEAN && CW == 4) {
                            if (c.rows4) stdp_rows4<NT>(c, nact, arows, rowmask, colmask, xnu0, xsrc, wtile, c0, tid);
                            else stdp_rows_lds<CascadeT, false, CW, NT>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                        } else
                        stdp_rows_lds<CascadeT, false, CW, NT>(c, nact, arows, rowmask, colmask, sbytes, xnu0, xsrc, wtile, c0, tid, Emain);
                        stdp_cols_lds<CascadeT, CW, NT>(c, acols, rowmask, colmask, xsrc, wtile, c0, tid, Emain);
                    }
                }
            }
        }
        const bool busy = (__builtin_amdgcn_readfirstlane(misc[2]) & 2) != 0;
        lds_barrier();
        DBG_MARK(4);
        if (LEAN && (busy || (mflags & 1))) {
            // a step the lean form does not handle.  Every workgroup derives this from the same exchanged data / input
            // digest, so all of them leave here in the same iteration and nobody is left waiting for a granule.
            if (tid == 0 && c.status) __hip_atomic_store(c.status, (int)SNN_ERR_RETRY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            failed = true;
            break;
        }
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
                    constexpr uint32_t GM = (1u << GCB) - 1u;
                    const int st = (cL > 0 ? (int)(gc & GM) : 0) + (cL > 1 ? (int)((gc >> GCB) & GM) : 0) + (cL > 2 ? (int)((gc >> (2 * GCB)) & GM) : 0);
                    const int nL = (int)((gc >> (GCB * cL)) & GM);
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
                        const int s4 = (int)(gc & GM) + (int)((gc >> GCB) & GM) + (int)((gc >> (2 * GCB)) & GM) + (int)((gc >> (3 * GCB)) & GM);
                        const int n5 = (int)((gc >> (4 * GCB)) & GM);
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
                constexpr uint32_t GM = (1u << GCB) - 1u;
                const int st = (pL > 0 ? (int)(gq & GM) : 0) + (pL > 1 ? (int)((gq >> GCB) & GM) : 0) + (pL > 2 ? (int)((gq >> (2 * GCB)) & GM) : 0);
                const int nL = (int)((gq >> (GCB * pL)) & GM);
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
        if constexpr (LEAN) {
            // publish step t as epoch t+1: ONE summary granule per tile wave -- {count, up to three events} with event =
            // type << 6 | lane (lane = sample-in-wave * CW + column); more than three: the wave's full bit granules first
            // (general layout), drained, then the summary with the overflow mark
            if (wave < NTW) {
                const uint64_t mE = __ballot(spE), mI = __ballot(spIn);
                const int nev = __popcll(mE) + __popcll(mI);
                uint32_t pay;
                if (nev <= 3) {
                    pay = (uint32_t)nev << 30;
                    int sh = 0;
                    for (uint64_t m = mE; m; m &= m - 1) { pay |= (uint32_t)(__ffsll((unsigned long long)m) - 1) << sh; sh += 8; }
                    for (uint64_t m = mI; m; m &= m - 1) { pay |= (uint32_t)(0x40 | (__ffsll((unsigned long long)m) - 1)) << sh; sh += 8; }
                } else {
                    const int sidx = lane / CW, b = wave * SPW + sidx;
                    const uint32_t v = (uint32_t)((mE >> (sidx * CW)) & 0xFFFFull) | ((uint32_t)((mI >> (sidx * CW)) & 0xFFFFull) << 16);
                    if ((lane % CW) == 0 && (sidx % SPG) == 0 && b < B)
                        granule_store(c.ex + (size_t)((t + 1) & 1) * NG + g * KB + b / SPG, ((unsigned long long)(uint32_t)(t + 1) << 32) | v);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pay = 0xC0FFFFFFu;
                }
                if (lane == 0) granule_store(c.exs + (size_t)((t + 1) & 1) * NGS + g * NTW + wave, ((unsigned long long)(uint32_t)(t + 1) << 32) | pay);
            }
        } else {   // publish crossing / spike bits of step t: epoch t+1.  A wave holds 64/CW samples x CW columns; SPG
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

    // ---- epilogue: state and weights back to the tensors the caller owns.  Nothing the caller owns as STATE (membrane
    //      state, traces, theta, weights, generator) has been written so far, so a run that gave up on a hand-off --
    //      here or in any other workgroup (device status word) -- returns with all of it untouched and the host re-runs
    //      the input on the one-launch-per-timestep plan (Network.run); only the monitor rasters hold garbage by then.
    if (tid == 0) misc[4] = c.status ? __hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (failed) misc[5] = 1;
    __syncthreads();
    if (misc[4] != 0 || misc[5] != 0) return;
    if (c.x_traces) {                                  // X trace after the last step (k_dc2015_xtrace's entry T)
        const float *src = c.xtr + (size_t)T * B * Nin;
        for (int k = g * NT + tid; k < B * Nin; k += c.G * NT) c.xX[1][k] = src[k];
    }
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

// =====================================================================================================================
// Lean form, second generation ("speculate both ways"): the same arithmetic and the same compact exchange as
// k_dc2015_run<4, 1024, true>, re-ordered so that nearly everything of a timestep runs while the spike exchange is in
// flight, and what remains behind it is a lookup.
//
// What of "finish step t-1, start step t" depends on the other workgroups' spikes of step t-1?  Only (1) which of the own
// crossings WON the one_spike arbitration (-> Ae trace, PostPre post-synaptic term of that column, its Ae -> Ai current)
// and (2) the inhibitory currents.  A (sample, column) pair without a winner -- all but ~3 of the 12 800 pairs of a step
// at cfg2 -- has a trace that just decays, a PostPre that is its pre-synaptic term, and X -> Ae currents that follow from
// those; and a workgroup KNOWS which of its pairs crossed, so it can prepare the "it won" outcome of a crossing column
// as well.  Weights are double-buffered for this: `wtile` holds the committed slice and is never written inside the
// window, `wnew` the speculative new values of the rows PostPre touches, `wwin` the whole column(s) of the won branch.
// Per iteration t:
//
//   window (no workgroup barrier; the exchange of epoch t is in flight):
//     waves 2..9    PostPre of step t-1 for every active row under the assumption "no own winner" (wtile -> wnew),
//                   LDS-counter barrier among these 8 waves, X -> Ae currents of step t from wnew
//     waves 0..1    for every own column with exactly ONE crossing sample: the column as it is if that pair wins
//                   (new trace in the pre-synaptic term + post-synaptic term, all rows: wtile -> wwin), barrier among
//                   the two, that column's X -> Ae currents from wwin
//     waves 10..13  poll + decode the summary granules of epoch t (crossings / Ai spikes of step t-1)
//     wave 14       spike-raster rows of step t-2, digest of step t+1 -> LDS, scratch resets
//     wave 15       generator run-ahead (16-block ring)
//   barrier R
//   fast iterations (no overflow granule, generator blocks resident, no own column with two crossing samples, Ae -> Ai
//   weights diagonal in the own slice):
//     waves 0..1    a crossing column won iff its sample has no other crossing (one candidate: no draw needed), else iff
//                   the arbitration says so (they wait for that sample's result only); select the X currents of the
//                   branch that happened, recurrent currents, membrane update, publish epoch t+1, the next window's
//                   x_tgt*nu0 and crossing bookkeeping
//     waves 2..15   arbitration (identical in every workgroup: generator position, rasters), LDS-counter barrier among
//                   the 14, final spike words, commit wnew (and wwin of the winners) -> wtile
//   slow iterations (anything else, and t = 0): first generation's order -- arbitration by everybody, barrier, winners'
//     trace, PostPre of the winning column(s) redone for all rows, their X currents redone, then the tile threads.
//   barrier E
// Bit-exactness: every value is produced by the same operations in the same order as in the first-generation kernel; a
// column of the won branch is computed by the (row, column) form of the update (stdp_rows_lds) from the committed weights.
constexpr int kSpecEarlyTwists = 2; // generator blocks refilled behind barrier R (the rest in the next window)
constexpr int kSpecRing = 16;      // generator blocks resident (a step with up to 11 crossing samples stays on the fast path)
constexpr size_t spec_fixed_lds() { return resident_fixed_lds(4) + (size_t)(kSpecRing - 8) * 624 * 4; }
// behind the digests: wnew [Nin][4], wwin [Nin][4], second crossing-bit buffer, curXwin [32][4], xwin [2][4], arbdone [32]
constexpr size_t spec_tail_lds(int Nin) { return (size_t)2 * Nin * 4 * 4 + kBitWords * 4 + MAXB * 4 * 4 + 8 * 4 + MAXB * 4; }
template <int NTR>
__global__ __launch_bounds__(NTR) void k_dc2015_spec(const DcCtx c) {
    constexpr int CW = 4, TT = MAXB * CW, NT = NTR;
    constexpr int NTW = TT / 64, SPW = 64 / CW, SPG = 16 / CW;
    constexpr int NWV = NT / 64;
    constexpr int W_Q0 = 2, W_QN = 8;                     // waves doing the speculative PostPre + X currents (QT threads)
    constexpr int QT = W_QN * 64;
    constexpr int W_P0 = 10, W_PN = 4;                    // polling / decoding waves
    constexpr int W_AUX = 14, W_RNG = 15;
    constexpr int NOT = NT - TT, NOW_ = NWV - NTW;        // threads / waves that are not tile waves
    constexpr int RB = kSpecRing, RMK = RB - 1;           // generator ring: RB blocks of 624 words
    static_assert(NT == 1024 && QT >= MAXB * CW * 4 && NTW == 2, "wave roles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = c.B, Nin = c.Nin, N = c.N, NW = c.NW, T = c.T;
    constexpr size_t O_CRS = 0, O_FINE = O_CRS + kBitWords * 4, O_SPI = O_FINE + kBitWords * 4, O_XNU0 = O_SPI + 2 * kBitWords * 4,
                     O_MT = O_XNU0 + MAXB * CW * 4, O_KEYS = O_MT + RB * 624 * 4,
                     O_LSTI = O_KEYS + MAXB * 8, O_LSTE = O_LSTI + MAXB * LR * 2, O_CNTI = O_LSTE + MAXB * LR * 2,
                     O_CNTE = O_CNTI + MAXB * 4, O_CNT = O_CNTE + MAXB * 4, O_COLM = O_CNT + 32 * 4, O_MISC = O_COLM + 32 * 4,
                     O_LSTIB = O_MISC + 32, O_CNTIB = O_LSTIB + MAXB * LR * 2,
                     O_CURB = O_CNTIB + MAXB * 4, O_ST = O_CURB + 2 * MAXB * CW * 4, O_WT = O_ST + 7 * MAXB * CW * 4;
    static_assert(O_WT == spec_fixed_lds() && O_WT % 16 == 0, "fixed LDS part");
    uint32_t *crsA = (uint32_t *)(smem + O_CRS);           // Ae crossings of step t-1, two buffers by iteration parity (crsA: even)
    uint32_t *finE = (uint32_t *)(smem + O_FINE);
    uint32_t *spI2 = (uint32_t *)(smem + O_SPI);
    float *xnu0 = (float *)(smem + O_XNU0);
    uint32_t *mt = (uint32_t *)(smem + O_MT);
    unsigned long long *keys = (unsigned long long *)(smem + O_KEYS);
    uint16_t *lstI0 = (uint16_t *)(smem + O_LSTI), *lstI1 = (uint16_t *)(smem + O_LSTIB);
    uint16_t *lstE = (uint16_t *)(smem + O_LSTE);
    int *cntI0 = (int *)(smem + O_CNTI), *cntI1 = (int *)(smem + O_CNTIB);
    // cnt[0..7]: crossing counts of the own columns by step parity; [16 + p]: samples with a crossing, [18 + p]: overflow mark
    // (p = iteration parity: the decode of iteration t fills what iteration t-1's aux wave cleared); counters of the partial
    // barriers: [20] the 8 speculative waves, [21] the 14 non-tile waves, [22] tile + speculative waves; [24 + p]: row chunks of the
    // won branch handed out so far
    int *cnt = (int *)(smem + O_CNT);
    // colmask[0..3]: samples with a final spike per own column (slow iterations); [8 + 4 p + q]: crossing samples of own
    // column q at the step whose finish is iteration parity p; [16]: the own Ae -> Ai slice has an off-diagonal weight
    uint32_t *colmask = (uint32_t *)(smem + O_COLM);
    int *misc = (int *)(smem + O_MISC);                    // [2] give-up flag (monotonic: once set the launch ends)
    float *curX = (float *)(smem + O_CURB);                // [B][CW] X -> Ae part of the Ae current of step t ("nobody won" branch)
    float *stl = (float *)(smem + O_ST);
    float *wtile = (float *)(smem + O_WT);                 // [Nin][CW] committed weights: never written inside the window
    float *wieT = wtile + (size_t)Nin * CW;
    float *weiT = wieT + (size_t)N * CW;
    const int DGS = (c.DGW + 63) & ~63;
    uint32_t *dgbuf = (uint32_t *)(weiT + (size_t)N * CW);
    float *wnew = (float *)(dgbuf + 2 * DGS);              // [Nin][CW] rows PostPre touches: their weights if no own pair wins
    float *wwin = wnew + (size_t)Nin * CW;                 // [Nin][CW] column q: the whole column if its crossing pair wins
    uint32_t *crsB = (uint32_t *)(wwin + (size_t)Nin * CW);
    float *curXwin = (float *)(crsB + kBitWords);          // [B][CW] X -> Ae currents of column q in its won branch
    float *xwinv = curXwin + MAXB * CW;                    // [2][CW] x_tgt*nu0 of the crossing pair of column q if it wins
    int *arbdone = (int *)(xwinv + 8);                     // [B] iteration in which the sample's arbitration result was stored
    // (buffers that are picked at run time are addressed as offsets from ONE base pointer each: a select between two
    //  pointers makes the compiler lose the LDS address space and fall back to flat loads)
    const int offN = (int)(wnew - wtile), offW = (int)(wwin - wtile), offCW = (int)(curXwin - curX), offCB = (int)(crsB - crsA);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (developer switch: the grid may be launched `mult` times too large, only every mult-th workgroup taking part -- spreads
    //  the working workgroups over the compute units differently)
    const int gmult = (c.spec_flags >> 8) > 1 ? (c.spec_flags >> 8) : 1;
    if ((int)blockIdx.x % gmult != 0) return;
    const int g = (int)blockIdx.x / gmult, c0 = g * CW;
    if (g == c.stall_wg) return;
    const int jj = tid % CW, bl = tid / CW;
    const int j = c0 + jj;
    const bool colv = j < N;
    const bool tailcol = c0 >= (N / 32) * 32;
    const int BW = B * NW;
    const bool mine = tid < TT && bl < B && colv;
    const unsigned kst = (unsigned)(bl * N + j);
    const int wb = (int)(((float)tid + 0.5f) * c.inv_NW), wj = tid - wb * NW;
    const int KB = c.KB, NG = c.G * KB;
    const int NGS = c.G * NTW;

    if (tid < 32) { cnt[tid] = 0; colmask[tid] = 0; }
    if (tid < 8) misc[tid] = 0;
    __syncthreads();
    for (int k = tid; k < Nin * CW; k += NT) {
        const int i = k / CW, q = k % CW;
        wtile[k] = (c0 + q < N) ? c.Wxe[i * N + c0 + q] : 0.f;
    }
    {
        bool offdiag = false;
        for (int k = tid; k < N * CW; k += NT) {
            const int i = k / CW, q = k % CW;
            wieT[k] = (c0 + q < N) ? c.Wie[i * N + c0 + q] : 0.f;
            const float we = (c0 + q < N) ? c.Wei[i * N + c0 + q] : 0.f;
            weiT[k] = we;
            offdiag = offdiag || (i != c0 + q && we != 0.f);
        }
        if (offdiag) colmask[16] = 1u;
    }
    auto fetch_digest = [&](int e, int first_wave, int nwaves) __attribute__((always_inline)) {        // by `nwaves` whole waves starting at `first_wave`
        const uint32_t *Dg = c.dig + (size_t)e * c.DW;
        uint32_t *dst = dgbuf + (e & 1) * DGS;
        for (int base = (wave - first_wave) * 256; base < c.DGW; base += nwaves * 256) {
            const int ub = __builtin_amdgcn_readfirstlane(base);
            if (ub + lane * 4 < c.DGW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Dg + ub + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(dst + ub), 16, 0, 0);
        }
    };
    fetch_digest(0, 0, NWV);
    bool last_sE = false, last_sI = false;
    if (mine) {
        stl[0 * TT + tid] = c.vE[kst]; stl[1 * TT + tid] = c.rE[kst]; stl[2 * TT + tid] = c.vI[kst]; stl[3 * TT + tid] = c.rI[kst];
        stl[4 * TT + tid] = c.pE.lif.traces ? c.xE[kst] : 0.f;
        stl[5 * TT + tid] = c.pI.traces ? c.xI[kst] : 0.f;
        stl[6 * TT + tid] = c.theta[j];
        last_sE = c.sE[kst] != 0; last_sI = c.sI[kst] != 0;
    }
    int rng_pos = 0, mb = 0, ahead = 0; long long rng_consumed = 0;
    {
        for (int k = tid; k < 624; k += NT) mt[k] = c.rng[0]->mt[k];
        rng_pos = __builtin_amdgcn_readfirstlane(c.rng[0]->pos);
        const long long cons0 = c.rng[0]->consumed;
        rng_consumed = ((long long)__builtin_amdgcn_readfirstlane((int)(cons0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)cons0);
    }
    if (tid < MAXB) { keys[tid] = 0ull; cntI0[tid] = 0; cntI1[tid] = 0; arbdone[tid] = -1; }
    if (tid < TT) { xnu0[tid] = 0.f; curX[tid] = 0.f; curXwin[tid] = 0.f; }
    if (tid < 8) xwinv[tid] = 0.f;
    if (tid < BW) { crsA[tid] = 0; crsB[tid] = 0; finE[tid] = 0; spI2[tid] = 0; spI2[kBitWords + tid] = 0; }
    bool failed = false;
    int sub_q = 0, sub_o = 0, sub_t = 0;                    // targets of the partial barriers (cnt[20], cnt[21], cnt[22])
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool diagE = colmask[16] == 0u;

    auto part_barrier = [&](int *ctr, int &target, int nwaves) __attribute__((always_inline)) {         // barrier among `nwaves` whole waves (LDS counter)
        target += nwaves;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) { if (c.spec_flags & 32) __builtin_amdgcn_s_sleep(1); }
        asm volatile("" ::: "memory");
    };
    auto win_of = [&](int b) __attribute__((always_inline)) -> int { return (int)(0xFFFFFFFFu - (uint32_t)(keys[b] & 0xFFFFFFFFull)); };
    // raster rows (final Ae spikes / Ai spikes) of one step, whole [N]-byte rows, row r of the 2*B rows by workgroup r mod G
    auto raster_rows = [&](int step, int spi_off, int first_thread, int nthreads) __attribute__((always_inline)) {   // spi_off: the Ai words, as an offset from finE
        for (int r = g; r < 2 * B; r += c.G) {
            const int b = r < B ? r : r - B;
            uint8_t *ras = r < B ? c.rasE : c.rasI;
            const uint32_t *bitsrc = finE + (r < B ? 0 : spi_off) + b * NW;
            if (ras) { uint8_t *row = ras + ((size_t)step * B + b) * N; for (int jx = tid - first_thread; jx < N; jx += nthreads) row[jx] = (uint8_t)bit_of(bitsrc, jx); }
        }
    };
    // one element of PostPre in its (row, column) form (= stdp_rows_lds's update; lean form: no tail elements, 0/1 spikes):
    // row i, column q, pre-synaptic samples m, post-synaptic samples cm; sample bst's x_tgt*nu0 replaced by xw when bst >= 0
    auto postpre_elem = [&](float w, int i, int q, uint32_t m, uint32_t cm, int bst, float xw, const float *xs, bool have_x = false, float xval = 0.f) __attribute__((always_inline)) -> float {
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
                    acc.add(b, (have_x ? xval : xs[b * Nin + i]) * (1.0f * c.nu1), B);   // (have_x: one post-synaptic sample, its X trace passed in)
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

    if (c.dbg && tid == 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)0 * 256 + g) * 4 + 3] = (long long)((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 16) | (long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFF);   // XCC_ID, HW_ID
    for (int t = 0; t <= T; ++t) {
        const bool phaseA = t >= 1, phaseB = t < T;
        const int par = t & 1;
        if (c.dbg && g == c.dbg_wg && threadIdx.x == 0) { c.dbg[(size_t)t * 24 + 0] = (long long)wall_clock64(); c.dbg[(size_t)t * 24 + 8] = (long long)clock64(); }
        if (c.dbg && threadIdx.x == 0) atomicMin((unsigned long long *)&c.dbg[(size_t)t * 24 + 20], (unsigned long long)wall_clock64());
        uint32_t *spI = spI2 + par * kBitWords;
        uint32_t *crs = crsA + (par ? offCB : 0);
        const uint32_t *dg = dgbuf + par * DGS;                           // digest of the X spikes of step t-1
        const uint16_t *lstX = (const uint16_t *)dg;
        const int *meta = (const int *)(dg + B * (LX / 2));
        const uint32_t *rowmask = dg + B * (LX / 2) + 40;
        const uint16_t *arows = (const uint16_t *)(rowmask + Nin);
        const uint16_t *lst2 = arows + 4 * ((Nin + 1) / 2);
        const uint32_t *gcnt = (const uint32_t *)(lst2 + B * LX);
        const uint32_t *gqn = gcnt + B;
        uint16_t *lstI = lstI0 + (par ? (int)(lstI1 - lstI0) : 0);
        int *cntI = cntI0 + (par ? (int)(cntI1 - cntI0) : 0);
        const int mflags = __builtin_amdgcn_readfirstlane(meta[33]);
        const bool do_stdp = phaseA && c.learning && c.rule == SNN_RULE_POSTPRE;
        const bool stdp_full = t == 1;
        const int nact = stdp_full ? Nin : __builtin_amdgcn_readfirstlane(meta[32]);
        const float *xsrc = c.xtr + (size_t)t * B * Nin;                  // X trace after step t-1
        const bool use_rng = phaseA;
        // crossing samples of the own columns at step t-1 (set by the tile threads at the end of the previous iteration)
        uint32_t xm[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) xm[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)colmask[8 + 4 * par + q]);
        const bool slowcols = __popc(xm[0]) > 1 || __popc(xm[1]) > 1 || __popc(xm[2]) > 1 || __popc(xm[3]) > 1 ||
                              ((c.spec_flags & 1) && (xm[0] | xm[1] | xm[2] | xm[3]) != 0u) || (c.spec_flags & 2);
        const bool prep = do_stdp && diagE && !slowcols && (xm[0] | xm[1] | xm[2] | xm[3]) != 0u;   // own columns with a prepared won branch

        // X -> Ae part of the Ae current of step t of (sample pb, column pq), lane pL of its four threads (from the X spikes
        // of step t-1 and the weights in `ws`): written by lane 0 to dst[pb][pq]
        auto x_current = [&](int pb, int pq, int pL, int woff, int doff) __attribute__((always_inline)) {
            const float *ws = wtile + woff; float *dst = curX + doff;
            const bool pv = c0 + pq < N;
            constexpr uint32_t GM = (1u << GCB) - 1u;
            if (tailcol) {
                // row_sum columns: lane pL walks ITS sub-list of the sample's events (grouped by source index mod 4), lane 0
                // then adds the n % 4 leftover sources in order
                const uint32_t gc = gcnt[pb];
                const int st = (pL > 0 ? (int)(gc & GM) : 0) + (pL > 1 ? (int)((gc >> GCB) & GM) : 0) + (pL > 2 ? (int)((gc >> (2 * GCB)) & GM) : 0);
                const int nL = (int)((gc >> (GCB * pL)) & GM);
                const uint16_t *l2 = lst2 + pb * LX;
                const int n4 = Nin >> 2;
                int ix[8]; float wx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ix[u] = min((int)l2[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) wx[u] = ws[ix[u] * CW + pq];
                CascadeFlat a; a.init();
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < nL) a.add(ix[u] >> 2, wx[u] * 1.0f, n4);
                for (int u = 8; u < nL; ++u) {
                    const int i = (int)l2[st + u];
                    a.add(i >> 2, ws[i * CW + pq] * 1.0f, n4);
                }
                float v = a.finish(n4);
                if (pL == 0) {
                    const int s4 = (int)(gc & GM) + (int)((gc >> GCB) & GM) + (int)((gc >> (2 * GCB)) & GM) + (int)((gc >> (3 * GCB)) & GM);
                    const int n5 = (int)((gc >> (4 * GCB)) & GM);
                    for (int u = 0; u < n5; ++u) {
                        const int i = (int)l2[s4 + u];
                        v += ws[i * CW + pq] * 1.0f;
                    }
                }
                const float v1 = __shfl_down(v, 1, 4), v2 = __shfl_down(v, 2, 4), v3 = __shfl_down(v, 3, 4);
                const float e1 = ((v + v1) + v2) + v3;
                if (pL == 0 && pv) dst[pb * CW + pq] = 0.0f + e1;
            } else {
                // multi_row_sum columns: the cascade's 256-position groups are independent partial sums
                const uint32_t gq = gqn[pb];
                const int st = (pL > 0 ? (int)(gq & GM) : 0) + (pL > 1 ? (int)((gq >> GCB) & GM) : 0) + (pL > 2 ? (int)((gq >> (2 * GCB)) & GM) : 0);
                const int nL = (int)((gq >> (GCB * pL)) & GM);
                const uint16_t *lx = lstX + pb * LX;
                int ix[8]; float wx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ix[u] = min((int)lx[min(st + u, LX - 1)], Nin - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) wx[u] = ws[ix[u] * CW + pq];
                CascadeFlat a; a.init();
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < nL) a.add(ix[u], wx[u] * 1.0f, Nin);
                for (int u = 8; u < nL; ++u) {
                    const int ii2 = (int)lx[st + u];
                    a.add(ii2, ws[ii2 * CW + pq] * 1.0f, Nin);
                }
                const float G = a.a1 + a.a0;
                const float G1 = __shfl_down(G, 1, 4), G2 = __shfl_down(G, 2, 4), G3 = __shfl_down(G, 3, 4);
                if (pL == 0 && pv) {
                    const int GL = (Nin >> 4) >> 4;
                    const float Gs[4] = {G, G1, G2, G3};
                    float A2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < GL) A2 = A2 + Gs[k];
                    float Gl = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k == GL) Gl = Gs[k];
                    const float res = ((0.0f + Gl) + A2) + 0.0f;
                    dst[pb * CW + pq] = 0.0f + res;
                }
            }
        };
        // ... for the columns in `cols`, by the speculative waves: qt <-> (sample qt / 16, column (qt / 4) % 4, lane qt % 4)
        auto x_currents = [&](int qt, uint32_t cols, int woff) __attribute__((always_inline)) {
            const int pb = qt / (CW * 4), pq = (qt >> 2) % CW, pL = qt & 3;
            if (pb >= B || !((cols >> pq) & 1u)) return;
            x_current(pb, pq, pL, woff, 0);
        };

        // ------------------------------------------------------------------ t == 0: spikes of the step before the run
        if (!phaseA) {
            if (tid < BW) {
                uint32_t me = 0, mi = 0;
                for (int qq = 0; qq < 32; ++qq) {
                    const int jx = wj * 32 + qq;
                    if (jx < N) { me |= (uint32_t)(c.sE[wb * N + jx] != 0) << qq; mi |= (uint32_t)(c.sI[wb * N + jx] != 0) << qq; }
                }
                finE[tid] = me; spI[tid] = mi;
            }
            lds_barrier();
        }
        // ================================================================== window
        if (wave < NTW && t < T) fetch_digest(t + 1, 0, NTW);           // next iteration's digest, by the tile waves (waited for in front of barrier E)
        if (wave < W_Q0 + W_QN) {
            if (wave >= W_Q0) {
                const int qt = tid - W_Q0 * 64;
                if (do_stdp) {
                    spec_rows4(c, stdp_full, nact, arows, rowmask, xnu0, wtile, wtile + offN, c0, qt, QT);
                    part_barrier(&cnt[20], sub_q, W_QN);      // (the pollers must not be held up: no workgroup barrier here)
                }
                if (phaseB) x_currents(qt, 0xFu, do_stdp ? offN : 0);
                if (c.dbg && g == c.dbg_wg && tid == W_Q0 * 64) c.dbg[(size_t)t * 24 + 1] = (long long)wall_clock64();
            }
            if (prep) {
                // ---- the won branch of every own column with ONE crossing sample: the whole column from the committed weights
                //      (wtile -> wwin), 64-row chunks handed out through an LDS counter: the tile waves start at once, the
                //      speculative waves join when their own work is done.  Then that column's X currents (tile waves).
                const int nch = (Nin + 63) >> 6;
                uint32_t pcols = 0;
#pragma unroll
                for (int q = 0; q < CW; ++q) if (xm[q] && c0 + q < N) pcols |= 1u << q;
                const int ntot = __popc(pcols) * nch;
                for (;;) {
                    int ch = 0;
                    if (lane == 0) ch = atomicAdd(&cnt[24 + par], 1);
                    ch = __builtin_amdgcn_readfirstlane(ch);
                    if (ch >= ntot) break;
                    const int ci = ch / nch, rc = ch - ci * nch;
                    uint32_t pc = pcols;
                    for (int u = 0; u < ci; ++u) pc &= pc - 1;
                    const int q = __ffs(pc) - 1;
                    const uint32_t xq = q == 0 ? xm[0] : (q == 1 ? xm[1] : (q == 2 ? xm[2] : xm[3]));
                    const int bst = __ffs(xq) - 1;
                    const int i = rc * 64 + lane;
                    if (i < Nin) {
                        const float xval = xsrc[bst * Nin + i];
                        wtile[offW + i * CW + q] = postpre_elem(wtile[i * CW + q], i, q, rowmask[i], 1u << bst, bst, xwinv[4 * par + q], xsrc, true, xval);
                    }
                }
                part_barrier(&cnt[22], sub_t, NTW + W_QN);
                if (wave < NTW && phaseB) {
#pragma unroll
                    for (int q = 0; q < CW; ++q) {
                        if (!((pcols >> q) & 1u)) continue;
                        if ((tid >> 2) < B) x_current(tid >> 2, q, tid & 3, offW, offCW);
                    }
                }
            }
        } else if (wave >= W_P0 && wave < W_P0 + W_PN) {
            if (phaseA) {
                // ---- one thread per summary granule: poll it, decode its events into the bit words / Ai event lists /
                //      crossing-sample mask
                const unsigned long long *sums = c.exs + (size_t)par * NGS;
                const unsigned long long *exr = c.ex + (size_t)par * NG;
                for (int gi = tid - W_P0 * 64; gi < NGS; gi += W_PN * 64) {
                    unsigned long long x;
                    for (unsigned spins = 0;; ++spins) {
                        x = granule_load(sums + gi);
                        if ((uint32_t)(x >> 32) == (uint32_t)t || failed) break;
                        if (spins > kPollLimit) { failed = true; if (c.status) __hip_atomic_store(c.status, (int)SNN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    const uint32_t pay = (uint32_t)x;
                    if (!pay) continue;
                    uint32_t my_any = 0;
                    auto event = [&](int bsm, int jx, bool inh) __attribute__((always_inline)) {       // one crossing / inhibitory spike of step t-1
                        if (bsm >= B || jx >= N) return;
                        if (!inh) { atomicOr((unsigned int *)&crs[bsm * NW + (jx >> 5)], 1u << (jx & 31)); my_any |= 1u << bsm; }
                        else {
                            atomicOr((unsigned int *)&spI[bsm * NW + (jx >> 5)], 1u << (jx & 31));
                            const int slot = atomicAdd(&cntI[bsm], 1);
                            if (slot < LR) lstI[bsm * LR + slot] = (uint16_t)jx;
                            if (slot >= 1) atomicOr((unsigned int *)&misc[2], 2u);   // more than the one-entry fast path takes
                        }
                    };
                    const int gsrc = gi / NTW, w = gi - gsrc * NTW;
                    if ((pay & 0xFFu) == 0xFFu) {                        // overflow: that wave's full bit granules
                        cnt[18 + par] = 1;
                        for (int q = 0; q < SPW / SPG; ++q) {
                            const int k = w * (SPW / SPG) + q;
                            if (k >= KB) break;
                            unsigned long long d;
                            for (unsigned sp2 = 0;; ++sp2) {
                                d = granule_load(exr + gsrc * KB + k);
                                if ((uint32_t)(d >> 32) == (uint32_t)t || failed) break;
                                if (sp2 > kPollLimit) { failed = true; if (c.status) __hip_atomic_store(c.status, (int)SNN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            uint32_t be = (uint32_t)d & 0xFFFFu, bi = ((uint32_t)d >> 16) & 0xFFFFu;
                            while (be) { const int p_ = __ffs(be) - 1; be &= be - 1; event(k * SPG + p_ / CW, gsrc * CW + p_ % CW, false); }
                            while (bi) { const int p_ = __ffs(bi) - 1; bi &= bi - 1; event(k * SPG + p_ / CW, gsrc * CW + p_ % CW, true); }
                        }
                    } else {
                        const int ne = (int)(pay >> 30);
                        for (int e = 0; e < ne; ++e) {
                            const uint32_t ev = (pay >> (8 * e)) & 0xFFu;
                            const int p_ = (int)(ev & 0x3Fu);
                            event(w * SPW + p_ / CW, gsrc * CW + p_ % CW, (ev & 0x40u) != 0);
                        }
                    }
                    if (my_any) atomicOr((unsigned int *)&cnt[16 + par], my_any);
                }
            } else {
                // t == 0: lists / "winners" straight from the layers' spike bits (more than one spike per sample: give up)
                for (int b = wave - W_P0; b < B; b += W_PN) {
                    const int ni = build_list(spI + b * NW, NW, lane, lstI + b * LR, LR);
                    const int ne = build_list(finE + b * NW, NW, lane, lstE + b * LR, LR);
                    if (lane == 0) {
                        cntI[b] = ni;
                        if (ni > 1 || ne > 1) atomicOr((unsigned int *)&misc[2], 2u);
                        if (ne) { keys[b] = (unsigned long long)(0xFFFFFFFFu - (uint32_t)lstE[b * LR]); atomicOr((unsigned int *)&cnt[16], 1u << b); }
                    }
                }
            }
            if (c.dbg && g == c.dbg_wg && tid == W_P0 * 64) c.dbg[(size_t)t * 24 + 2] = (long long)wall_clock64();
        } else if (wave == W_AUX) {
            // scratch of the iterations to come: their last readers ended before barrier E, their writers start behind R
            if (lane < CW) { cnt[par * CW + lane] = 0; colmask[8 + 4 * (par ^ 1) + lane] = 0; }   // this step's crossing counts / crossing samples
            if (lane == 8) { cnt[16 + (par ^ 1)] = 0; cnt[18 + (par ^ 1)] = 0; cnt[24 + (par ^ 1)] = 0; }   // what the NEXT decode accumulates into; the next won-branch chunk counter
            for (int k = lane; k < BW; k += 64) crsA[(par ? 0 : offCB) + k] = 0;                      // ... its crossing words too
            if (t >= 1 && lane < MAXB) keys[lane] = 0ull;                   // (t == 0: the list pass above writes them)
            if (t >= 2) raster_rows(t - 2, (int)(spI2 - finE) + (par ^ 1) * kBitWords, W_AUX * 64, 64);
        } else if (wave == W_RNG) {
            if (use_rng)
                for (int m = ahead; m < RMK; ++m) mt_twist_block_wave(mt + ((mb + m) & RMK) * 624, mt + ((mb + m + 1) & RMK) * 624, lane);
            if (c.dbg && g == c.dbg_wg && tid == W_RNG * 64) c.dbg[(size_t)t * 24 + 3] = (long long)wall_clock64();
        }
        if (use_rng) ahead = RMK;
        if (c.dbg && g == c.dbg_wg && lane == 0) c.dbg[(size_t)t * 24 + 13] = c.dbg[(size_t)t * 24 + 13] | ((long long)1 << wave), c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + 200 + wave) * 4 + 3] = (long long)wall_clock64();   // arrival of every wave at R
        lds_barrier();                                                    // ---- R: exchange decoded, speculative results in place
        if (c.dbg && g == c.dbg_wg && threadIdx.x == 0) c.dbg[(size_t)t * 24 + 4] = (long long)wall_clock64();
        if (c.dbg && threadIdx.x == 0) { c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4] = (long long)wall_clock64(); c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 2] = (prep ? 1 : 0) + (slowcols ? 2 : 0) + 4 * __popc(xm[0] | xm[1] | xm[2] | xm[3]); }   // per workgroup: behind R; kind of window
        // ---- a step the lean form does not handle: every workgroup derives this from the same exchanged data / input
        //      digest, so all of them leave here in the same iteration and nobody is left waiting for a granule
        if ((__builtin_amdgcn_readfirstlane(misc[2]) & 2) || (mflags & 5)) {
            if (tid == 0 && c.status) __hip_atomic_store(c.status, (int)SNN_ERR_RETRY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            failed = true;
            break;
        }
        const uint32_t anym = (uint32_t)__builtin_amdgcn_readfirstlane(cnt[16 + par]);   // samples with an Ae crossing (t == 0: with an Ae spike)
        bool heavy = phaseA && __builtin_amdgcn_readfirstlane(cnt[18 + par]) != 0;
        int arb_rows = 0, arb_E = 0, arb_ntw = 0;
        if (use_rng) {
            arb_rows = __popc(anym);
            arb_E = rng_pos + 2 * arb_rows * N;
            arb_ntw = arb_rows ? (arb_E - 1) / 624 : 0;
            if (arb_ntw > RMK) heavy = true;
        }
        const bool fast = phaseA && !heavy && !slowcols && (diagE || !arb_rows);
        // one sample's arbitration by one wave (see k_dc2015_run for the reasoning); r = the sample's rank among the crossing ones
        auto arbitrate_sample = [&](int bsm, int r) __attribute__((always_inline)) {
            const uint32_t bits = lane < NW ? crs[bsm * NW + lane] : 0u;
            unsigned long long k1 = ~0ull, k2 = ~0ull;
            for (uint32_t bb = bits; bb; bb &= bb - 1) {
                const int jx = lane * 32 + __ffs(bb) - 1;
                const int w0 = rng_pos + 2 * (r * N + jx), w1 = w0 + 1;
                const int m0 = w0 / 624, m1 = w1 / 624;
                const uint32_t hi = mt_temper(mt[((mb + m0) & RMK) * 624 + w0 - 624 * m0]);
                const uint32_t lo = mt_temper(mt[((mb + m1) & RMK) * 624 + w1 - 624 * m1]);
                const unsigned long long m = (((unsigned long long)hi << 32) | lo) & ((1ull << 53) - 1ull);
                const unsigned long long key = (m << 10) | (unsigned long long)jx;
                if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
            }
            if (lane == 0) keys[bsm] = ~0ull;
            if (bits) atomicMin(&keys[bsm], k1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long kmin = *(volatile unsigned long long *)&keys[bsm];
            const unsigned long long mmin = kmin >> 10, zone = mmin + (mmin >> c.zone_shift) + 1ull;
            const bool close = (k1 != ~0ull && k1 != kmin && (k1 >> 10) <= zone) || (k2 != ~0ull && (k2 >> 10) <= zone);
            int win = (int)(kmin & 1023ull);
            if (__any(close)) {
                if (lane == 0) keys[bsm] = 0ull;
                for (uint32_t bb = bits; bb; bb &= bb - 1) {
                    const int jx = lane * 32 + __ffs(bb) - 1;
                    const int w0 = rng_pos + 2 * (r * N + jx), w1 = w0 + 1;
                    const int m0 = w0 / 624, m1 = w1 / 624;
                    const float q = exp1_from_words(mt_temper(mt[((mb + m0) & RMK) * 624 + w0 - 624 * m0]),
                                                    mt_temper(mt[((mb + m1) & RMK) * 624 + w1 - 624 * m1]));
                    const float val = 1.0f / q;
                    atomicMax(&keys[bsm], ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                win = (int)(0xFFFFFFFFu - (uint32_t)(*(volatile unsigned long long *)&keys[bsm] & 0xFFFFFFFFull));
            }
            if (lane == 0) { keys[bsm] = (unsigned long long)(0xFFFFFFFFu - (uint32_t)win); arbdone[bsm] = t; }
        };
        // final spike words of step t-1 (for the raster rows): the winner's bit, or nothing; word k <-> (sample k / NW, word k % NW)
        auto final_words = [&](int first_thread, int nthreads) __attribute__((always_inline)) {
            for (int k = tid - first_thread; k < BW; k += nthreads) {
                const int b = k / NW, w = k - b * NW;
                uint32_t wbits = 0;
                if ((anym >> b) & 1u) { const int win = win_of(b); if ((win >> 5) == w) wbits = 1u << (win & 31); }
                finE[k] = wbits;
            }
        };
        // what the NEXT decode accumulates into (its previous readers ended before barrier R)
        auto clear_next = [&](int first_thread, int nthreads) __attribute__((always_inline)) {
            if (phaseB) {
                for (int k = tid - first_thread; k < BW; k += nthreads) spI2[(par ^ 1) * kBitWords + k] = 0;
                if (tid >= first_thread && tid < first_thread + MAXB) cntI0[(par ? 0 : (int)(cntI1 - cntI0)) + tid - first_thread] = 0;
            }
        };
        // commit of step t-1's weights: wnew of the touched rows (and wwin of the columns in `wcols`) -> wtile
        auto commit = [&](int first_thread, int nthreads, uint32_t wcols) __attribute__((always_inline)) {
            if (!do_stdp) return;
            if (!wcols) {
                for (int k = tid - first_thread; k < nact; k += nthreads) {
                    const int i = stdp_full ? k : (int)arows[k];
                    *(float4 *)(wtile + i * 4) = *(const float4 *)(wtile + offN + i * 4);
                }
            } else {
                for (int i = tid - first_thread; i < Nin; i += nthreads) {
                    const bool touched = stdp_full || rowmask[i] != 0;
                    float4 v = *(const float4 *)(wtile + (touched ? offN : 0) + i * 4);
                    const float4 ww = *(const float4 *)(wtile + offW + i * 4);
                    if (wcols & 1u) v.x = ww.x;
                    if (wcols & 2u) v.y = ww.y;
                    if (wcols & 4u) v.z = ww.z;
                    if (wcols & 8u) v.w = ww.w;
                    *(float4 *)(wtile + i * 4) = v;
                }
            }
        };
        // own columns with a winner, derived by every wave from the stored winners (uniform result)
        auto own_winners = [&]() __attribute__((always_inline)) -> uint32_t {
            uint32_t ow = 0;
            if (phaseA && anym) {
                const int bsm = lane & 31;
                int q = -1;
                if (lane < 32 && bsm < B && ((anym >> bsm) & 1u)) { const int win = win_of(bsm); if (win >= c0 && win < c0 + CW) q = win - c0; }
#pragma unroll
                for (int qq = 0; qq < CW; ++qq) if (__ballot(q == qq)) ow |= 1u << qq;
            }
            return ow;
        };

        int rng_early = 0;                                                // generator wave: blocks refilled behind barrier R of this iteration
        float curE = 0.f, curI = 0.f;                                     // currents of step t of this tile thread's pair
        if (fast) {
            if (wave < NTW) {
                // ---- tile waves: did the crossing pair of my column win?  One candidate in its sample: yes, no draw needed;
                //      otherwise wait for that sample's arbitration (the other waves are at it)
                bool colwon = false; int bst = -1;
                const uint32_t xq = xm[0] * (jj == 0) + xm[1] * (jj == 1) + xm[2] * (jj == 2) + xm[3] * (jj == 3);
                if (xq && colv) {
                    bst = __ffs(xq) - 1;
                    int n = 0;
                    for (int w = 0; w < NW; ++w) n += __popc(crs[bst * NW + w]);
                    if (n == 1) colwon = true;
                    else {
                        for (unsigned spins = 0; __hip_atomic_load(&arbdone[bst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != t && spins < 100000000u; ++spins) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                        colwon = win_of(bst) == j;
                    }
                }
                if (tid < TT && bl < B && colv) {
                    const bool sp = colwon && bl == bst;
                    if (c.pE.lif.traces) stl[4 * TT + tid] = trace_next(stl[4 * TT + tid], sp, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive);
                    last_sE = sp;
                    if (phaseB) {
                        const int nI = cntI[bl];                           // Ai spikes of step t-1 in this sample: 0 or 1
                        const int iI = min((int)lstI[bl * LR], N - 1);
                        const float e2 = nI ? wieT[iI * CW + jj] * 1.0f + 0.0f : 0.0f;
                        const float e3 = sp ? weiT[j * CW + jj] * 1.0f + 0.0f : 0.0f;   // (own slice diagonal: only the own winner feeds Ai_j)
                        curE = curX[((colwon && do_stdp) ? offCW : 0) + bl * CW + jj] + e2;
                        curI = 0.0f + e3;
                    }
                }
            } else {
                // ---- the other 14 waves: arbitration of every crossing sample (generator position, rasters), commit
                if (arb_rows) {
                    int r = 0;
                    for (uint32_t rem = anym; rem; rem &= rem - 1, ++r) {
                        if ((r % NOW_) + NTW != wave) continue;
                        arbitrate_sample(__ffs(rem) - 1, r);
                    }
                    part_barrier(&cnt[21], sub_o, NOW_);
                }
                if (wave == W_RNG) {
                    // the generator wave takes no share of the commit: the blocks this step's arbitration consumed are free
                    // now, so it starts the refill here (up to kSpecEarlyTwists blocks, what fits before barrier E) instead
                    // of doing all of it in the next window, where it was the last wave to reach barrier R
                    if (!(c.spec_flags & 16)) {
                        const int mbn = (mb + arb_ntw) & RMK, ah = ahead - arb_ntw;        // (fast iteration: arb_ntw <= ahead)
                        for (int m = ah; m < RMK && rng_early < kSpecEarlyTwists; ++m, ++rng_early)
                            mt_twist_block_wave(mt + ((mbn + m) & RMK) * 624, mt + ((mbn + m + 1) & RMK) * 624, lane);
                    }
                } else {
                    clear_next(TT, NOT - 64);
                    final_words(TT, NOT - 64);
                    commit(TT, NOT - 64, own_winners());
                }
            }
        } else {
            // ---- slow iteration (and t == 0): the first generation's order
            clear_next(0, NT);
            if (use_rng) {
                if (!heavy && arb_rows) {
                    int r = 0;
                    for (uint32_t rem = anym; rem; rem &= rem - 1, ++r) {
                        if ((r % NWV) != wave) continue;
                        arbitrate_sample(__ffs(rem) - 1, r);
                    }
                }
                if (heavy && arb_ntw <= RMK && tid < BW) {
                    uint32_t bits = crs[tid];
                    const int myrank = __popc(anym & ((1u << (wb & 31)) - 1u));
                    while (bits) {
                        const int jx = wj * 32 + __ffs(bits) - 1; bits &= bits - 1;
                        const int d = myrank * N + jx;
                        const int w0 = rng_pos + 2 * d, w1 = w0 + 1;
                        const int m0 = w0 / 624, m1 = w1 / 624;
                        const float q = exp1_from_words(mt_temper(mt[((mb + m0) & RMK) * 624 + w0 - 624 * m0]),
                                                        mt_temper(mt[((mb + m1) & RMK) * 624 + w1 - 624 * m1]));
                        const float val = 1.0f / q;
                        const unsigned long long key = ((unsigned long long)__float_as_uint(val) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)jx);
                        atomicMax(&keys[wb], key);
                    }
                }
            }
            if (heavy || arb_rows > 0) lds_barrier();
            if (use_rng && arb_ntw > RMK) {
                const int myrank = __popc(anym & ((1u << (wb & 31)) - 1u));
                arbitrate_slow(mt, crs, keys, mb, rng_pos, N, arb_ntw, arb_rows, myrank, wb, wj, BW, tid, NT, RMK);
                ahead = arb_ntw;
                lds_barrier();
            }
            if (use_rng) final_words(0, NT);
            const uint32_t ownwin = own_winners();
            // Ae trace of step t-1 with the final spikes (nodes.py:96-103); winners also refresh x_tgt*nu0
            if (phaseA && tid < TT && bl < B && colv) {
                const bool sp = ((anym >> bl) & 1u) && win_of(bl) == j;
                if (c.pE.lif.traces) {
                    const float xn = trace_next(stl[4 * TT + tid], sp, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive);
                    stl[4 * TT + tid] = xn;
                    if (sp) xnu0[bl * CW + jj] = xn * c.nu0;
                }
                if (sp && do_stdp) atomicOr(&colmask[jj], 1u << bl);
                last_sE = sp;
            }
            if (ownwin && do_stdp) {
                // ---- repair: the winning column(s) only, every row from its committed weight
                lds_barrier();
                for (uint32_t cols = ownwin; cols; cols &= cols - 1) {
                    const int q = __ffs(cols) - 1;
                    if (c0 + q >= N) continue;
                    const uint32_t cm = colmask[q];
                    for (int i = tid; i < Nin; i += NT) {
                        const uint32_t m = rowmask[i];
                        const float w = postpre_elem(wtile[i * CW + q], i, q, m, cm, -1, 0.f, xsrc);
                        wtile[((stdp_full || m != 0) ? offN : 0) + i * CW + q] = w;
                    }
                }
                lds_barrier();
                if (phaseB && wave >= W_Q0 && wave < W_Q0 + W_QN) x_currents(tid - W_Q0 * 64, ownwin, offN);
                if (tid < CW) colmask[tid] = 0;
                lds_barrier();
            }
            if (wave >= NTW) commit(TT, NOT, 0u);
            if (mine && phaseB) {
                const int nI = cntI[bl];
                const int iI = min((int)lstI[bl * LR], N - 1);
                const float e2 = nI ? wieT[iI * CW + jj] * 1.0f + 0.0f : 0.0f;
                const bool hasE = (anym >> bl) & 1u;                       // final Ae spike of step t-1 in this sample: 0 or 1
                const int iE = hasE ? min(win_of(bl), N - 1) : 0;
                const float e3 = hasE ? weiT[iE * CW + jj] * 1.0f + 0.0f : 0.0f;
                curE = curX[bl * CW + jj] + e2;                            // (zeros + X->Ae) + Ai->Ae   (network.py:225-248)
                curI = 0.0f + e3;                                          // zeros + Ae->Ai
            }
        }
        if (use_rng) {                                                    // generator bookkeeping (every thread, from the crossing mask alone)
            if (fast || arb_ntw <= RMK) ahead -= arb_ntw; else ahead = 0;
            ahead += rng_early;                        // (generator wave only: blocks it has already put back behind barrier R)
            mb = (mb + arb_ntw) & RMK;
            rng_pos = arb_E - 624 * arb_ntw;
            rng_consumed += (long long)arb_rows * N;
        }
        if (c.dbg && g == c.dbg_wg && threadIdx.x == 0) { c.dbg[(size_t)t * 24 + 5] = (long long)wall_clock64(); c.dbg[(size_t)t * 24 + 10] = arb_rows; c.dbg[(size_t)t * 24 + 11] = arb_ntw + (heavy ? 100 : 0); c.dbg[(size_t)t * 24 + 12] = fast ? 0 : 1; c.dbg[(size_t)t * 24 + 6] = c.dbg[(size_t)t * 24 + 5]; }
        if (!phaseB) break;

        // ================================================================== start step t (tile threads)
        bool spE = false, spIn = false;
        float r_vE = 0.f, r_vI = 0.f;
        if (mine) {
            float r_rE = stl[1 * TT + tid], r_rI = stl[3 * TT + tid], th = stl[6 * TT + tid];
            r_vE = stl[0 * TT + tid]; r_vI = stl[2 * TT + tid];
            if (c.pE.learning && t >= 1) th = th + c.pE.theta_plus * (float)cnt[(par ^ 1) * CW + jj];
            if (c.pE.learning) th = th * c.pE.theta_decay;
            spE = dc_update(r_vE, r_rE, curE, c.pE.lif.thresh + th, c.pE.lif);
            if (spE) atomicAdd(&cnt[par * CW + jj], 1);
            float ci = curI;
            if (r_rI > 0.f) ci = 0.f;
            spIn = lif_update(r_vI, r_rI, ci, c.pI);
            last_sI = spIn;
            stl[0 * TT + tid] = r_vE; stl[1 * TT + tid] = r_rE; stl[2 * TT + tid] = r_vI; stl[3 * TT + tid] = r_rI; stl[6 * TT + tid] = th;
            if (c.pI.traces) stl[5 * TT + tid] = trace_next(stl[5 * TT + tid], spIn, c.pI.trace_decay, c.pI.trace_scale, c.pI.traces_additive);
        }
        if (wave < NTW) {
            // publish step t as epoch t+1: ONE summary granule per tile wave (see k_dc2015_run)
            const uint64_t mE = __ballot(spE), mI = __ballot(spIn);
            const int nev = __popcll(mE) + __popcll(mI);
            uint32_t pay;
            if (nev <= 3) {
                pay = (uint32_t)nev << 30;
                int sh = 0;
                for (uint64_t m = mE; m; m &= m - 1) { pay |= (uint32_t)(__ffsll((unsigned long long)m) - 1) << sh; sh += 8; }
                for (uint64_t m = mI; m; m &= m - 1) { pay |= (uint32_t)(0x40 | (__ffsll((unsigned long long)m) - 1)) << sh; sh += 8; }
            } else {
                const int sidx = lane / CW, b = wave * SPW + sidx;
                const uint32_t v = (uint32_t)((mE >> (sidx * CW)) & 0xFFFFull) | ((uint32_t)((mI >> (sidx * CW)) & 0xFFFFull) << 16);
                if ((lane % CW) == 0 && (sidx % SPG) == 0 && b < B)
                    granule_store(c.ex + (size_t)(par ^ 1) * NG + g * KB + b / SPG, ((unsigned long long)(uint32_t)(t + 1) << 32) | v);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pay = 0xC0FFFFFFu;
            }
            if (lane == 0) granule_store(c.exs + (size_t)(par ^ 1) * NGS + g * NTW + wave, ((unsigned long long)(uint32_t)(t + 1) << 32) | pay);
            if (c.dbg && tid == 0) c.dbg[(size_t)24 * (T + 1) + ((size_t)t * 256 + g) * 4 + 1] = (long long)wall_clock64();          // per workgroup: published
        }
        if (tid < TT && bl < B) {
            // for the next window: x_tgt*nu0 of step t as it is if this pair does not win; a crossing pair also leaves its
            // sample in the column's crossing mask and the value it has if it wins
            float xs = 0.f;
            if (colv && c.pE.lif.traces) xs = trace_next(stl[4 * TT + tid], 0, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive);
            xnu0[bl * CW + jj] = xs * c.nu0;
            if (spE) {
                atomicOr(&colmask[8 + 4 * (par ^ 1) + jj], 1u << bl);
                if (c.pE.lif.traces) xwinv[4 * (par ^ 1) + jj] = trace_next(stl[4 * TT + tid], 1, c.pE.lif.trace_decay, c.pE.lif.trace_scale, c.pE.lif.traces_additive) * c.nu0;
            }
        }
        if (mine) {
            if (c.rasVE) (c.rasVE + (size_t)t * B * N)[kst] = r_vE;
            if (c.rasVI) (c.rasVI + (size_t)t * B * N)[kst] = r_vI;
        }
        if (c.dbg && g == c.dbg_wg && threadIdx.x == 0) { c.dbg[(size_t)t * 24 + 7] = (long long)wall_clock64(); c.dbg[(size_t)t * 24 + 9] = (long long)clock64(); }
        if (c.dbg && threadIdx.x == 0) atomicMax((unsigned long long *)&c.dbg[(size_t)t * 24 + 21], (unsigned long long)wall_clock64());
        if (wave < NTW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next iteration's digest has landed
        lds_barrier();                                                    // ---- E
    }

    // ---- epilogue (as k_dc2015_run's): nothing the caller owns as STATE has been written so far
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a digest fetch into LDS may still be in flight when the loop was left early:
                                                         //  it must not land after this workgroup's LDS has been handed to another one)
    if (tid == 0) misc[4] = c.status ? __hip_atomic_load(c.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (failed) misc[5] = 1;
    __syncthreads();
    if (misc[4] != 0 || misc[5] != 0) return;
    raster_rows(T - 1, (int)(spI2 - finE) + (T & 1) * kBitWords, 0, NT);            // the last step's rows (the aux wave writes a step's rows two iterations later)
    if (c.x_traces) {
        const float *src = c.xtr + (size_t)T * B * Nin;
        for (int k = g * NT + tid; k < B * Nin; k += c.G * NT) c.xX[1][k] = src[k];
    }
    if (mine) {
        float th = stl[6 * TT + tid];
        if (c.pE.learning) th = th + c.pE.theta_plus * (float)cnt[((T - 1) & 1) * CW + jj];
        c.vE[kst] = stl[0 * TT + tid]; c.rE[kst] = stl[1 * TT + tid]; c.vI[kst] = stl[2 * TT + tid]; c.rI[kst] = stl[3 * TT + tid];
        if (bl == 0) c.theta[j] = th;
        if (c.pI.traces) c.xI[kst] = stl[5 * TT + tid];
        if (c.pE.lif.traces) c.xE[kst] = stl[4 * TT + tid];
        c.sE[kst] = last_sE; c.sI[kst] = last_sI;
    }
    if (c.has_norm) {
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
    if (g == 0) {
        snn_rng_state *wr = c.rng[0];
        for (int k = tid; k < 624; k += NT) wr->mt[k] = mt[mb * 624 + k];
        if (tid == 0) { wr->pos = rng_pos; wr->consumed = rng_consumed; }
    }
}

// words of one digest entry / of its part that the step kernel copies into LDS
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

}  // namespace

size_t snn_dc2015_resident_lds(int B, int Nin, int N, int cw) { return lds_bytes_resident(B, Nin, N, cw); }
// second-generation lean form (k_dc2015_spec): 16-block generator ring, speculative / won-branch weight buffers (spec_tail_lds)
size_t snn_dc2015_spec_lds(int B, int Nin, int N) { return lds_bytes_resident(B, Nin, N, 4) + (size_t)(kSpecRing - 8) * 624 * 4 + spec_tail_lds(Nin); }
int snn_dc2015_resident_cw(int N) { return resident_cw(N); }
int snn_dc2015_resident_nt() { return resident_nt(); }

static const void *resident_variant(int cw, int nt, int lean = 0) {   // lean: 0 general form, 1 lean form, 2 its second generation
    if (lean == 2) return (const void *)k_dc2015_spec<1024>;
    if (lean) return (const void *)k_dc2015_run<4, 1024, true>;         // the lean forms exist for 4-column tiles only
    if (nt == 512 && cw == 4) return (const void *)k_dc2015_run<4, 512, false>;
    if (nt == 512) return (const void *)k_dc2015_run<2, 512, false>;
    if (cw == 8) return (const void *)k_dc2015_run<8, 1024, false>;
    if (cw == 4) return (const void *)k_dc2015_run<4, 1024, false>;
    return (const void *)k_dc2015_run<2, 1024, false>;
}

static bool resident_attr_once() {
    static int state = 0;          // 0 = not tried, 1 = ok, -1 = failed
    if (!state) {   // the kernel uses more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        state = 1;
        const int cws[7] = {8, 4, 2, 4, 2, 4, 4}, nts[7] = {1024, 1024, 1024, 512, 512, 1024, 1024}, lns[7] = {0, 0, 0, 0, 0, 1, 2};
        for (int k = 0; k < 7; ++k)
            if (snn_check(hipFuncSetAttribute(resident_variant(cws[k], nts[k], lns[k]), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))) state = -1;
    }
    return state == 1;
}

// How many workgroups of this variant the current device is guaranteed to hold at once (CUs x workgroups per CU
// from the occupancy calculator; 0 when the device cannot launch cooperatively).  The spin-wait hand-off between
// workgroups is only correct when the whole grid is co-resident, so the plan is refused beyond this number.
int snn_dc2015_resident_capacity(int cw, int nt, size_t lds) {
    if (!resident_attr_once()) return 0;
    int dev = 0, cus = 0, coop = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess) coop = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, resident_variant(cw, nt), nt, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (getenv("SNN_DC_FAKE_CUS")) cus = atoi(getenv("SNN_DC_FAKE_CUS"));     // test hook: pretend to be a smaller device
    return coop ? cus * per_cu : 0;
}

// Cooperative launch: the runtime itself refuses (hipErrorCooperativeLaunchTooLarge) a grid it cannot make co-resident.
// Returns SNN_ERR_UNSUPPORTED in that case so that the caller takes the one-launch-per-timestep plan instead.
int snn_dc2015_resident_launch(const DcCtx &c, int cw, int nt, size_t lds, int lean, hipStream_t st, bool ordinary) {
    if (!resident_attr_once()) return SNN_ERR_LAUNCH;
    static const bool coop_env = !(getenv("SNN_DC_COOP") && atoi(getenv("SNN_DC_COOP")) == 0);
    const bool coop = coop_env && !ordinary;
    DcCtx arg = c;
    void *args[1] = {(void *)&arg};
    const void *fn = resident_variant(cw, nt, lean);
    if (!coop)         // developer switch: ordinary launch (co-residency then rests on snn_dc2015_resident_capacity alone)
        return snn_check(hipLaunchKernel(fn, dim3(c.G), dim3(nt), args, lds, st));
    const int gmult = (lean == 2 && (c.spec_flags >> 8) > 1) ? (c.spec_flags >> 8) : 1;
    const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(c.G * gmult), dim3(nt), args, (unsigned)lds, st);
    if (e == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); return SNN_ERR_UNSUPPORTED; }
    return snn_check(e);
}

// snn_order.hpp -- event-driven restatement of ATen's CPU float-sum order for gfx950 kernels.
//
// The reference (BindsNET on PyTorch-CPU) reduces with aten/src/ATen/native/cpu/SumKernel.cpp:
//   multi_row_sum : 4-level cascade, 16-term blocks summed sequentially, block sums added
//                   sequentially, flushed upward every 16 blocks / 256 blocks (level power 4,
//                   valid while the term count is <= 2^19);
//   row_sum       : 4 interleaved lanes (term index mod 4), each a multi_row_sum over n/4 terms,
//                   n%4 leftovers added to lane 0, lanes combined ((l0+l1)+l2)+l3;
//   vectorized_outer_sum: of `ncols` contiguous output columns, those below 32*floor(ncols/32)
//                   use multi_row_sum, the rest row_sum (serial order).
// Spikes are sparse, so the kernels feed only the NON-ZERO terms, in ascending term index.
// Skipping a zero term is exact (x + 0.0f == x), but the block structure is defined on the term
// INDEX, so the accumulators must still be flushed at the same index boundaries -- that is what
// Cascade::advance() does.  Built with -ffp-contract=off: every add below is one rounding.
// The accumulators are __host__ __device__: tests/hostcheck/order_rng_host.hip runs them on the CPU, term by term as the
// kernels' threads do, against the oracle (tests/test_order_rng_host.py) -- the build container has no GPU.
#pragma once
#include <hip/hip_runtime.h>

namespace snn {

constexpr int kMaxTerms = 1 << 19;  // level power stays 4 up to here

struct Cascade {
    float a0, a1, a2, a3;
    int cb;  // index of the 16-term block whose partial sum sits in a0 (-1: none yet)

    __host__ __device__ __forceinline__ void init() { a0 = a1 = a2 = a3 = 0.f; cb = -1; }

    // Close block `cb` and move to block nb > cb.  A level-1 flush happens at every block
    // boundary that is a multiple of 16 blocks inside (cb, nb], a level-2 flush at multiples of
    // 256; repeated flushes of an already-zero accumulator are no-ops, so one test per level
    // is enough.
    __host__ __device__ __forceinline__ void advance(int nb) {
        a1 += a0; a0 = 0.f;
        if ((nb >> 4) != (cb >> 4)) {
            a2 += a1; a1 = 0.f;
            if ((nb >> 8) != (cb >> 8)) { a3 += a2; a2 = 0.f; }
        }
        cb = nb;
    }

    // pos: index of the term in the full (dense) sequence; nfull: number of complete blocks
    // (n >> 4).  Terms past the last complete block all land in pseudo-block `nfull`.
    __host__ __device__ __forceinline__ void add(int pos, float term, int nfull) {
        int blk = pos >> 4;
        blk = blk < nfull ? blk : nfull;
        if (blk != cb) advance(blk);
        a0 += term;
    }

    __host__ __device__ __forceinline__ float finish(int nfull) {
        if (cb != nfull) advance(nfull);
        return ((a0 + a1) + a2) + a3;
    }
};

struct RowSum4 {
    Cascade l0, l1, l2, l3;
    float L0;
    bool closed0;

    __host__ __device__ __forceinline__ void init() { l0.init(); l1.init(); l2.init(); l3.init(); L0 = 0.f; closed0 = false; }

    // n: total number of terms of the dense sequence.
    __host__ __device__ __forceinline__ void add(int pos, float term, int n) {
        const int n4 = n >> 2, nfull4 = n4 >> 4;
        if (pos >= (n4 << 2)) {  // leftovers: added to lane 0 after its cascade has been combined
            if (!closed0) { L0 = l0.finish(nfull4); closed0 = true; }
            L0 += term;
            return;
        }
        const int p = pos >> 2;
        switch (pos & 3) {
            case 0: l0.add(p, term, nfull4); break;
            case 1: l1.add(p, term, nfull4); break;
            case 2: l2.add(p, term, nfull4); break;
            default: l3.add(p, term, nfull4); break;
        }
    }

    __host__ __device__ __forceinline__ float finish(int n) {
        const int nfull4 = (n >> 2) >> 4;
        if (!closed0) { L0 = l0.finish(nfull4); closed0 = true; }
        float r = L0 + l1.finish(nfull4);
        r = r + l2.finish(nfull4);
        r = r + l3.finish(nfull4);
        return r;
    }
};

// Branch-free Cascade for sums of fewer than 4096 terms (the level-3 accumulator is never reached):
// same arithmetic as Cascade, written with selects so a wave whose lanes are at different points of
// different samples' event lists does not diverge.
struct CascadeFlat {
    float a0, a1, a2;
    int cb;
    __host__ __device__ __forceinline__ void init() { a0 = a1 = a2 = 0.f; cb = -1; }
    __host__ __device__ __forceinline__ void add(int pos, float term, int n) {
        const int nfull = n >> 4;
        int blk = pos >> 4;
        blk = blk < nfull ? blk : nfull;
        const bool nb = blk != cb, ng = (blk >> 4) != (cb >> 4);
        const float s1 = a1 + a0;              // what a1 becomes if block cb is closed
        a1 = nb ? s1 : a1;
        a0 = nb ? 0.f : a0;
        const float s2 = a2 + a1;              // ... and a2 if a 256-term group is closed too
        a2 = ng ? s2 : a2;
        a1 = ng ? 0.f : a1;
        cb = blk;
        a0 += term;
    }
    __host__ __device__ __forceinline__ float finish(int n) {
        const int nfull = n >> 4;
        if (cb != nfull) {
            a1 += a0; a0 = 0.f;
            if ((nfull >> 4) != (cb >> 4)) { a2 += a1; a1 = 0.f; }
        }
        return ((a0 + a1) + a2) + 0.0f;
    }
};

// Cascade with the (pos, term, n) interface of RowSum4, for code templated on the column class.
struct CascadeN {
    Cascade c;
    __host__ __device__ __forceinline__ void init() { c.init(); }
    __host__ __device__ __forceinline__ void add(int pos, float term, int n) { c.add(pos, term, n >> 4); }
    __host__ __device__ __forceinline__ float finish(int n) { return c.finish(n >> 4); }
};

// One output element's reduction; `tail` selects row_sum (column >= 32*floor(ncols/32)).
struct OuterSum {
    Cascade c;
    RowSum4 r;
    bool tail;
    __host__ __device__ __forceinline__ void init(bool is_tail) { tail = is_tail; c.init(); if (is_tail) r.init(); }
    __host__ __device__ __forceinline__ void add(int pos, float term, int n) {
        if (!tail) c.add(pos, term, n >> 4); else r.add(pos, term, n);
    }
    __host__ __device__ __forceinline__ float finish(int n) { return tail ? r.finish(n) : c.finish(n >> 4); }
};

// ATen's vectorised INNER sum of one contiguous row of n floats -- what `w[filter].sum(0)` of Conv2dConnection.normalize
// (bindsnet/network/topology.py:824-837) runs: SumKernel.cpp's vectorized_inner_sum with 8-float vectors (torch 2.10 takes that
// path under every ATEN_CPU_CAPABILITY; probed).  Lane l of 8 sums x[l], x[8 + l], ... through row_sum over the n / 8 vectors
// (= RowSum4); then a fresh accumulator takes the n mod 8 leftover elements in order and after them the 8 lane sums in lane
// order.  Rows shorter than one vector: scalar row_sum over the n terms.
__host__ __device__ inline float inner_sum8(const float *x, int n) {
    if (n < 8) {
        RowSum4 r;
        r.init();
        for (int i = 0; i < n; ++i) r.add(i, x[i], n);
        return r.finish(n);
    }
    const int vs = n >> 3;
    float part[8];
    for (int l = 0; l < 8; ++l) {
        RowSum4 r;
        r.init();
        for (int i = 0; i < vs; ++i) r.add(i, x[i * 8 + l], vs);
        part[l] = r.finish(vs);
    }
    float fin = 0.f;
    for (int k = vs * 8; k < n; ++k) fin += x[k];
    for (int l = 0; l < 8; ++l) fin += part[l];
    return fin;
}

// Plain ascending sequential sum (canonical order of the dense Connection path).
struct SeqSum {
    float a;
    __host__ __device__ __forceinline__ void init(bool) { a = 0.f; }
    __host__ __device__ __forceinline__ void add(int, float term, int) { a += term; }
    __host__ __device__ __forceinline__ float finish(int) { return a; }
};

}  // namespace snn

// tools/probe_latency.hip -- MI355X micro-measurements behind the third-generation D&C resident kernel:
//   1. mt19937 twist of one 624-word block by ONE wave in LDS: the current form (csrc/snn_rng.hpp) against a read2 / unclamped
//      form, alone on its SIMD and beside spinning waves, with and without s_setprio
//   2. LDS-counter barrier among 2 / 4 / 8 waves, s_barrier of 4 / 8 / 16 waves
//   3. all-to-all granule exchange: G workgroups publish `per` 8-byte tagged granules per epoch and poll everybody's, nothing else
//      (the exchange-bound period of the resident kernels), and a 2-workgroup ping-pong (one-way latency), same / other XCD
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_latency tools/probe_latency.hip && tools/bin/probe_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../bindsnet_amd/csrc/snn_rng.hpp"

using namespace snn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- twist variants -------------------------------------------------------------------------------------------------
// v1: no clamps (the ring is padded by the caller: src[624] readable), ds_read2-friendly (src[i], src[i+1] adjacent)
__device__ __forceinline__ void twist_v1(const uint32_t *src, uint32_t *dst, int lane) {
    {
        uint32_t a[4], a1[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; const int ic = i < 227 ? i : 226; a[k] = src[ic]; a1[k] = src[ic + 1]; b[k] = src[ic + 397]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; if (i < 227) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
    {
        uint32_t a[4], a1[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = 227 + lane + 64 * k; const int ic = i < 454 ? i : 453; a[k] = src[ic]; a1[k] = src[ic + 1]; b[k] = dst[ic - 227]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = 227 + lane + 64 * k; if (i < 454) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
    {
        uint32_t a[3], a1[3], b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int i = 454 + lane + 64 * k; const int ic = i < 624 ? i : 623; a[k] = src[ic]; a1[k] = (ic == 623) ? dst[0] : src[ic + 1]; b[k] = dst[ic - 227]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int i = 454 + lane + 64 * k; if (i < 624) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
}

// v2: the stream as ONE sequence x[n] = x[n-227] ^ mix(x[n-624], x[n-623]) produced in stripes of 192 = 3 x 64 words over a ring of
// RW words (a multiple of 192 and of 624 is not needed: positions are taken modulo RW, RW a power of two >= 2048): 3.25 stripes per
// 624 words instead of 3 phases with 11 chunk-passes; every stripe is 3 independent chunk-passes
__device__ __forceinline__ void twist_stream192(uint32_t *ring, unsigned RWm, unsigned n0, int lane) {   // words n0 .. n0+191
    uint32_t a[3], a1[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const unsigned n = n0 + lane + 64 * k; a[k] = ring[(n - 624) & RWm]; a1[k] = ring[(n - 623) & RWm]; b[k] = ring[(n - 227) & RWm]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { const unsigned n = n0 + lane + 64 * k; ring[n & RWm] = b[k] ^ mt_mix(a[k], a1[k]); }
}

__global__ __launch_bounds__(1024) void k_twist(int variant, int nblocks, int nspin, int prio, const uint32_t *seed_state, uint32_t *out_state,
                                                long long *out_cycles, int sleepy) {
    extern __shared__ uint32_t lds[];
    uint32_t *ring = lds;                       // 16 blocks of 624 (+ pad) or a 16384-word stream ring
    int *stop = (int *)(lds + 16384);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < 624; k += blockDim.x) ring[k] = seed_state[k];
    if (tid == 0) *stop = 0;
    __syncthreads();
    if (wave == 0) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        const long long t0 = clock64();
        if (variant == 0) for (int m = 0; m < nblocks; ++m) mt_twist_block_wave(ring + (m & 15) * 624, ring + ((m + 1) & 15) * 624, lane);
        else if (variant == 1) for (int m = 0; m < nblocks; ++m) twist_v1(ring + (m & 15) * 624, ring + ((m + 1) & 15) * 624, lane);
        else {
            const unsigned total = (unsigned)nblocks * 624u;
            for (unsigned n = 624; n < 624 + total; n += 192) twist_stream192(ring, 16383u, n, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long t1 = clock64();
        if (lane == 0) { out_cycles[0] = t1 - t0; __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (variant <= 1) { for (int k = lane; k < 624; k += 64) out_state[k] = ring[(nblocks & 15) * 624 + k]; }
        else { for (int k = lane; k < 624; k += 64) out_state[k] = ring[((unsigned)nblocks * 624u + k) & 16383u]; }
    } else if (wave <= nspin) {
        // spinning neighbours: the tight LDS polling loop of the resident kernels' partial barriers
        while (!__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { if (sleepy) __builtin_amdgcn_s_sleep(1); }
    }
}

// ---- barriers -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_barriers(int mode, int nwaves, int iters, long long *out) {
    __shared__ int ctr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) ctr = 0;
    __syncthreads();
    if (wave >= nwaves) return;
    int target = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else {
            target += nwaves;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) { if (mode == 2) __builtin_amdgcn_s_sleep(1); }
            asm volatile("" ::: "memory");
        }
    }
    const long long t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
}

// ---- exchange -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long gload(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every workgroup: epoch e = 1..E: publish `per` granules (tag e) into ring slot e & 3, poll all G * per granules of the slot.
// lag = 0: lock step (publish e, wait for everybody's e); lag = 1: publish e, wait for everybody's e - 1 (one epoch of slack)
__global__ __launch_bounds__(256) void k_allgather(unsigned long long *gr, int G, int per, int epochs, int lag, int gmult, long long *out, int *xcc) {
    if ((int)blockIdx.x % gmult) return;
    const int g = (int)blockIdx.x / gmult, tid = threadIdx.x;
    const int NG = G * per;
    if (tid == 0) xcc[g] = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF);
    __shared__ int sink;
    const long long t0 = wall_clock64();
    for (int e = 1; e <= epochs; ++e) {
        if (tid < per) gstore(gr + (size_t)(e & 3) * NG + g * per + tid, ((unsigned long long)(uint32_t)e << 32) | (uint32_t)(g + 1));
        const int we = e - lag;
        if (we >= 1) {
            int acc = 0;
            for (int gi = tid; gi < NG; gi += 256) {
                unsigned long long x;
                unsigned spins = 0;
                do { x = gload(gr + (size_t)(we & 3) * NG + gi); } while ((uint32_t)(x >> 32) != (uint32_t)we && ++spins < 4000000u);
                acc += (int)(uint32_t)x;
            }
            if (acc == 12345678) sink = acc;
        }
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) out[g] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_pingpong(unsigned long long *f, int a, int b, int iters, long long *out, int *xcc) {
    const int g = blockIdx.x;
    if (g != a && g != b) return;
    if (threadIdx.x != 0) return;
    xcc[g == a ? 0 : 1] = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF);
    const long long t0 = wall_clock64();
    if (g == a) {
        for (int i = 1; i <= iters; ++i) { gstore(f, (unsigned long long)i); unsigned s = 0; while (gload(f + 16) != (unsigned long long)i && ++s < 40000000u) {} }
        out[0] = wall_clock64() - t0;
    } else {
        for (int i = 1; i <= iters; ++i) { unsigned s = 0; while (gload(f) != (unsigned long long)i && ++s < 40000000u) {} gstore(f + 16, (unsigned long long)i); }
    }
}

// host mt19937 block twist
static void host_twist(uint32_t *mt) {
    for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    printf("{\"clock_khz\": %d, \"wall_clock_khz\": %d}\n", clk_khz, wall_khz);
    const double cyc_us = clk_khz / 1000.0;   // shader cycles per us at the reported (peak) clock: an upper bound of the real rate
    // ---- 1. twist
    std::vector<uint32_t> st(624), ref(624), got(624);
    st[0] = 5489u; for (int i = 1; i < 624; ++i) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + i;
    uint32_t *d_seed, *d_out; long long *d_cyc; CK(hipMalloc(&d_seed, 624 * 4)); CK(hipMalloc(&d_out, 624 * 4)); CK(hipMalloc(&d_cyc, 4096 * 8));
    CK(hipMemcpy(d_seed, st.data(), 624 * 4, hipMemcpyHostToDevice));
    const int nblocks = 2000;
    ref = st; for (int m = 0; m < nblocks; ++m) host_twist(ref.data());
    CK(hipFuncSetAttribute((const void *)k_twist, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (int variant = 0; variant < 3; ++variant)
        for (int nspin = 0; nspin <= 15; nspin += (nspin == 0 ? 3 : 12))
            for (int prio = 0; prio < 2; ++prio)
                for (int sleepy = 0; sleepy < 2; ++sleepy) {
                    if (nspin == 0 && (prio || sleepy)) continue;
                    long long cyc = 0; double best = 1e30;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipLaunchKernelGGL(k_twist, dim3(1), dim3(1024), 16400 * 4, 0, variant, nblocks, nspin, prio, d_seed, d_out, d_cyc, sleepy);
                        CK(hipDeviceSynchronize());
                        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
                        if ((double)cyc < best) best = (double)cyc;
                    }
                    CK(hipMemcpy(got.data(), d_out, 624 * 4, hipMemcpyDeviceToHost));
                    const bool ok = memcmp(got.data(), ref.data(), 624 * 4) == 0;
                    printf("{\"probe\": \"twist\", \"variant\": %d, \"spinning_waves\": %d, \"setprio\": %d, \"spin_sleep\": %d, \"cycles_per_block\": %.1f, \"us_per_block_at_peak_clock\": %.3f, \"correct\": %s}\n",
                           variant, nspin, prio, sleepy, best / nblocks, best / nblocks / cyc_us, ok ? "true" : "false");
                }
    // ---- 2. barriers
    for (int mode = 0; mode < 3; ++mode)
        for (int nw = 2; nw <= 16; nw *= 2) {
            long long cyc = 0;
            hipLaunchKernelGGL(k_barriers, dim3(1), dim3(1024), 0, 0, mode, nw, 2000, d_cyc);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            printf("{\"probe\": \"barrier\", \"kind\": \"%s\", \"waves\": %d, \"cycles\": %.1f}\n", mode == 0 ? "s_barrier" : (mode == 1 ? "lds_counter_tight" : "lds_counter_sleep"), nw, cyc / 2000.0);
        }
    // ---- 3. exchange
    unsigned long long *d_gr; int *d_xcc; CK(hipMalloc(&d_gr, 4 * 256 * 8 * 8)); CK(hipMalloc(&d_xcc, 256 * 4));
    std::vector<long long> hout(256); std::vector<int> hx(256);
    const double wall_us = wall_khz / 1000.0;
    for (int G : {100, 25, 200})
        for (int per : {1, 2, 8})
            for (int lag = 0; lag < 2; ++lag)
                for (int gmult : {1, 2}) {
                    if (G * gmult > 256) continue;
                    const int epochs = 2000;
                    CK(hipMemset(d_gr, 0, 4 * 256 * 8 * 8));
                    void *args[] = {&d_gr, (void *)&G, (void *)&per, (void *)&epochs, (void *)&lag, (void *)&gmult, &d_cyc, &d_xcc};
                    CK(hipLaunchCooperativeKernel((const void *)k_allgather, dim3(G * gmult), dim3(256), args, 0, 0));
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(hout.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost));
                    long long mx = 0; for (int g = 0; g < G; ++g) if (hout[g] > mx) mx = hout[g];
                    printf("{\"probe\": \"allgather\", \"workgroups\": %d, \"granules_per_wg\": %d, \"lag\": %d, \"grid_mult\": %d, \"us_per_epoch\": %.3f}\n", G, per, lag, gmult, mx / wall_us / epochs);
                }
    for (int b : {1, 8, 9, 33, 64}) {
        const int iters = 2000, a = 0;
        CK(hipMemset(d_gr, 0, 1024));
        hipLaunchKernelGGL(k_pingpong, dim3(128), dim3(64), 0, 0, d_gr, a, b, iters, d_cyc, d_xcc);
        CK(hipDeviceSynchronize());
        long long w = 0; CK(hipMemcpy(&w, d_cyc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), d_xcc, 8, hipMemcpyDeviceToHost));
        printf("{\"probe\": \"pingpong\", \"block_a\": %d, \"block_b\": %d, \"xcc_a\": %d, \"xcc_b\": %d, \"round_trip_us\": %.3f}\n", a, b, hx[0], hx[1], w / wall_us / iters);
    }
    return 0;
}

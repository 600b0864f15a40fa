// tools/probe_won_branch.hip -- stand-alone timing of the won-branch row blocks of k_dc2015_async ("what a
// crossing iteration's extra 3 us is bound by"): ONE workgroup of 512 threads with the compute workgroup's LDS arrays (wtile, wbak,
// wwin [Nin][4], rowmask, x_tgt*nu0), the X trace in global memory laid out as the pre-pass leaves it ([steps][B][Nin], written by
// another kernel a moment earlier), one crossing column with one crossing sample.  Each variant runs `reps` times on a fresh
// (step, sample) row of the trace with ~2 us of sleep in between, timed by wave 2 / wave 0 with the 100 MHz wall clock:
//   A  untouched rows, the product's loop: a thread's rows one after the other, each with its own X-trace load
//   B  untouched rows, the three values asked for together, then the rows
//   C  touched rows, the product's loop (one row per thread, threads 0..nact-1)
//   D  touched rows with the X-trace value asked for ~1 us ahead (the upper bound of what a prefetch could give)
//   E  A without the global load (the value comes from LDS): what the rows cost without the trace
//   F  untouched rows with every read of the thread's three rows first and the one-term cascade written out
// postpre_elem is the product's (the kernel source is included); nothing here is part of libsnnhip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Ibindsnet_amd/csrc -o tools/bin/probe_won_branch \
//         tools/probe_won_branch.hip bindsnet_amd/csrc/build/snn_{api,ops,run,dc2015,dc2015_resident,twolayer,convlif,encode,mfma,dist}.o -ldl
#include "../bindsnet_amd/csrc/snn_dc2015_async.hip"
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {
constexpr int PB = 32, PNIN = 784, PSTEPS = 64;

__global__ void k_fill_trace(float *xtr, size_t n) {
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        xtr[k] = (k % 5 == 0) ? 0.f : 0.001f * (float)(k % 997);
}

// variant: see the header.  out[variant][rep] = 10 ns ticks of the block
__global__ __launch_bounds__(512) void k_probe_won(const float *xtr, const uint32_t *rowmask_g, const uint16_t *arows_g, int nact, int reps, int variant, long long *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CW = ACW, TT = AT, NBC = ANT - AT;
    const int B = PB, Nin = PNIN;
    float *wtile = (float *)smem, *wbak = wtile + Nin * CW, *wwin = wbak + Nin * CW;
    uint32_t *rowmask = (uint32_t *)(wwin + Nin * CW);
    uint16_t *arows = (uint16_t *)(rowmask + Nin);
    float *xn0 = (float *)(arows + 2 * ((Nin + 1) / 2));
    float *xfake = xn0 + TT;                                   // [Nin] variant E: the "trace" in LDS
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int k = tid; k < Nin * CW; k += 512) { wtile[k] = 0.2f + 1e-4f * (float)(k % 311); wbak[k] = wtile[k] - 1e-3f; wwin[k] = 0.f; }
    for (int k = tid; k < Nin; k += 512) { rowmask[k] = rowmask_g[k]; xfake[k] = 0.01f * (float)(k % 89); }
    for (int k = tid; k < nact; k += 512) arows[k] = arows_g[k];
    if (tid < TT) xn0[tid] = 1e-4f * (float)(tid % 17);
    PPar pp; pp.nu0 = vgpr(1e-4f); pp.nu1 = vgpr(1e-2f); pp.dt = vgpr(1.0f); pp.wmin = vgpr(0.f); pp.wmax = vgpr(1.f); pp.use_dt = 1; pp.has_min = 1; pp.has_max = 1;
    __syncthreads();
    const int q = 1;                                           // the crossing column
    for (int rep = 0; rep < reps; ++rep) {
        const int bst = (rep * 7) % B, step = (rep * 13) % PSTEPS;
        const uint32_t cm = 1u << bst;
        const float *xsrc = xtr + (size_t)step * B * Nin;
        const float xw = 3e-4f;
        __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(64);                      // ~3.4 us of nothing: the rest of an iteration
        __syncthreads();
        long long t0 = wall_clock64();
        if (variant == 0 || variant == 4) {
            if (wave >= 2)
                for (int i = tid - TT; i < Nin; i += NBC) {
                    const uint32_t m = rowmask[i];
                    if (m != 0) continue;
                    const float wold = wtile[i * CW + q];
                    const float xv = variant == 4 ? xfake[i] : xsrc[bst * Nin + i];
                    wwin[i * CW + q] = postpre_elem(pp, B, Nin, xn0, wold, i, q, m, cm, bst, xw, xsrc, true, xv);
                }
        } else if (variant == 1) {
            if (wave >= 2) {
                float pf[3]; bool ok[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = tid - TT + r * NBC;
                    ok[r] = i < Nin && rowmask[min(i, Nin - 1)] == 0;
                    pf[r] = ok[r] ? xsrc[bst * Nin + i] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = tid - TT + r * NBC;
                    if (!ok[r]) continue;
                    wwin[i * CW + q] = postpre_elem(pp, B, Nin, xn0, wtile[i * CW + q], i, q, 0u, cm, bst, xw, xsrc, true, pf[r]);
                }
            }
        } else if (variant == 5) {
            // untouched rows: every LDS / global read of the thread's three rows first, then the rows; no pre-synaptic term (no source spiked),
            // the post-synaptic cascade of one term written as 0 + the term
            if (wave >= 2) {
                uint32_t mm[3]; float w3[3], x3[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = tid - TT + r * NBC, ic = min(i, Nin - 1);
                    mm[r] = i < Nin ? rowmask[ic] : 1u; w3[r] = wtile[ic * CW + q]; x3[r] = xsrc[bst * Nin + ic];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int i = tid - TT + r * NBC;
                    if (mm[r] != 0) continue;
                    float w = w3[r];
                    if (pp.nu0 != 0.f) { float uu = 0.f; if (pp.use_dt) uu = uu * pp.dt; w = w - uu; }
                    if (pp.nu1 != 0.f) { float uu = 0.0f + x3[r] * (1.0f * pp.nu1); if (pp.use_dt) uu = uu * pp.dt; w = w + uu; }
                    if (pp.has_min && w < pp.wmin) w = pp.wmin;
                    if (pp.has_max && w > pp.wmax) w = pp.wmax;
                    wwin[i * CW + q] = w;
                }
            }
        } else {                                               // touched rows: 2 = product, 3 = value asked for ahead
            float pf = 0.f;
            int i = 0;
            if (tid < nact) { i = (int)arows[tid]; if (variant == 3) pf = xsrc[bst * Nin + i]; }
            if (variant == 3) { __builtin_amdgcn_s_sleep(40); t0 = wall_clock64(); }
            if (tid < nact) {
                const uint32_t m = rowmask[i];
                const float xv = variant == 3 ? pf : xsrc[bst * Nin + i];
                wwin[i * CW + q] = postpre_elem(pp, B, Nin, xn0, wbak[i * CW + q], i, q, m, cm, bst, xw, xsrc, true, xv);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long t1 = wall_clock64();
        __syncthreads();
        const long long t2 = wall_clock64();
        if (tid == ((variant == 2 || variant == 3) ? 0 : 128)) { out[(size_t)rep * 2] = t1 - t0; out[(size_t)rep * 2 + 1] = t2 - t0; }
    }
    if (tid == 0 && wwin[5] == 12345.f) out[0] = 0;              // (keeps the stores alive)
}
}  // namespace

int main() {
    const size_t ntr = (size_t)PSTEPS * PB * PNIN;
    float *xtr; CK(hipMalloc(&xtr, ntr * 4));
    std::vector<uint32_t> rm(PNIN, 0); std::vector<uint16_t> ar;
    srand(7);
    for (int i = 0; i < PNIN; ++i) if (rand() % 1000 < 314) { rm[i] = 1u << (rand() % PB); ar.push_back((uint16_t)i); }   // ~246 touched rows
    uint32_t *rmg; uint16_t *arg; long long *out;
    const int reps = 300;
    CK(hipMalloc(&rmg, PNIN * 4)); CK(hipMalloc(&arg, 2048)); CK(hipMalloc(&out, (size_t)reps * 16));
    CK(hipMemcpy(rmg, rm.data(), PNIN * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(arg, ar.data(), ar.size() * 2, hipMemcpyHostToDevice));
    const size_t lds = (size_t)3 * PNIN * ACW * 4 + PNIN * 4 + 2 * ((PNIN + 1) / 2) * 2 + AT * 4 + PNIN * 4 + 64;
    CK(hipFuncSetAttribute((const void *)k_probe_won, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char *names[6] = {"A untouched rows, product loop", "B untouched rows, three values asked for together", "C touched rows, product loop",
                            "D touched rows, value asked for ~1 us ahead", "E untouched rows, value from LDS (no global load)", "F untouched rows, all reads first, no cascade bookkeeping"};
    printf("{\"probe\": \"won_branch\", \"touched_rows\": %d, \"lds_bytes\": %zu}\n", (int)ar.size(), lds);
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 6; ++v) {
            hipLaunchKernelGGL(k_fill_trace, dim3(512), dim3(256), 0, 0, xtr, ntr);      // the trace as another kernel leaves it
            CK(hipMemset(out, 0, (size_t)reps * 16));
            hipLaunchKernelGGL(k_probe_won, dim3(1), dim3(512), lds, 0, xtr, rmg, arg, (int)ar.size(), reps, v, out);
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)reps * 2);
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            double a = 0, b = 0; int n = 0;
            for (int r = 20; r < reps; ++r, ++n) { a += (double)h[(size_t)r * 2]; b += (double)h[(size_t)r * 2 + 1]; }
            printf("{\"variant\": \"%s\", \"own_wave_us\": %.3f, \"all_waves_us\": %.3f}\n", names[v], a / n / 100.0, b / n / 100.0);
        }
    return 0;
}

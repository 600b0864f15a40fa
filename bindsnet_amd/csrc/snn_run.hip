// snn_run.hip -- Network.run() in C++: the per-timestep scheduler of
// bindsnet/network/network.py:380-465 driving the gfx950 kernels.
//
// Plan "generic": per-operator launches in exactly the reference's order (connections in
// insertion order feed `zeros + c1 + c2`, layers step in insertion order, then learning rules,
// monitors are written by the node kernels themselves, normalisation after the loop).
// Fused plans (below) replace the whole step by one launch for graphs they recognise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/snnhip.h"
#include "snn_common.hpp"

static thread_local const char *g_plan = "none";
static int g_plan_mode = 0;

// ---- profiling samples (bench.py roofline) ------------------------------------------------------
static int g_prof_stride = 0;
static constexpr int kMaxSamples = 256;
static hipEvent_t g_ev0[kMaxSamples], g_ev1[kMaxSamples];
static int g_nsamples = 0;
static bool g_ev_ready = false;

extern "C" void snn_profile_enable(int stride) { g_prof_stride = stride > 0 ? stride : 0; g_nsamples = 0; }

bool snn_prof_active() { return g_prof_stride > 0; }

bool snn_prof_begin(int t, hipStream_t st) {
    if (!g_prof_stride || t % g_prof_stride || g_nsamples >= kMaxSamples) return false;
    if (!g_ev_ready) {
        for (int i = 0; i < kMaxSamples; ++i) { (void)hipEventCreate(&g_ev0[i]); (void)hipEventCreate(&g_ev1[i]); }
        g_ev_ready = true;
    }
    (void)hipEventRecord(g_ev0[g_nsamples], st);
    return true;
}
void snn_prof_end(hipStream_t st) { (void)hipEventRecord(g_ev1[g_nsamples++], st); }

extern "C" int snn_profile_collect(double *sum_ms, int *n) {
    if (!sum_ms || !n) return SNN_ERR_INVALID;
    double acc = 0.0;
    for (int i = 0; i < g_nsamples; ++i) {
        float ms = 0.f;
        if (snn_check(hipEventElapsedTime(&ms, g_ev0[i], g_ev1[i]))) return SNN_ERR_LAUNCH;
        acc += ms;
    }
    *sum_ms = acc; *n = g_nsamples; g_nsamples = 0;
    return SNN_OK;
}

extern "C" const char *snn_plan_name(void) { return g_plan; }
void snn_set_plan_name(const char *name) { g_plan = name; }
extern "C" void snn_set_plan_mode(int mode) { g_plan_mode = mode; }

int snn_try_fused_dc2015(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                         hipStream_t st, int resident, int allow_lean, int *handled, unsigned *normalized);
int snn_try_fused_twolayer(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                           hipStream_t st, int *handled, unsigned *normalized);
int snn_try_fused_convlif(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                          hipStream_t st, int *handled);
int snn_try_fused_convpp(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, hipStream_t st, int *handled);

int snn_launch_dc_membrane(float *v, float *refrac, uint8_t *s, float *theta, const float *I, int B, int N,
                           const snn_dc_params &p, long long *cursor, float *raster_v, hipStream_t st);
int snn_launch_dc_arbitrate(uint8_t *s, float *x, int B, int N, const snn_dc_params &p, const float *Q, long long q_len,
                            long long *cursor, int *status, uint8_t *raster_s, hipStream_t st);
int snn_launch_rng_fill(snn_rng_state *rng, const uint8_t *s, int B, int N, float *qbuf, long long *cursor,
                        hipStream_t st);

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// ---- run(..., clamp / unclamp / injects_v / masks): small elementwise helpers of the generic plan (network.py:395-449)
__global__ __launch_bounds__(256) void k_clamp(uint8_t *__restrict__ s, uint8_t *__restrict__ raster, const uint8_t *__restrict__ clamp,
                                               const uint8_t *__restrict__ unclamp, long total, int n) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) {
        const int j = (int)(k % n);
        uint8_t v = s[k];
        if (clamp && clamp[j]) v = 1;                  // :416-421
        if (unclamp && unclamp[j]) v = 0;              // :424-429
        s[k] = v;
        if (raster) raster[k] = v;                     // monitors record after the clamps
    }
}
__global__ __launch_bounds__(256) void k_inject(float *__restrict__ v, const float *__restrict__ inj, long total, int len) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) v[k] = v[k] + inj[k % len];   // :399-404
}
__global__ __launch_bounds__(256) void k_add_current(float *__restrict__ cur, const float *__restrict__ ext, long total) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) cur[k] = cur[k] + ext[k];      // network.py:386-392
}
__global__ __launch_bounds__(256) void k_mask_fill(float *__restrict__ W, const uint8_t *__restrict__ mask, long total) {
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long)gridDim.x * 256) if (mask[k]) W[k] = 0.f;     // topology.py:129-133
}
static unsigned grid_for(long n) { const long g = (n + 255) / 256; return (unsigned)(g < 4096 ? (g > 0 ? g : 1) : 4096); }

static int validate(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R) {
    if (!L || nL <= 0 || (nC > 0 && !C) || !R || R->B <= 0 || R->T < 0) return SNN_ERR_INVALID;
    for (int l = 0; l < nL; ++l) {
        const snn_layer_desc &d = L[l];
        if (d.n <= 0) return SNN_ERR_INVALID;
        switch (d.kind) {
            case SNN_LAYER_INPUT:
                if (!d.ext_spikes || !d.s) return SNN_ERR_INVALID;
                if (d.p.lif.traces && !d.x) return SNN_ERR_INVALID;
                break;
            case SNN_LAYER_DC:
                if (!d.theta) return SNN_ERR_INVALID;
                if (d.p.one_spike && (!R->cursor || !R->status)) return SNN_ERR_INVALID;
                if (d.p.one_spike && !R->noise_q && !(R->rng && R->qbuf)) return SNN_ERR_INVALID;
                if (R->B > 1024) return SNN_ERR_UNSUPPORTED;
                if (d.thresh_vec) return SNN_ERR_UNSUPPORTED;          // per-neuron thresholds: LIF layers
                /* fallthrough */
            case SNN_LAYER_LIF:
                if (!d.v || !d.refrac || !d.s || !d.current) return SNN_ERR_INVALID;
                if (d.p.lif.traces && !d.x) return SNN_ERR_INVALID;
                break;
            default: return SNN_ERR_INVALID;
        }
    }
    for (int c = 0; c < nC; ++c) {
        const snn_conn_desc &d = C[c];
        if (d.src < 0 || d.src >= nL || d.dst < 0 || d.dst >= nL || !d.w) return SNN_ERR_INVALID;
        if (L[d.dst].kind == SNN_LAYER_INPUT) return SNN_ERR_UNSUPPORTED;
        const bool conv_mstdp = d.kind == SNN_CONN_CONV2D && d.rule == SNN_RULE_MSTDP;      // learning.py:1942-2015, batch 1
        if (d.kind == SNN_CONN_CONV2D && d.rule != SNN_RULE_NONE && !conv_mstdp && (d.rule != SNN_RULE_POSTPRE || !d.rule_ws)) return SNN_ERR_UNSUPPORTED;
        if (conv_mstdp && (!d.p_plus || !d.p_minus || !d.e_trace || d.reward_vec)) return SNN_ERR_INVALID;
        if (conv_mstdp && R->B != 1) return SNN_ERR_UNSUPPORTED;
        if (d.rule == SNN_RULE_POSTPRE && (!L[d.src].x || !L[d.dst].x)) return SNN_ERR_INVALID;
        if ((d.rule == SNN_RULE_MSTDP || d.rule == SNN_RULE_MSTDPET) && !conv_mstdp && (!d.p_plus || !d.p_minus || !d.s_src_prev || !d.s_tgt_prev)) return SNN_ERR_INVALID;
        if ((d.rule == SNN_RULE_HEBBIAN || d.rule == SNN_RULE_WDPOSTPRE) && (!L[d.src].x || !L[d.dst].x)) return SNN_ERR_INVALID;
        if (d.rule == SNN_RULE_MSTDPET && (!d.e_trace || R->B != 1)) return SNN_ERR_INVALID;
        if (d.rule < SNN_RULE_NONE || d.rule > SNN_RULE_MSTDPET) return SNN_ERR_INVALID;
        if (d.has_norm && (!d.norm_ws || d.kind == SNN_CONN_CONV2D)) return SNN_ERR_INVALID;
    }
    return SNN_OK;
}

// spikes of layer l as seen by connections BEFORE (after=false) / AFTER (after=true) its forward() at step t
static const uint8_t *layer_spikes(const snn_layer_desc &d, int B, int t, bool after) {
    if (d.kind != SNN_LAYER_INPUT) return d.s;
    const size_t stride = (size_t)B * d.n;
    if (after) return d.ext_spikes + (size_t)t * stride;
    return t == 0 ? d.s : d.ext_spikes + (size_t)(t - 1) * stride;
}

static int run_generic(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                       hipStream_t st) {
    const int B = R->B;
    bool fed[64];
    if (nL > 64) return SNN_ERR_UNSUPPORTED;
    for (int t = 0; t < R->T; ++t) {
        const bool prof = snn_prof_begin(t, st);
        // (1) network.py:384 _get_inputs(): previous-step spikes through every connection, in order.  one_step
        //     (network.py:388-393): the same per target layer, right before that layer steps, from the sources' CURRENT
        //     spikes -- a source that has already stepped in this timestep (lower layer index) contributes its new ones.
        for (int l = 0; l < nL; ++l) fed[l] = false;
        auto feed = [&](int only_dst) -> int {
            for (int c = 0; c < nC; ++c) {
                const snn_conn_desc &d = C[c];
                if (only_dst >= 0 && d.dst != only_dst) continue;
                const snn_layer_desc &S = L[d.src], &D = L[d.dst];
                const uint8_t *sp = layer_spikes(S, B, t, only_dst >= 0 && d.src < only_dst);
                const int acc = fed[d.dst] ? 1 : 0;
                if (d.kind == SNN_CONN_MCC) TRY(snn_prop_cascade_f32(d.w, sp, D.current, B, S.n, D.n, acc, st));
                else if (d.kind == SNN_CONN_DENSE) TRY(snn_prop_dense_f32(d.w, d.bias, sp, D.current, B, S.n, D.n, acc, st));
                else TRY(snn_prop_conv2d_f32(d.w, d.bias, sp, D.current, B, d.cin, d.h, d.wd, d.cout, d.kh, d.kw,
                                             d.stride, d.pad, acc, st));
                fed[d.dst] = true;
            }
            return SNN_OK;
        };
        if (!R->one_step) TRY(feed(-1));
        // (2) network.py:386-413 layers in insertion order
        for (int l = 0; l < nL; ++l) {
            const snn_layer_desc &d = L[l];
            if (R->one_step && d.kind != SNN_LAYER_INPUT) TRY(feed(l));
            const size_t off = (size_t)t * B * d.n;
            uint8_t *rs = d.raster_s ? d.raster_s + off : nullptr;
            float *rv = d.raster_v ? d.raster_v + off : nullptr;
            if (d.kind == SNN_LAYER_INPUT) {
                TRY(snn_input_step(d.ext_spikes + off, d.p.lif.traces ? d.x : nullptr, (long)B * d.n,
                                   d.p.lif.trace_decay, d.p.lif.trace_scale, d.p.lif.traces_additive, rs, st));
                continue;
            }
            if (!fed[l]) TRY(snn_check(hipMemsetAsync(d.current, 0, sizeof(float) * (size_t)B * d.n, st)));  // :409-413
            // an external current for this layer: added behind the connections' sums (:386-392).  With one_step the reference
            // OVERWRITES it for a layer that has an incoming connection: `current_inputs[l] = inputs[l][t]` is followed by
            // `current_inputs.update(self._get_inputs(layers=[l]))` (:388-393), which replaces the entry -- the current survives
            // only where no connection feeds the layer
            if (d.ext_current && !(R->one_step && fed[l]))
                hipLaunchKernelGGL(k_add_current, dim3(grid_for((long)B * d.n)), dim3(256), 0, st, d.current, d.ext_current + off, (long)B * d.n);
            if (d.inject_v) {
                const int len = d.inject_len > 0 ? d.inject_len : d.n;
                hipLaunchKernelGGL(k_inject, dim3(grid_for((long)B * d.n)), dim3(256), 0, st, d.v,
                                   d.inject_v + (d.inject_per_step ? (size_t)t * len : 0), (long)B * d.n, len);
            }
            if (d.kind == SNN_LAYER_LIF) TRY(snn_lif_step_vth(d.v, d.refrac, d.s, d.x, d.current, B, d.n, &d.p.lif, d.thresh_vec, rs, rv, st));
            else if (R->rng && d.p.one_spike) {   // device generator: membrane -> draws for this step -> arbitration
                TRY(snn_launch_dc_membrane(d.v, d.refrac, d.s, d.theta, d.current, B, d.n, d.p, R->cursor, rv, st));
                TRY(snn_launch_rng_fill(R->rng, d.s, B, d.n, R->qbuf, R->cursor, st));
                TRY(snn_launch_dc_arbitrate(d.s, d.x, B, d.n, d.p, R->qbuf, (long long)B * d.n, R->cursor, R->status,
                                            rs, st));
            } else TRY(snn_dc_step(d.v, d.refrac, d.s, d.x, d.theta, d.current, B, d.n, &d.p, R->noise_q, R->q_len,
                                   R->cursor, R->status, rs, rv, st));
            // clamp / unclamp right behind the layer's own step (network.py:394-429): with one_step a later layer's
            // currents are taken from these spikes
            if (d.clamp || d.unclamp)
                hipLaunchKernelGGL(k_clamp, dim3(grid_for((long)B * d.n)), dim3(256), 0, st, d.s, rs,
                                   d.clamp ? d.clamp + (d.clamp_per_step ? (size_t)t * d.n : 0) : nullptr,
                                   d.unclamp ? d.unclamp + (d.unclamp_per_step ? (size_t)t * d.n : 0) : nullptr, (long)B * d.n, d.n);
        }
        // (3) network.py:431-454 learning rules, connection order
        if (R->learning)
            for (int c = 0; c < nC; ++c) {
                const snn_conn_desc &d = C[c];
                if (d.rule == SNN_RULE_NONE) continue;
                const snn_layer_desc &S = L[d.src], &D = L[d.dst];
                const uint8_t *ss = layer_spikes(S, B, t, true);
                if (d.rule == SNN_RULE_MSTDP && d.kind == SNN_CONN_CONV2D)
                    TRY(snn_conv2d_mstdp_step(d.w, d.e_trace, d.p_plus, d.p_minus, ss, D.s, d.cin, d.h, d.wd, d.cout, d.kh, d.kw, d.stride,
                                              d.pad, d.reward, d.nu0, d.a_plus, d.a_minus, d.decay_plus, d.decay_minus, d.wdecay,
                                              d.has_min, d.wmin, d.has_max, d.wmax, st));
                else if (d.rule == SNN_RULE_POSTPRE && d.kind == SNN_CONN_CONV2D)
                    TRY(snn_conv2d_postpre(d.w, ss, S.x, D.s, D.x, B, d.cin, d.h, d.wd, d.cout, d.kh, d.kw, d.stride, d.pad, d.nu0, d.nu1,
                                           d.wdecay, d.has_min, d.wmin, d.has_max, d.wmax, d.rule_ws, st));
                else if (d.rule == SNN_RULE_POSTPRE)
                    TRY(snn_stdp_postpre(d.w, ss, S.x, D.s, D.x, B, S.n, D.n, d.nu0, d.nu1, d.use_dt, R->dt, d.wdecay,
                                         d.has_min, d.wmin, d.has_max, d.wmax, /*assume_clamped=*/t > 0, st));
                else if (d.rule == SNN_RULE_HEBBIAN || d.rule == SNN_RULE_WDPOSTPRE)
                    TRY(snn_stdp_hebbian(d.w, ss, S.x, D.s, D.x, B, S.n, D.n, d.nu0, d.nu1, d.rule == SNN_RULE_WDPOSTPRE, d.wdecay,
                                         d.has_min, d.wmin, d.has_max, d.wmax, st));
                else if (d.rule == SNN_RULE_MSTDPET)
                    TRY(snn_mstdpet_step(d.w, d.e_trace, d.p_plus, d.p_minus, d.s_src_prev, d.s_tgt_prev, ss, D.s, S.n, D.n, d.reward,
                                         d.nu0, R->dt, d.a_plus, d.a_minus, d.decay_plus, d.decay_minus, d.decay_e, d.tc_e, d.wdecay,
                                         d.has_min, d.wmin, d.has_max, d.wmax, st));
                else
                    TRY(snn_mstdp_step(d.w, d.p_plus, d.p_minus, d.s_src_prev, d.s_tgt_prev, ss, D.s, B, S.n, D.n,
                                       d.reward, d.reward_vec, d.nu0, d.a_plus, d.a_minus, d.decay_plus, d.decay_minus,
                                       d.wdecay, d.has_min, d.wmin, d.has_max, d.wmax, st));
            }
        for (int c = 0; c < nC; ++c)          // masks apply every step, learning or not (topology.py:124-133)
            if (C[c].mask && C[c].kind != SNN_CONN_CONV2D)
                hipLaunchKernelGGL(k_mask_fill, dim3(grid_for((long)L[C[c].src].n * L[C[c].dst].n)), dim3(256), 0, st, C[c].w, C[c].mask,
                                   (long)L[C[c].src].n * L[C[c].dst].n);
        for (int c = 0; c < nC; ++c)          // network.py:456-458: monitors record last, i.e. the weights this step leaves behind
            if (C[c].raster_w) {
                const size_t ne = C[c].kind == SNN_CONN_CONV2D ? (size_t)C[c].cout * C[c].cin * C[c].kh * C[c].kw
                                                               : (size_t)L[C[c].src].n * L[C[c].dst].n;
                if (hipMemcpyAsync(C[c].raster_w + (size_t)t * ne, C[c].w, ne * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return SNN_ERR_LAUNCH;
            }
        if (prof) snn_prof_end(st);
    }
    return snn_check_launch();
}

// Spike monitors on Input layers (snn_layer_desc.raster_s of an INPUT layer): the raster of an Input layer is a copy of its input
// (monitors.py:94-111 clone `s`, which aliases the input slice), so no plan writes it step by step -- the plans see the descriptors
// WITHOUT it, and snn_net_run makes ONE device copy per monitored input behind the plan; a plan that has the copy made inside its own
// launch (third-generation D&C form: its producer workgroups, snn_dc2015_async.hip) takes the request from snn_input_raster_request()
// and says so with snn_input_raster_done().
static thread_local uint8_t *g_in_raster_req[8];
static thread_local unsigned g_in_raster_done;
uint8_t *snn_input_raster_request(int layer) { return layer >= 0 && layer < 8 ? g_in_raster_req[layer] : nullptr; }
void snn_input_raster_done(int layer) { if (layer >= 0 && layer < 8) g_in_raster_done |= 1u << layer; }

static int net_run_plans(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, hipStream_t st);
void snn_dc2015_ws_key_in(unsigned long long key);   // snn_dc2015.hip: the caller's host_state[0] as this run found it

extern "C" int snn_net_run(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                           snn_stream_t stream) {
    TRY(validate(L, nL, C, nC, R));
    hipStream_t st = (hipStream_t)stream;
    bool any = false;
    for (int l = 0; l < nL; ++l) any = any || (L[l].kind == SNN_LAYER_INPUT && L[l].raster_s);
    if (!any) return net_run_plans(L, nL, C, nC, R, st);
    std::vector<snn_layer_desc> L2(L, L + nL);
    g_in_raster_done = 0;
    for (int l = 0; l < nL; ++l) {
        if (l < 8) g_in_raster_req[l] = nullptr;
        if (L[l].kind != SNN_LAYER_INPUT || !L[l].raster_s) continue;
        L2[l].raster_s = nullptr;
        if (l < 8) g_in_raster_req[l] = L[l].raster_s;
    }
    const int rc = net_run_plans(L2.data(), nL, C, nC, R, st);
    for (int l = 0; l < nL && l < 8; ++l) g_in_raster_req[l] = nullptr;
    if (rc) return rc;
    for (int l = 0; l < nL; ++l)
        if (L[l].kind == SNN_LAYER_INPUT && L[l].raster_s && !(l < 8 && ((g_in_raster_done >> l) & 1u)))
            TRY(snn_check(hipMemcpyAsync(L[l].raster_s, L[l].ext_spikes, (size_t)R->T * R->B * L[l].n, hipMemcpyDeviceToDevice, st)));
    return SNN_OK;
}

static int net_run_plans(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, hipStream_t st) {
    // What a pipelined caller's earlier runs knew about the workspace's content (snn_run_desc.host_state[0]: "the exchange areas are clean
    // for this layout") holds only from one chained lean D&C run to the next: EVERY run takes the word away first, and only that path puts
    // it back (snn_dc2015.hip) -- a plan that scribbles over the workspace can never leave a stale "clean" behind.
    const unsigned long long ws_key = R->host_state ? R->host_state[0] : 0ull;
    if (R->host_state) R->host_state[0] = 0ull;
    snn_dc2015_ws_key_in(ws_key);
    int handled = 0;
    unsigned normalized = 0;       // bit c: connection c was already normalised by the plan's own kernel
    int mode = g_plan_mode ? g_plan_mode : R->plan;            // the process-wide test switch wins over the per-run request
    for (int l = 0; l < nL; ++l) if (L[l].clamp || L[l].unclamp || L[l].inject_v || L[l].ext_current) mode = 1;   // only the generic plan implements these
    // (a Conv2dConnection with a rule: the generic plan, except PostPre on the Input -> Conv2d -> LIF graph -- snn_try_fused_convpp, whole-run form only)
    bool conv_rule = false;
    for (int c = 0; c < nC; ++c) if (C[c].kind == SNN_CONN_CONV2D && C[c].rule != SNN_RULE_NONE) conv_rule = true;
    for (int c = 0; c < nC; ++c) if (C[c].mask || C[c].raster_w) mode = 1;
    if (R->one_step) mode = 1;
    for (int l = 0; l < nL; ++l) if (L[l].thresh_vec) mode = 1;      // per-neuron thresholds: generic plan
    if (conv_rule) {
        if (mode == 0 || mode == 3) TRY(snn_try_fused_convpp(L, nL, C, nC, R, st, &handled));
        if (!handled) mode = 1;
    }
    if (mode != 1 && !handled) TRY(snn_try_fused_dc2015(L, nL, C, nC, R, st, mode == 0 || mode == 3, mode == 0, &handled, &normalized));
    if (mode != 1 && !handled) TRY(snn_try_fused_twolayer(L, nL, C, nC, R, st, &handled, &normalized));
    if (mode != 1 && !handled) TRY(snn_try_fused_convlif(L, nL, C, nC, R, st, &handled));
    if (!handled) {
        g_plan = "generic";
        TRY(run_generic(L, nL, C, nC, R, st));
    }
    // network.py:464-465: normalise every connection after the loop (learning or not)
    for (int c = 0; c < nC; ++c)
        if (C[c].has_norm && !((normalized >> c) & 1u))
            TRY(snn_normalize(C[c].w, L[C[c].src].n, L[C[c].dst].n, C[c].norm, C[c].norm_abs, C[c].norm_ws, st));
    return SNN_OK;
}


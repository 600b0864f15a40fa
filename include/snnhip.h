/*
 * snnhip.h -- C ABI of libsnnhip.so: the MI355X (gfx950) implementation of BindsNET's
 * per-timestep Network.run() hot path.
 *
 * BindsNET has no FFI / operator-plugin interface of its own (it is pure Python on PyTorch),
 * so the entry points below are what a binding of that path needs: one per reference function
 * of SURVEY.md section 8(a), plus the multi-step driver snn_net_run.  Each declaration cites the
 * reference function it replaces (paths relative to the BindsNET repository root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every pointer is a DEVICE pointer unless the
 *     parameter name starts with `h_`.  The caller owns every buffer, scratch included;
 *     the library allocates no device memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All calls are
 *     asynchronous and stream-ordered; none synchronises the device.
 *   - return value: 0 = SNN_OK, negative = error (snn_error_string()).  No C++ exception ever
 *     crosses the boundary.
 *   - arithmetic: IEEE binary32, round-to-nearest-even, no FMA contraction; every reduction
 *     follows the order the reference executes on CPU (ATen cascade sum: SURVEY.md Appendix A;
 *     serial order, see DESIGN.md "Summation order").  Spikes are uint8 (0/1), row-major
 *     [B, n]; weights are float32 row-major [Nin, N] (source-major, like BindsNET's `w`).
 */
#ifndef SNNHIP_H
#define SNNHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNN_ABI_VERSION 8

typedef void *snn_stream_t;

enum {
    SNN_OK = 0,
    SNN_ERR_INVALID = -1,      /* bad argument (null pointer, non-positive size, ...) */
    SNN_ERR_UNSUPPORTED = -2,  /* size outside what the kernels were built for */
    SNN_ERR_LAUNCH = -3,       /* hip launch / runtime error (see snn_last_hip_error) */
    SNN_ERR_NOISE = -4,        /* one_spike noise stream exhausted (device status word) */
    SNN_ERR_NO_DEVICE = -5,
    SNN_ERR_TIMEOUT = -6,      /* an in-kernel workgroup hand-off gave up waiting (device status word) */
    SNN_ERR_RETRY = -7         /* the lean form of a plan met a step it does not handle (device status word): no state
                                  was touched; run the same input again with snn_run_desc.plan = 3 */
};

int snn_abi_version(void);
const char *snn_error_string(int code);
/* hipGetErrorString() of the last failing HIP call made by this library on this thread. */
const char *snn_last_hip_error(void);
/* Number of visible HIP devices, or SNN_ERR_NO_DEVICE. */
int snn_device_count(void);

/* ---- a5: MulticompartmentConnection.compute + Weight.compute -------------------------------
 * bindsnet/network/topology.py:437-479, bindsnet/network/topology_features.py:633-645
 * out[b,j] (+)= sum_i W[i,j] * s[b,i] in ATen sum(dim=1) order (cascade over i for columns
 * j < 32*floor(N/32), 4-lane row_sum for the rest).  accumulate=0: out = 0 + r;
 * accumulate=1: out = out + r (network.py:240-248, connection insertion order).
 * Limits: Nin <= 2^19.                                                                     */
int snn_prop_cascade_f32(const float *W, const uint8_t *s, float *out,
                         int B, int Nin, int N, int accumulate, snn_stream_t stream);

/* ---- a6: Connection.compute ---------------------------------------------------------------
 * bindsnet/network/topology.py:332-346.  out[b,j] (+)= sum_i s[b,i]*W[i,j] (+ bias[j]),
 * canonical ascending-i sequential f32 (the reference's MKL order is not reproducible,
 * SURVEY.md finding 5).  bias may be NULL.                                                  */
int snn_prop_dense_f32(const float *W, const float *bias, const uint8_t *s, float *out,
                       int B, int Nin, int N, int accumulate, snn_stream_t stream);

/* The same product on the f32 matrix cores (v_mfma_f32_16x16x4_f32): ONE k-ordered accumulator chain per output
 * tile, bit-identical to snn_prop_dense_f32 when every spike byte is 0 or 1 (products are then exact and gfx950's
 * f32 MFMA is a k-ordered fmaf chain).  Cost is Nin / 4 dependent MFMAs per tile regardless of sparsity; kept as an
 * operator for dense inputs and as the measured alternative to the event-driven kernel (profiles/).   */
int snn_prop_dense_mfma_f32(const float *W, const float *bias, const uint8_t *s, float *out,
                            int B, int Nin, int N, int accumulate, snn_stream_t stream);

/* ---- a7: Conv2dConnection.compute ---------------------------------------------------------
 * bindsnet/network/topology.py:799-815 (F.conv2d).  s [B,Cin,H,W] u8, W [Cout,Cin,KH,KW],
 * out [B,Cout,OH,OW]; accumulated sequentially in (kh,kw,cin) order -- taps row-major, input channels
 * innermost: what the reference's oneDNN kernel does for Cin <= 16 (bit-exact against reference fixtures for Cin =
 * 1, 3, 8, 16) --, then bias.  Cin > 16: SNN_ERR_UNSUPPORTED (oneDNN switches kernels, order not characterised). */
int snn_prop_conv2d_f32(const float *W, const float *bias, const uint8_t *s, float *out,
                        int B, int Cin, int H, int Wd, int Cout, int KH, int KW,
                        int stride, int pad, int accumulate, snn_stream_t stream);

/* ---- a2: Input.forward + Nodes.forward trace ------------------------------------------------
 * bindsnet/network/nodes.py:211-221, :96-107.  s is the caller's input slice (aliased, never
 * copied); x (nullable) is the trace, updated in place; raster_out (nullable) receives s.   */
int snn_input_step(const uint8_t *s, float *x, long n_total, float trace_decay,
                   float trace_scale, int additive, uint8_t *raster_out, snn_stream_t stream);

typedef struct {
    float decay, rest, reset, thresh, refrac, dt;
    int has_lbound; float lbound;
    int traces; float trace_decay, trace_scale; int traces_additive;
} snn_lif_params;

/* ---- a3: LIFNodes.forward -------------------------------------------------------------------
 * bindsnet/network/nodes.py:500-529.  v, refrac [B,N] f32 in/out; s [B,N] u8 out; x nullable
 * trace; I [B,N] input current, masked in place where refractory (nodes.py:511);
 * raster_s / raster_v nullable per-step monitor slices.                                      */
int snn_lif_step(float *v, float *refrac, uint8_t *s, float *x, float *I, int B, int N,
                 const snn_lif_params *h_p, uint8_t *raster_s, float *raster_v,
                 snn_stream_t stream);
/* The same step with PER-NEURON thresholds (nodes.py:425-498 take `thresh` as a tensor; examples/mnist/reservoir.py builds its
 * LIF layer that way): thresh_vec [N] f32 on the device replaces h_p->thresh, neuron j of every sample compares against
 * thresh_vec[j].  thresh_vec == NULL: snn_lif_step.  (ABI 8)                                                              */
int snn_lif_step_vth(float *v, float *refrac, uint8_t *s, float *x, float *I, int B, int N,
                     const snn_lif_params *h_p, const float *thresh_vec, uint8_t *raster_s, float *raster_v,
                     snn_stream_t stream);

typedef struct {
    snn_lif_params lif;
    float theta_decay, theta_plus;
    int learning;      /* nodes.py:1078,1093: theta decays / grows only while learning */
    int one_spike;     /* nodes.py:1097 */
} snn_dc_params;

/* ---- a4: DiehlAndCookNodes.forward ----------------------------------------------------------
 * bindsnet/network/nodes.py:1069-1111.  theta [N] shared by the batch.  one_spike winner
 * selection reproduces torch.multinomial on the CPU generator: noise_q is the pre-drawn
 * Exp(1) stream (torch.empty(K).exponential_(1)), *cursor (device int64) the number of
 * draws consumed so far; a step with r rows that crossed threshold consumes r*N draws
 * (SURVEY.md Appendix B).  *status (device int32) is set to SNN_ERR_NOISE, and the step
 * leaves s as the un-arbitrated crossings, if fewer than r*N draws remain.                   */
int snn_dc_step(float *v, float *refrac, uint8_t *s, float *x, float *theta, const float *I,
                int B, int N, const snn_dc_params *h_p,
                const float *noise_q, long long q_len, long long *cursor, int *status,
                uint8_t *raster_s, float *raster_v, snn_stream_t stream);

/* The second half of snn_dc_step on its own: bindsnet/network/nodes.py:1097-1111 -- one_spike winner selection on the
 * crossings `s` [B,N] (in: every neuron that crossed its threshold; out: one winner per row that had a crossing), then the
 * trace update with the final spikes (nodes.py:96-107) and the raster slice.  For callers that put something between
 * the membrane update and the arbitration: the exact batch-sharded multi-GPU mode runs the membrane half
 * (snn_dc_step with one_spike = 0, traces = 0, learning = 0) on each rank's rows, gathers the crossings of all ranks, and
 * then arbitrates the GLOBAL batch on every rank, so that the draws are consumed in the global row order.  noise_q /
 * q_len as in snn_dc_step; cursor[1] must hold the offset of this step's first draw inside noise_q
 * (snn_rng_fill_exponential leaves 0 there and the draws in qbuf); cursor[0] receives cursor[1] + rows_with_a_crossing * N.
 * h_p->learning and theta are not used here (the adaptive threshold belongs to the membrane half).           */
int snn_dc_arbitrate(uint8_t *s, float *x, int B, int N, const snn_dc_params *h_p,
                     const float *noise_q, long long q_len, long long *cursor, int *status,
                     uint8_t *raster_s, snn_stream_t stream);

/* ---- device-resident emulation of torch's CPU generator --------------------------------------
 * Replaces the pre-drawn noise_q stream: the library reproduces the draws torch.multinomial
 * (bindsnet/network/nodes.py:1100-1102) would consume -- mt19937 -> random64 -> u in [0,1) ->
 * (float)(-log1p(-u)) -- on the device, bit-exactly, from the generator state the host uploads
 * (parsed from torch.get_rng_state()).  pos: index of the next output inside the current 624-word
 * block, 624 = "twist before the next output" (at::mt19937's left_ == 1).  consumed counts draws.
 * After the run the host downloads the struct and writes it back with torch.set_rng_state(). */
typedef struct {
    uint32_t mt[624];
    int32_t pos;
    int32_t reserved;
    long long consumed;
} snn_rng_state;

/* For the rows of `crossings` [B,N] that contain a non-zero entry (r of them, in row order) write
 * the draws of the [r, N] operand's candidate positions to qbuf[rank*N + j], advance *rng by r*N
 * draws and zero cursor[1], so that snn_dc_step-style arbitration can index qbuf from 0.      */
int snn_rng_fill_exponential(snn_rng_state *rng, const uint8_t *crossings, int B, int N, float *qbuf,
                             long long *cursor, snn_stream_t stream);

/* ---- a8 / a9: PostPre -----------------------------------------------------------------------
 * MCC: bindsnet/learning/MCC_learning.py:224-302 + :86-110 (use_dt = 1: each reduced update
 * is multiplied by connection.dt).  Dense: bindsnet/learning/learning.py:390-420 + :87-104
 * (use_dt = 0).  W -= sum_b s_src (x) (x_tgt*nu0); W += sum_b x_src (x) (s_tgt*nu1);
 * W *= decay; clamp.  Batch sums in ATen sum(dim=0) order.  nu0 == 0 / nu1 == 0 skip that
 * half like the reference.  assume_clamped = 1: the caller guarantees decay == 1 and W already
 * inside [wmin, wmax], so elements with no pre- and no post-synaptic spike are skipped.
 * Limits: B <= 256.                                                                          */
int snn_stdp_postpre(float *W, const uint8_t *s_src, const float *x_src,
                     const uint8_t *s_tgt, const float *x_tgt, int B, int Nin, int N,
                     float nu0, float nu1, int use_dt, float dt, float decay,
                     int has_min, float wmin, int has_max, float wmax, int assume_clamped,
                     snn_stream_t stream);

/* ---- a10: MSTDP -----------------------------------------------------------------------------
 * bindsnet/learning/learning.py:1504-1574.  The reference's dense eligibility [B,Nin,N] is kept
 * FACTORED: elig[b] = p_plus[b] (x) s_tgt_prev[b] + s_src_prev[b] (x) p_minus[b], where p_plus /
 * p_minus are the values left by the previous call and *_prev the spikes of the previous call
 * (in/out, zero before the first call == the reference's zero-initialised eligibility).
 * Order: W += nu0 * sum_b reward*elig[b]  (batch sum in ATen sum(dim=0) order);
 *        p_plus = p_plus*decay_plus + a_plus*s_src;  p_minus = p_minus*decay_minus + a_minus*s_tgt;
 *        *_prev = current spikes;  W *= wdecay;  clamp.
 * reward_vec (nullable, device [B]) overrides the scalar reward.  Limits: B <= 256.          */
int snn_mstdp_step(float *W, float *p_plus, float *p_minus,
                   uint8_t *s_src_prev, uint8_t *s_tgt_prev,
                   const uint8_t *s_src, const uint8_t *s_tgt, int B, int Nin, int N,
                   float reward, const float *reward_vec, float nu0, float a_plus, float a_minus,
                   float decay_plus, float decay_minus, float wdecay,
                   int has_min, float wmin, int has_max, float wmax, snn_stream_t stream);

/* ---- f3: Hebbian / WeightDependentPostPre (dense Connection) --------------------------------------
 * bindsnet/learning/learning.py:1110-1135 and :626-653 (+ LearningRule.update :87-104).  U1 = sum_b s_src (x) x_tgt,
 * U2 = sum_b x_src (x) s_tgt (ATen batch-sum order), then
 *   weight_dependent = 0:  W += nu0 * U1;  W += nu1 * U2
 *   weight_dependent = 1:  update = 0 - (nu0 U1)(W - wmin) + (nu1 U2)(wmax - W);  W += update   (needs both bounds)
 * followed by W *= decay and the clamp.  Limits: B <= 256.                                              */
int snn_stdp_hebbian(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                     int B, int Nin, int N, float nu0, float nu1, int weight_dependent, float decay,
                     int has_min, float wmin, int has_max, float wmax, snn_stream_t stream);

/* ---- f4: PostPre on a Conv2dConnection -----------------------------------------------------------------
 * bindsnet/learning/learning.py:457-497 (+ :87-104).  With k = (cin,kh,kw), l = (oy,ox) and unfold = im2col:
 *   W[co,k] -= nu0 * sum_b sum_l x_tgt[b,co,l] * unfold(s_src)[b,k,l];  W[co,k] += nu1 * sum_b sum_l s_tgt[b,co,l] *
 *   unfold(x_src)[b,k,l];  W *= decay; clamp.  The sum over l is sequential in ascending l (the reference's runs inside
 * torch.bmm: BLAS order, compared within tolerance), the batch sum in ATen's sum(dim=0) order.
 * ws: device scratch of 2 * B * Cout*Cin*KH*KW floats.                                                       */
int snn_conv2d_postpre(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                       int B, int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad, float nu0, float nu1,
                       float decay, int has_min, float wmin, int has_max, float wmax, float *ws, snn_stream_t stream);

/* ---- f4: MSTDP on a Conv2dConnection (batch 1) -----------------------------------------------------------
 * bindsnet/learning/learning.py:1942-2015 (+ :87-104), defined at batch size 1 only (the reference views its [B, Cout, K]
 * eligibility as the weight's shape, :2013).  elig [Cout, K = Cin*KH*KW] is the rule's eligibility (in/out), p_plus the
 * P^+ trace in input space [Cin, H, W] (the reference's unfolded copy carries the same values), p_minus P^- [Cout, OH*OW].
 * Order:  W[co,k] += nu0 * sum_co' (reward * elig[co',k])  (the reference's torch.sum(update, dim=0) over a weight-shaped
 * eligibility: summed over the OUTPUT CHANNELS in ATen's order and broadcast back, :1966-1967);  W *= wdecay; clamp;
 * p_plus = p_plus * decay_plus + a_plus * s_src;  p_minus likewise with s_tgt;
 * elig[co,k] = sum_l s_tgt[co,l] * unfold(p_plus)[k,l] + sum_l p_minus[co,l] * unfold(s_src)[k,l]  (each ascending in l; the
 * reference's run inside torch.bmm: BLAS order, compared within tolerance).                                       */
int snn_conv2d_mstdp_step(float *W, float *elig, float *p_plus, float *p_minus, const uint8_t *s_src, const uint8_t *s_tgt,
                          int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad, float reward, float nu0,
                          float a_plus, float a_minus, float decay_plus, float decay_minus, float wdecay, int has_min,
                          float wmin, int has_max, float wmax, snn_stream_t stream);

/* ---- f3: MSTDPET (dense Connection, batch 1) -----------------------------------------------------------
 * bindsnet/learning/learning.py:2187-2248.  e_trace [Nin,N] is the rule's dense eligibility trace (in/out); the point
 * eligibility is p_plus (x) s_tgt_prev + s_src_prev (x) p_minus of the previous call's factors, formed on the fly.
 * Order: e_trace = e_trace * decay_e + elig / tc_e;  W += ((nu0 * dt) * reward) * e_trace;  W *= wdecay; clamp;
 * then p_plus / p_minus / *_prev advance as in snn_mstdp_step.                                           */
int snn_mstdpet_step(float *W, float *e_trace, float *p_plus, float *p_minus, uint8_t *s_src_prev, uint8_t *s_tgt_prev,
                     const uint8_t *s_src, const uint8_t *s_tgt, int Nin, int N, float reward, float nu0, float dt,
                     float a_plus, float a_minus, float decay_plus, float decay_minus, float decay_e, float tc_e,
                     float wdecay, int has_min, float wmin, int has_max, float wmax, snn_stream_t stream);

/* ---- a11: normalize -------------------------------------------------------------------------
 * AbstractFeature.normalize, bindsnet/network/topology_features.py:250-266 (use_abs = 0) and
 * Connection.normalize, bindsnet/network/topology.py:383-392 (use_abs = 1).
 * colsum in ATen sum(dim=0) order, zero -> 1, W *= norm * (1/colsum).
 * colsum_ws: device scratch of N floats.                                                      */
int snn_normalize(float *W, int Nin, int N, float norm, int use_abs, float *colsum_ws,
                  snn_stream_t stream);

/* Conv2dConnection.normalize, bindsnet/network/topology.py:824-837: W viewed as [n_filters = Cout*Cin, taps = KH*KW]; every filter is
 * scaled to sum `norm` -- w[f] *= norm * (1 / sum_k w[f,k]), the sum in ATen's vectorised inner-sum order (8 interleaved lanes of
 * row_sum, leftovers, then the lanes), no zero guard like the reference.  (ABI 6)                                  */
int snn_normalize_conv2d(float *W, int n_filters, int taps, float norm, snn_stream_t stream);

/* ---- f2: spike encoders on the device ---------------------------------------------------------
 * bindsnet/encoding/encodings.py:51-98 (bernoulli): out [steps, n] u8 = what torch.bernoulli(max_prob *
 * datum.repeat([steps, 1])) draws from the HOST generator whose state is in *rng -- one 32-bit mt19937 output
 * per element, u = (r & 0xFFFFFF) * 2^-24 < p -- bit for bit; *rng is advanced by steps * n outputs.
 * bindsnet/encoding/encodings.py:101-152 (poisson): same construction (intervals ~ Poisson(1000 / (x dt)), zeros
 * bumped to one, cumulated) from a Philox-4x32-10 stream keyed by (seed, element): same distribution, NOT the reference's
 * stream (ATen's sampler draws a data-dependent number of outputs per element).  The stream is SPECIFIED operation by
 * operation (csrc/snn_encode.hip: IEEE f32 / f64 adds, multiplies, divides, f32 sqrt, integer conversions; exp / log /
 * log k! are fixed series, no libm) and restated in oracle/snn_oracle.c (orc_encode_poisson), bit for bit.  out is
 * written completely (zeroed, then the spikes).                                                        */
int snn_encode_bernoulli(snn_rng_state *rng, const float *datum, int n, int steps, float max_prob, uint8_t *out,
                         snn_stream_t stream);
int snn_encode_poisson(const float *datum, int n, int steps, float dt, unsigned long long seed, uint8_t *out,
                       snn_stream_t stream);

/* ---- Network.reset_state_variables ----------------------------------------------------------
 * bindsnet/network/network.py:467-481 (-> nodes.py:109-120, :531-538, :1113-1120): spikes, traces and
 * refractory counters <- 0, voltages <- rest.  One launch fills up to SNN_MAX_FILL_SEGMENTS device buffers:
 * `pattern` is the 32-bit word every aligned word of the buffer receives (0, or the bit pattern of the f32
 * rest potential; a non-zero pattern needs a 4-byte aligned buffer of a multiple of 4 bytes).         */
#define SNN_MAX_FILL_SEGMENTS 32
typedef struct { void *ptr; unsigned long long bytes; uint32_t pattern; } snn_fill_segment;
int snn_fill_segments(const snn_fill_segment *h_segs, int n, snn_stream_t stream);

/* ---- a1: Network.run ------------------------------------------------------------------------
 * bindsnet/network/network.py:380-465 (the per-timestep loop and the post-loop normalisation),
 * for graphs built from {Input, LIFNodes, DiehlAndCookNodes} x {MulticompartmentConnection+
 * Weight, Connection, Conv2dConnection} x {no rule, PostPre, MSTDP}.  The descriptors are HOST
 * structs holding DEVICE pointers; layers and connections are listed in network insertion
 * order, which fixes the evaluation order exactly as the reference's dict iteration does.   */
enum { SNN_LAYER_INPUT = 0, SNN_LAYER_LIF = 1, SNN_LAYER_DC = 2 };
enum { SNN_CONN_MCC = 0, SNN_CONN_DENSE = 1, SNN_CONN_CONV2D = 2 };
enum { SNN_RULE_NONE = 0, SNN_RULE_POSTPRE = 1, SNN_RULE_MSTDP = 2, SNN_RULE_HEBBIAN = 3, SNN_RULE_WDPOSTPRE = 4,
       SNN_RULE_MSTDPET = 5 };

typedef struct {
    int kind;                   /* SNN_LAYER_* */
    int n;                      /* neurons per sample */
    snn_dc_params p;            /* LIF uses p.lif only; INPUT uses p.lif.traces/trace_* only */
    float *v, *refrac, *x, *theta;   /* state [B,n] ([n] for theta); NULL where the layer has none */
    uint8_t *s;                 /* [B,n] spikes at entry, updated in place (INPUT: entry value, read only) */
    const uint8_t *ext_spikes;  /* INPUT: the [T,B,n] input tensor (aliased per step, never copied) */
    uint8_t *raster_s;          /* nullable [T,B,n] spike monitor */
    float *raster_v;            /* nullable [T,B,n] voltage monitor */
    float *current;             /* [B,n] scratch for the summed input current (non-INPUT layers) */
    /* run(..., clamp= / unclamp= / injects_v=), network.py:395-429 (nullable; handled by the generic plan):
     * after the layer's step  s[:, clamp] = 1, then s[:, unclamp] = 0  (u8 masks [n], or [T,n] when *_per_step);
     * before it               v += inject_v  (f32 [n] broadcast over the batch, or [T,...] when inject_per_step, each
     *                         slice [inject_len] with inject_len = n or B*n) */
    const uint8_t *clamp, *unclamp; int clamp_per_step, unclamp_per_step;
    const float *inject_v; int inject_per_step; int inject_len;
    /* run(inputs={<non-Input layer>: current}), network.py:386-392 (nullable; generic plan): f32 [T,B,n], slice t is added
     * to the layer's summed input current after the connections' contributions, before the layer steps */
    const float *ext_current;
    /* LIF / DC layers with per-neuron thresholds (nodes.py:425-498: `thresh` given as a tensor): nullable f32 [n] that replaces
     * p.lif.thresh, broadcast over the batch.  Generic plan (a graph that has one is not offered to the fused plans).  (ABI 8) */
    const float *thresh_vec;
} snn_layer_desc;

typedef struct {
    int kind;                   /* SNN_CONN_* */
    int src, dst;               /* indices into the layer array */
    float *w;                   /* [Nin,N] (CONV2D: [Cout,Cin,KH,KW]) */
    const float *bias;          /* nullable */
    int cin, h, wd, cout, kh, kw, stride, pad;   /* CONV2D geometry */
    int rule;                   /* SNN_RULE_* */
    float nu0, nu1;
    int use_dt;                 /* 1: MCC PostPre multiplies updates by dt */
    float wdecay;               /* multiplicative decay actually applied (1.0 = none) */
    int has_min; float wmin; int has_max; float wmax;
    float *p_plus, *p_minus;    /* MSTDP state [B,Nin] / [B,N] */
    uint8_t *s_src_prev, *s_tgt_prev;
    float reward; const float *reward_vec; float a_plus, a_minus, decay_plus, decay_minus;
    int has_norm; float norm; int norm_abs;   /* post-run normalisation (norm_abs: Connection) */
    float *norm_ws;             /* [N] scratch when has_norm */
    float *e_trace;             /* MSTDPET: dense eligibility trace [Nin,N]; CONV2D + MSTDP: the eligibility [Cout,Cin*KH*KW]
                                   (p_plus is then [Cin,H,W], p_minus [Cout,OH*OW]; batch 1, s_*_prev unused) */
    float decay_e, tc_e;        /* MSTDPET: exp(-dt / tc_e_trace), tc_e_trace */
    float *rule_ws;             /* CONV2D + PostPre: scratch of 2 * B * Cout*Cin*KH*KW floats */
    const uint8_t *mask;        /* nullable [Nin,N] (same layout as w): weights forced to zero after every step's update --
                                   run(..., masks=) / LocalConnection.mask, topology.py:129-133 (generic plan) */
    float *raster_w;            /* nullable [T, numel(w)] weight monitor (Monitor / NetworkMonitor on a connection's `w`,
                                   monitors.py:94-111,222-262): w as it stands at the END of every timestep, i.e. after that
                                   step's learning update and mask and before the post-run normalisation (generic plan) */
} snn_conn_desc;

typedef struct {
    int B, T;
    float dt;
    int learning;               /* Network.learning */
    const float *noise_q;       /* pre-drawn Exp(1) stream for one_spike (see snn_dc_step); nullable */
    long long q_len;
    snn_rng_state *rng;         /* OR: device generator state (preferred; noise_q ignored when set) */
    float *qbuf;                /* with rng: scratch of B * max(DC layer n) floats */
    void *workspace;            /* device scratch for fused plans (snn_net_workspace_bytes); nullable */
    unsigned long long workspace_bytes;
    long long *cursor;          /* device int64[2] */
    int *status;                /* device int32[1]: 0, SNN_ERR_NOISE or SNN_ERR_TIMEOUT after the run */
    int one_step;               /* network.py:388-393 one_step=True: every layer's input is computed right before its own step
                                   from the CURRENT spikes of its sources (those already stepped in this timestep contribute
                                   their new spikes) -- a feed-forward pass per timestep.  Generic plan only. */
    int plan;                   /* 0 = automatic, 1 = generic per-operator launches, 2 = fused plans in their
                                   one-launch-per-timestep form (what a caller re-runs with after SNN_ERR_TIMEOUT),
                                   3 = automatic, but never the lean form of a plan (what a caller re-runs with after
                                   SNN_ERR_RETRY).  A run that reports either status has left every caller-owned STATE
                                   tensor untouched. */
    int *status2;               /* nullable device int32[1] (zeroed by the caller).  Pipelined callers -- those that do not read
                                   *status back before they enqueue the next run -- pass it to have the SECOND attempt of a lean
                                   plan enqueued right behind the first: the general form of the same plan, on the device gated
                                   on *status == SNN_ERR_RETRY (its workgroups return at once otherwise), reporting into *status2.
                                   Afterwards: the run succeeded iff *status == 0, or *status == SNN_ERR_RETRY and *status2 == 0.
                                   (ABI 8) */
    unsigned long long *host_state; /* nullable: 16 bytes of HOST memory that belong to `workspace` -- zeroed by the caller whenever the
                                   workspace is (re)allocated or written by anybody else, otherwise left alone.  The library notes there
                                   what it knows about the workspace's content, so that consecutive pipelined runs need not clear their
                                   exchange area with a memset each (the gated second attempt does it for the next run).  (ABI 8) */
} snn_run_desc;

/* Runs T timesteps.  Asynchronous; the caller synchronises the stream before reading *status /
 * cursor[0].  Picks a fused plan when the graph matches one (snn_plan_name reports which).  */
int snn_net_run(const snn_layer_desc *h_layers, int n_layers, const snn_conn_desc *h_conns, int n_conns,
                const snn_run_desc *h_run, snn_stream_t stream);
/* Device scratch (bytes) a fused plan would need for this network; 0 if none applies.  A run
 * whose descriptor carries less falls back to the generic plan.                               */
unsigned long long snn_net_workspace_bytes(const snn_layer_desc *h_layers, int n_layers, const snn_conn_desc *h_conns,
                                           int n_conns, const snn_run_desc *h_run);
/* Name of the plan the last snn_net_run on this thread used ("generic", "dc2015-fused", ...). */
const char *snn_plan_name(void);
/* Which resident form of the DiehlAndCook2015 plan the last such run of this process took: 0 = general form, 1 / 2 / 3 = first /
 * second / third generation of the lean form (csrc/snn_dc2015_resident.hip, snn_dc2015_async.hip), -1 = one launch per timestep or no
 * such run yet.  ABI 7; a diagnostic for bench.py and the tests (the plan NAME stays "dc2015-resident-lean" for all lean forms). */
int snn_dc2015_last_form(void);
/* Profiling aid (bench.py roofline): when stride > 0, snn_net_run brackets the launches of every
 * stride-th timestep with hipEvents recorded on the run's stream (at most 64 samples per run).
 * snn_profile_collect (call after synchronising the stream) returns the summed elapsed
 * milliseconds and the sample count, and clears the samples.                                 */
void snn_profile_enable(int stride);
int snn_profile_collect(double *h_sum_ms, int *h_samples);
/* Fused-plan bookkeeping: runs issued as plain launches / captured into a hipGraph / replayed from one. */
void snn_graph_stats(int *h_plain, int *h_captured, int *h_replayed);
/* Force a plan for testing: 0 = automatic, 1 = generic per-operator launches only,
 * 2 = fused plans in their one-launch-per-timestep form (no resident kernel), 3 = no lean forms. */
void snn_set_plan_mode(int mode);

/* ---- (e) multi-GPU: one process per GPU, RCCL over xGMI ----------------------------------------
 * SURVEY.md 8(e).  Rank 0 calls snn_dist_unique_id and distributes the 128 bytes; every rank then calls snn_dist_init
 * with the HIP device it will use already current.  snn_dist_allreduce_dw = the north-star schedule (sum the
 * per-input weight / threshold deltas over ranks, in place); snn_dist_allgather_step = the exact per-timestep mode
 * (every rank's packed step factors, concatenated in rank order).  Both are stream-ordered and asynchronous.
 * SNN_ERR_UNSUPPORTED: no RCCL library could be opened on this machine.                               */
typedef struct snn_dist snn_dist;
int snn_dist_unique_id(void *h_id128);
int snn_dist_init(int rank, int world, const void *h_id128, snn_dist **h_out);
int snn_dist_world(const snn_dist *d, int *h_rank, int *h_world);
int snn_dist_allreduce_dw(snn_dist *d, float *buf, long long count, snn_stream_t stream);
int snn_dist_allgather_step(snn_dist *d, const void *send, void *recv, long long bytes_per_rank, snn_stream_t stream);
int snn_dist_destroy(snn_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* SNNHIP_H */

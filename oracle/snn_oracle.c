/*
 * snn_oracle.c -- CPU restatement of the BindsNET Network.run() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (bindsnet_amd + libsnnhip)
 * never links, imports or falls back to it.
 *
 * Every function restates one reference function in plain scalar C (IEEE f32, no FMA:
 * build with -ffp-contract=off, no -ffast-math) and cites the reference file:line it
 * follows (paths relative to /root/reference).  The floating-point *order* of every
 * reduction follows ATen's CPU sum kernel (aten/src/ATen/native/cpu/SumKernel.cpp,
 * torch 2.10: multi_row_sum / row_sum / vectorized_outer_sum), restated in
 * SURVEY.md Appendix A, because that is what the reference executes.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 * fixtures produced by running the unmodified reference (tests/golden/make_golden.py).
 * The order pinned by default is the reference's SERIAL one (= its runs with up to 8 torch threads at BASELINE.md's sizes);
 * orc_set_reference_threads(t) follows the reference as it sums with t threads (the last N mod 32 < 8 columns move to another
 * kernel of SumKernel.cpp), pinned against torch itself and a 16-thread reference run by tests/test_aten_sum_threads.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * ATen float-sum order (SumKernel.cpp: multi_row_sum<acc_t, nrows>, row_sum, and the column
 * split of vectorized_outer_sum).  Loader callback keeps the order code independent of where
 * the summands come from.
 * ---------------------------------------------------------------------------------------- */
typedef float (*orc_term_fn)(const void *ctx, long i);

static int ceil_log2_l(long x)
{
    if (x <= 2) return 1;
    int l = 0; long v = x - 1;
    while (v > 0) { v >>= 1; ++l; }
    return l;
}

/* multi_row_sum for one column: 4-level cascade, block 2^p, p = max(4, ceil_log2(n)/4). */
static float cascade_sum(orc_term_fn f, const void *ctx, long start, long stride, long n)
{
    int p = ceil_log2_l(n) / 4; if (p < 4) p = 4;
    const long step = 1L << p, mask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    long i = 0;
    while (i + step <= n) {
        for (long j = 0; j < step; ++j, ++i) acc[0] += f(ctx, start + i * stride);
        for (int l = 1; l < 4; ++l) {
            acc[l] += acc[l - 1];
            acc[l - 1] = 0.f;
            const long m = mask << (l * p);
            if ((i & m) != 0) break;
        }
    }
    for (; i < n; ++i) acc[0] += f(ctx, start + i * stride);
    for (int l = 1; l < 4; ++l) acc[0] += acc[l];
    return acc[0];
}

/* row_sum: 4 interleaved lanes, each a cascade over n/4 terms, leftovers into lane 0. */
static float row_sum4(orc_term_fn f, const void *ctx, long n)
{
    const long n4 = n / 4;
    float lane[4];
    for (int k = 0; k < 4; ++k) lane[k] = cascade_sum(f, ctx, k, 4, n4);
    for (long i = n4 * 4; i < n; ++i) lane[0] += f(ctx, i);
    for (int k = 1; k < 4; ++k) lane[0] += lane[k];
    return lane[0];
}

/* Reduce n terms that feed output column `col` of `ncols` contiguous columns
 * (vectorized_outer_sum: columns below 32*floor(ncols/32) take the 4-vector multi_row_sum
 * path, the rest take row_sum -- vector or scalar row_sum give the same per-column order). */
static float outer_sum(orc_term_fn f, const void *ctx, long n, long col, long ncols)
{
    if (col < (ncols / 32) * 32) return cascade_sum(f, ctx, 0, 1, n);
    return row_sum4(f, ctx, n);
}

/* The reference's sums are not independent of torch's intra-op thread count (DESIGN.md section 2,
 * tools/probe_aten_sum_threads.py): a reduction of `outer` x n x ncols elements over n runs serially below
 * at::internal::GRAIN_SIZE = 32768 elements; else it is split over the outermost non-reduced dimension that has at least
 * `threads` entries (if neither has: the larger one, ties to the outer) -- and when that is the COLUMNS, cut into `threads`
 * ranges of c = ceil(ncols / threads) whose ends are rounded down to multiples of 32, a last range that holds only the
 * ncols mod 32 < 8 tail columns is summed by scalar_outer_sum: groups of FOUR columns in the cascade order of a full group,
 * the rest row_sum.  orc_set_reference_threads(t) makes orc_prop_mcc / orc_normalize (and the runs built on them) follow
 * the reference AS IT RUNS WITH t THREADS; the default, 1, is the serial order everything else in this repository pins. */
static int g_ref_threads = 1;
ORC_API void orc_set_reference_threads(int t) { g_ref_threads = t > 0 ? t : 1; }
ORC_API int orc_get_reference_threads(void) { return g_ref_threads; }

static int tail_isolated(long outer, long n, long ncols)
{
    const long t = g_ref_threads, tail = ncols % 32;
    if (t <= 1 || tail == 0 || tail >= 8) return 0;
    if (outer * n * ncols < 32768) return 0;
    if (outer >= t) return 0;
    if (ncols < t && ncols <= outer) return 0;
    const long c = (ncols + t - 1) / t;
    return c * ((ncols - 1) / c) >= ncols - tail;
}

static float outer_sum_threads(orc_term_fn f, const void *ctx, long n, long col, long ncols, long outer)
{
    const long full = (ncols / 32) * 32;
    if (col >= full && tail_isolated(outer, n, ncols) && col - full < ((ncols - full) / 4) * 4)
        return cascade_sum(f, ctx, 0, 1, n);
    return outer_sum(f, ctx, n, col, ncols);
}

/* ------------------------------------------------------------------------------------------
 * a5: MulticompartmentConnection.compute + Weight.compute
 *     bindsnet/network/topology.py:437-479, bindsnet/network/topology_features.py:633-645
 *     out[b,j] = sum_i value[i,j] * s[b,i], as broadcast multiply then torch.sum(dim=1).
 * ---------------------------------------------------------------------------------------- */
typedef struct { const float *W; const uint8_t *s; long N; long j; } prop_ctx;
static float prop_term(const void *c, long i)
{
    const prop_ctx *p = (const prop_ctx *)c;
    return p->W[i * p->N + p->j] * (float)p->s[i];
}

ORC_API void orc_prop_mcc(const float *W, const uint8_t *s, float *out,
                          int B, int Nin, int N, int accumulate)
{
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j) {
            prop_ctx c = { W, s + (long)b * Nin, N, j };
            float r = outer_sum_threads(prop_term, &c, Nin, j, N, B);
            /* network.py:240-248: inputs = zeros; inputs += compute(...) per connection */
            out[(long)b * N + j] = (accumulate ? out[(long)b * N + j] : 0.0f) + r;
        }
}

/* a6: Connection.compute, bindsnet/network/topology.py:332-346.  The reference calls MKL
 * sgemm whose order is not reproducible (SURVEY.md finding 5); this is the ORDER-PINNED
 * canonical form the HIP path is held to: ascending-k sequential f32 (spikes are 0/1 so
 * fma(s,w,acc) == acc + s*w exactly). */
ORC_API void orc_prop_dense(const float *W, const float *bias, const uint8_t *s, float *out,
                            int B, int Nin, int N, int accumulate)
{
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j) {
            float acc = 0.f;
            for (int i = 0; i < Nin; ++i) acc += (float)s[(long)b * Nin + i] * W[(long)i * N + j];
            if (bias) acc += bias[j];
            out[(long)b * N + j] = (accumulate ? out[(long)b * N + j] : 0.0f) + acc;
        }
}

/* a7: Conv2dConnection.compute, bindsnet/network/topology.py:799-815 (F.conv2d, oneDNN).
 * Order per SURVEY.md finding 5 / probe P8 and the round-2 probe below: bias-free sequential
 * accumulation over (kh, kw, c_in) -- taps row-major, channels innermost --, then + bias. */
ORC_API void orc_prop_conv2d(const float *W, const float *bias, const uint8_t *s, float *out,
                             int B, int Cin, int H, int Wd, int Cout, int KH, int KW,
                             int stride, int pad, int accumulate)
{
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < OH; ++oy)
                for (int ox = 0; ox < OW; ++ox) {
                    float acc = 0.f;
                    /* probe (round 2, torch 2.10 / oneDNN 3.7.1, threads 1 and 8, strides / paddings / batch sizes):
                     * for C_in <= 16 the reference accumulates taps in (kh, kw) row-major order with the input
                     * channels INNERMOST; for C_in = 1 that is the plain row-major tap order.  Wider inputs take
                     * other oneDNN kernels (C_in = 32: channel-major; 64: blocked) and are rejected by the library. */
                    for (int ky = 0; ky < KH; ++ky)
                        for (int kx = 0; kx < KW; ++kx)
                            for (int ci = 0; ci < Cin; ++ci) {
                                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
                                acc += (float)s[(((long)b * Cin + ci) * H + iy) * Wd + ix] *
                                       W[(((long)co * Cin + ci) * KH + ky) * KW + kx];
                            }
                    if (bias) acc += bias[co];
                    const long o = (((long)b * Cout + co) * OH + oy) * OW + ox;
                    out[o] = (accumulate ? out[o] : 0.0f) + acc;
                }
}

/* ------------------------------------------------------------------------------------------
 * a2: Nodes.forward trace update, bindsnet/network/nodes.py:96-107 (Input.forward :211-221
 *     just aliases s = x first).
 * ---------------------------------------------------------------------------------------- */
static void trace_update(float *x, const uint8_t *s, long n, float trace_decay,
                         float trace_scale, int additive)
{
    for (long k = 0; k < n; ++k) {
        float t = x[k] * trace_decay;
        if (additive) t = t + trace_scale * (float)s[k];
        else if (s[k]) t = trace_scale;
        x[k] = t;
    }
}

ORC_API void orc_input_step(const uint8_t *s, float *x, long n_total, int traces,
                            float trace_decay, float trace_scale, int additive)
{
    if (traces) trace_update(x, s, n_total, trace_decay, trace_scale, additive);
}

/* a3: LIFNodes.forward, bindsnet/network/nodes.py:500-529. `I` is masked in place like the
 * reference does to its input (nodes.py:511). */
ORC_API void orc_lif_step(float *v, float *refrac, uint8_t *s, float *x, float *I,
                          int B, int N, float decay, float rest, float reset, float thresh,
                          float refrac0, float dt, int has_lbound, float lbound,
                          int traces, float trace_decay, float trace_scale, int additive)
{
    const long n = (long)B * N;
    for (long k = 0; k < n; ++k) {
        float vv = v[k] - rest;          /* nodes.py:508 three separately rounded ops */
        vv = decay * vv;
        vv = vv + rest;
        if (refrac[k] > 0.f) I[k] = 0.f;  /* :511 */
        refrac[k] = refrac[k] - dt;       /* :514 */
        vv = vv + I[k];                   /* :516 */
        const uint8_t sp = vv >= thresh;  /* :519 */
        if (sp) { refrac[k] = refrac0; vv = reset; } /* :522-523 */
        if (has_lbound && vv < lbound) vv = lbound;  /* :526-527 */
        v[k] = vv; s[k] = sp;
    }
    if (traces) trace_update(x, s, n, trace_decay, trace_scale, additive);
}

/* a4: DiehlAndCookNodes.forward, bindsnet/network/nodes.py:1069-1111.
 * Q / cursor: pre-drawn torch.empty(K).exponential_(1) stream standing in for the
 * torch.multinomial call at :1100-1102 (multinomial(p,1) ==
 * argmax(p / q) with one draw per element of the [rows_with_spike, N] operand).
 * Returns number of rows that consumed noise, or -1 if Q is too short. */
ORC_API int orc_dc_step(float *v, float *refrac, uint8_t *s, float *x, float *theta,
                        const float *I, int B, int N,
                        float decay, float rest, float reset, float thresh, float refrac0,
                        float dt, float theta_decay, float theta_plus, int learning,
                        int one_spike, int has_lbound, float lbound,
                        int traces, float trace_decay, float trace_scale, int additive,
                        const float *Q, long Qlen, long *cursor)
{
    if (learning)                                  /* :1078-1079 */
        for (int j = 0; j < N; ++j) theta[j] = theta[j] * theta_decay;
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j) {
            const long k = (long)b * N + j;
            float vv = v[k] - rest;                /* :1077 */
            vv = decay * vv;
            vv = vv + rest;
            const float gate = (refrac[k] <= 0.f) ? 1.0f : 0.0f;   /* :1082 */
            const float gx = gate * I[k];
            vv = vv + gx;
            refrac[k] = refrac[k] - dt;            /* :1085 */
            const float th = thresh + theta[j];    /* :1088 */
            const uint8_t sp = vv >= th;
            if (sp) { refrac[k] = refrac0; vv = reset; }   /* :1091-1092 */
            v[k] = vv; s[k] = sp;
        }
    if (learning)                                  /* :1093-1094: sum over batch, pre-winner */
        for (int j = 0; j < N; ++j) {
            float cnt = 0.f;
            for (int b = 0; b < B; ++b) cnt += (float)s[(long)b * N + j];
            theta[j] = theta[j] + theta_plus * cnt;
        }
    int rows = 0;
    if (one_spike) {                               /* :1097-1105 */
        for (int b = 0; b < B; ++b) {
            uint8_t *row = s + (long)b * N;
            int any = 0;
            for (int j = 0; j < N; ++j) any |= row[j];
            if (!any) continue;
            const long off = *cursor + (long)rows * N;
            if (off + N > Qlen) return -1;
            int best = 0; float bestv = -INFINITY;
            for (int j = 0; j < N; ++j) {
                const float val = (float)row[j] / Q[off + j];
                if (val > bestv) { bestv = val; best = j; }   /* argmax: first maximal index */
            }
            memset(row, 0, (size_t)N);
            row[best] = 1;
            ++rows;
        }
        *cursor += (long)rows * N;
    }
    if (has_lbound)                                /* :1108-1109 */
        for (long k = 0; k < (long)B * N; ++k) if (v[k] < lbound) v[k] = lbound;
    if (traces) trace_update(x, s, (long)B * N, trace_decay, trace_scale, additive);  /* :1111 */
    return rows;
}

/* ------------------------------------------------------------------------------------------
 * a8/a9: PostPre.  MCC: bindsnet/learning/MCC_learning.py:224-302 + base update :86-110.
 *        dense: bindsnet/learning/learning.py:390-420 + LearningRule.update :87-104.
 * Batch reduction = torch.sum(dim=0) of the [B, Nin, N] outer products: element e = i*N+j of
 * Nin*N contiguous columns, reduced over b in outer_sum order.
 *   dt_scale: MCC multiplies each reduced update by connection.dt (use_dt=1); dense does not.
 *   decay:    the multiplicative weight decay actually applied (1.0 by default).
 * ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t *sp; const float *tr; long ns, nt, is, it; float nu; int spike_is_tgt; } pp_ctx;
static float pp_term(const void *c, long b)
{
    const pp_ctx *p = (const pp_ctx *)c;
    if (!p->spike_is_tgt)   /* pre: source_s[b,i] * (target_x[b,j] * nu0) */
        return (float)p->sp[b * p->ns + p->is] * (p->tr[b * p->nt + p->it] * p->nu);
    /* post: source_x[b,i] * (target_s[b,j] * nu1) */
    return p->tr[b * p->nt + p->it] * ((float)p->sp[b * p->ns + p->is] * p->nu);
}

ORC_API void orc_postpre(float *W, const uint8_t *s_src, const float *x_src,
                         const uint8_t *s_tgt, const float *x_tgt,
                         int B, int Nin, int N, float nu0, float nu1,
                         int use_dt, float dt, float decay,
                         int has_min, float wmin, int has_max, float wmax)
{
    const long E = (long)Nin * N;
    for (int i = 0; i < Nin; ++i)
        for (int j = 0; j < N; ++j) {
            const long e = (long)i * N + j;
            float w = W[e];
            if (nu0 != 0.f) {
                pp_ctx c = { s_src, x_tgt, Nin, N, i, j, nu0, 0 };
                float u = outer_sum(pp_term, &c, B, e, E);
                if (use_dt) u = u * dt;
                w = w - u;
            }
            if (nu1 != 0.f) {
                pp_ctx c = { s_tgt, x_src, N, Nin, j, i, nu1, 1 };
                float u = outer_sum(pp_term, &c, B, e, E);
                if (use_dt) u = u * dt;
                w = w + u;
            }
            w = w * decay;
            if (has_min && w < wmin) w = wmin;
            if (has_max && w > wmax) w = wmax;
            W[e] = w;
        }
}

/* f3: Hebbian._connection_update (learning.py:1110-1135) and WeightDependentPostPre._connection_update
 * (learning.py:626-653) on a dense Connection, then LearningRule.update (:87-104: *= weight_decay, clamp).
 * Both reduce the raw outer products s_src (x) x_tgt and x_src (x) s_tgt over the batch first (bmm with K = 1 is one
 * exact multiply per element; torch.sum(dim=0) order) and scale afterwards:
 *   Hebbian:  w += nu0 * U1;  w += nu1 * U2
 *   WDPP:     update = 0 - (nu0 * U1) * (w - wmin)   [if nu0];  update += (nu1 * U2) * (wmax - w)   [if nu1];  w += update */
static float raw_term(const void *c, long b)
{
    const pp_ctx *p = (const pp_ctx *)c;
    return !p->spike_is_tgt ? (float)p->sp[b * p->ns + p->is] * p->tr[b * p->nt + p->it]
                            : p->tr[b * p->nt + p->it] * (float)p->sp[b * p->ns + p->is];
}

ORC_API void orc_hebbian_wdpp(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                              int B, int Nin, int N, float nu0, float nu1, int weight_dependent, float decay,
                              int has_min, float wmin, int has_max, float wmax)
{
    const long E = (long)Nin * N;
    for (int i = 0; i < Nin; ++i)
        for (int j = 0; j < N; ++j) {
            const long e = (long)i * N + j;
            float w = W[e];
            pp_ctx c1 = { s_src, x_tgt, Nin, N, i, j, 1.0f, 0 }, c2 = { s_tgt, x_src, N, Nin, j, i, 1.0f, 1 };
            if (!weight_dependent) {
                const float u1 = outer_sum(raw_term, &c1, B, e, E);
                w = w + nu0 * u1;
                const float u2 = outer_sum(raw_term, &c2, B, e, E);
                w = w + nu1 * u2;
            } else {
                float upd = 0.f; int have = 0;
                if (nu0 != 0.f) { const float u1 = outer_sum(raw_term, &c1, B, e, E); upd = 0.0f - (nu0 * u1) * (w - wmin); have = 1; }
                if (nu1 != 0.f) { const float u2 = outer_sum(raw_term, &c2, B, e, E); const float y = (nu1 * u2) * (wmax - w); upd = have ? upd + y : y; have = 1; }
                if (have) w = w + upd;
            }
            w = w * decay;
            if (has_min && w < wmin) w = wmin;
            if (has_max && w > wmax) w = wmax;
            W[e] = w;
        }
}

/* f4: PostPre._conv2d_connection_update, learning.py:457-497 (+ LearningRule.update :87-104).
 * pre[co,k]  = sum_b sum_l x_tgt[b,co,l] * unfold(s_src)[b,k,l];  w -= nu0 * pre
 * post[co,k] = sum_b sum_l s_tgt[b,co,l] * unfold(x_src)[b,k,l];  w += nu1 * post       (k = (ci,kh,kw), l = (oy,ox))
 * The reference's inner sum over l runs inside torch.bmm (BLAS order, not reproducible: SURVEY.md finding 5); the
 * canonical order pinned here is ascending l, sequential f32, then torch.sum(dim=0) order over the batch.            */
typedef struct { const float *part; long E, e; } cp_ctx;
static float cp_term(const void *c, long b) { const cp_ctx *p = (const cp_ctx *)c; return p->part[b * p->E + p->e]; }

ORC_API void orc_conv2d_postpre(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                                int B, int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad,
                                float nu0, float nu1, float decay, int has_min, float wmin, int has_max, float wmax)
{
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1, L = OH * OW;
    const long K = (long)Cin * KH * KW, E = (long)Cout * K;
    float *pre = (float *)calloc((size_t)B * E, sizeof(float)), *post = (float *)calloc((size_t)B * E, sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci)
                for (int ky = 0; ky < KH; ++ky)
                    for (int kx = 0; kx < KW; ++kx) {
                        float a = 0.f, p = 0.f;
                        for (int l = 0; l < L; ++l) {
                            const int oy = l / OW, ox = l - oy * OW;
                            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                            const int in = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
                            const long si = (((long)b * Cin + ci) * H + (in ? iy : 0)) * Wd + (in ? ix : 0);
                            const long ti = ((long)b * Cout + co) * L + l;
                            a += x_tgt[ti] * (in ? (float)s_src[si] : 0.0f);
                            p += (float)s_tgt[ti] * (in ? x_src[si] : 0.0f);
                        }
                        const long e = (long)co * K + ((long)ci * KH + ky) * KW + kx;
                        pre[b * E + e] = a; post[b * E + e] = p;
                    }
    for (long e = 0; e < E; ++e) {
        float w = W[e];
        if (nu0 != 0.f) { cp_ctx c = { pre, E, e }; w = w - nu0 * outer_sum(cp_term, &c, B, e, E); }
        if (nu1 != 0.f) { cp_ctx c = { post, E, e }; w = w + nu1 * outer_sum(cp_term, &c, B, e, E); }
        w = w * decay;
        if (has_min && w < wmin) w = wmin;
        if (has_max && w > wmax) w = wmax;
        W[e] = w;
    }
    free(pre); free(post);
}

/* f4: MSTDP._conv2d_connection_update, learning.py:1942-2015 (+ LearningRule.update :87-104), batch size 1 -- the only
 * batch size at which the reference's `eligibility.view(w.size())` (:2013) is defined.  State: E = eligibility
 * [Cout, K] (K = Cin*KH*KW), P = P^+ and Q = P^-.  The reference keeps P^+ unfolded ([K, L], im2col of zeros at :1982-1988);
 * every unfolded element goes through exactly the operations of the input pixel it copies (padding stays 0*d + a*0 = 0),
 * so P is held in input space [Cin, H, W] and unfolded on the fly.  Order of one call:
 *   w += nu0 * torch.sum(reward * E, dim=0)      -- :1966-1967.  After the first call E has the WEIGHT's shape, so this
 *        sum runs over the OUTPUT CHANNELS and its [Cin, KH, KW] result is broadcast back over them (the reference's
 *        behaviour, reproduced as is); during the first call E is zeros and the term vanishes either way;
 *   P = P * decay_plus + a_plus * s_src;  Q = Q * decay_minus + a_minus * s_tgt                       -- :1998-2001
 *   E[co,k] = sum_l s_tgt[co,l] * P_unf[k,l]  +  sum_l Q[co,l] * s_src_unf[k,l]                       -- :2004-2007
 *        (two torch.bmm calls: BLAS order, not reproducible -- canonical order pinned here: ascending l, sequential f32)
 *   w *= weight_decay; clamp                                                                          -- :87-104    */
typedef struct { const float *E; long K, k; float reward; } cm_ctx;
static float cm_term(const void *c, long co) { const cm_ctx *p = (const cm_ctx *)c; return p->reward * p->E[co * p->K + p->k]; }

ORC_API void orc_conv2d_mstdp(float *W, float *E, float *P, float *Q, const uint8_t *s_src, const uint8_t *s_tgt,
                              int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad,
                              float reward, float nu0, float a_plus, float a_minus, float decay_plus, float decay_minus,
                              float wdecay, int has_min, float wmin, int has_max, float wmax)
{
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1, L = OH * OW;
    const long K = (long)Cin * KH * KW;
    for (long k = 0; k < K; ++k) {
        cm_ctx c = { E, K, k, reward };
        const float S = outer_sum(cm_term, &c, Cout, k, K);
        for (int co = 0; co < Cout; ++co) {
            float w = W[co * K + k] + nu0 * S;
            w = w * wdecay;
            if (has_min && w < wmin) w = wmin;
            if (has_max && w > wmax) w = wmax;
            W[co * K + k] = w;
        }
    }
    for (long i = 0; i < (long)Cin * H * Wd; ++i) { float p = P[i] * decay_plus; P[i] = p + a_plus * (float)s_src[i]; }
    for (long i = 0; i < (long)Cout * L; ++i) { float q = Q[i] * decay_minus; Q[i] = q + a_minus * (float)s_tgt[i]; }
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx) {
                    float a = 0.f, b = 0.f;
                    for (int l = 0; l < L; ++l) {
                        const int oy = l / OW, ox = l - oy * OW;
                        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                        const int in = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
                        const long si = ((long)ci * H + (in ? iy : 0)) * Wd + (in ? ix : 0);
                        a += (float)s_tgt[(long)co * L + l] * (in ? P[si] : 0.0f);
                        b += Q[(long)co * L + l] * (in ? (float)s_src[si] : 0.0f);
                    }
                    E[(long)co * K + ((long)ci * KH + ky) * KW + kx] = a + b;
                }
}

/* f3: MSTDPET._connection_update, learning.py:2187-2248 (batch size 1: the reference flattens the spikes).
 * elig / e_trace are the reference's dense [Nin,N] tensors.  Order: e_trace *= exp(-dt/tc_e); e_trace += elig / tc_e;
 * w += ((nu0 * dt) * reward) * e_trace; p_plus / p_minus; elig = p_plus (x) s_tgt + s_src (x) p_minus; decay; clamp. */
ORC_API void orc_mstdpet(float *W, float *elig, float *e_trace, float *p_plus, float *p_minus,
                         const uint8_t *s_src, const uint8_t *s_tgt, int Nin, int N, float reward, float nu0, float dt,
                         float a_plus, float a_minus, float decay_plus, float decay_minus, float decay_e, float tc_e,
                         float wdecay, int has_min, float wmin, int has_max, float wmax)
{
    const float scale = (nu0 * dt) * reward;
    for (long e = 0; e < (long)Nin * N; ++e) {
        float et = e_trace[e] * decay_e;
        et = et + elig[e] / tc_e;
        e_trace[e] = et;
        W[e] = W[e] + scale * et;
    }
    for (int i = 0; i < Nin; ++i) { const float p = p_plus[i] * decay_plus; p_plus[i] = p + a_plus * (float)s_src[i]; }
    for (int j = 0; j < N; ++j) { const float p = p_minus[j] * decay_minus; p_minus[j] = p + a_minus * (float)s_tgt[j]; }
    for (int i = 0; i < Nin; ++i)
        for (int j = 0; j < N; ++j)
            elig[(long)i * N + j] = p_plus[i] * (float)s_tgt[j] + (float)s_src[i] * p_minus[j];
    for (long e = 0; e < (long)Nin * N; ++e) {
        float w = W[e] * wdecay;
        if (has_min && w < wmin) w = wmin;
        if (has_max && w > wmax) w = wmax;
        W[e] = w;
    }
}

/* a10: MSTDP._connection_update, bindsnet/learning/learning.py:1504-1574.
 * elig is the dense [B,Nin,N] eligibility of the reference (kept dense here on purpose). */
typedef struct { const float *elig; long E; long e; float reward; const float *reward_vec; } ms_ctx;
static float ms_term(const void *c, long b)
{
    const ms_ctx *p = (const ms_ctx *)c;
    const float r = p->reward_vec ? p->reward_vec[b] : p->reward;
    return r * p->elig[b * p->E + p->e];
}

ORC_API void orc_mstdp(float *W, float *elig, float *p_plus, float *p_minus,
                       const uint8_t *s_src, const uint8_t *s_tgt,
                       int B, int Nin, int N, float reward, const float *reward_vec,
                       float nu0, float a_plus, float a_minus, float decay_plus,
                       float decay_minus, float wdecay,
                       int has_min, float wmin, int has_max, float wmax)
{
    const long E = (long)Nin * N;
    for (long e = 0; e < E; ++e) {                 /* :1558-1561 */
        ms_ctx c = { elig, E, e, reward, reward_vec };
        const float u = outer_sum(ms_term, &c, B, e, E);
        W[e] = W[e] + nu0 * u;
    }
    for (long k = 0; k < (long)B * Nin; ++k) {     /* :1564-1565 */
        float p = p_plus[k] * decay_plus;
        p_plus[k] = p + a_plus * (float)s_src[k];
    }
    for (long k = 0; k < (long)B * N; ++k) {       /* :1566-1567 */
        float p = p_minus[k] * decay_minus;
        p_minus[k] = p + a_minus * (float)s_tgt[k];
    }
    for (int b = 0; b < B; ++b)                    /* :1570-1572 */
        for (int i = 0; i < Nin; ++i)
            for (int j = 0; j < N; ++j) {
                const float a = p_plus[(long)b * Nin + i] * (float)s_tgt[(long)b * N + j];
                const float c2 = (float)s_src[(long)b * Nin + i] * p_minus[(long)b * N + j];
                elig[(long)b * E + (long)i * N + j] = a + c2;
            }
    for (long e = 0; e < E; ++e) {                 /* learning.py:92-104 */
        float w = W[e] * wdecay;
        if (has_min && w < wmin) w = wmin;
        if (has_max && w > wmax) w = wmax;
        W[e] = w;
    }
}

/* a11: AbstractFeature.normalize (signed column sum), topology_features.py:250-266, and
 * Connection.normalize (abs column sum), topology.py:383-392.  `norm / colsum` is evaluated
 * by torch as reciprocal(colsum) * norm (python scalar / tensor). */
typedef struct { const float *W; long N; long j; int use_abs; } nm_ctx;
static float nm_term(const void *c, long i)
{
    const nm_ctx *p = (const nm_ctx *)c;
    const float w = p->W[i * p->N + p->j];
    return p->use_abs ? fabsf(w) : w;
}

ORC_API void orc_normalize(float *W, int Nin, int N, float norm, int use_abs)
{
    float *scale = (float *)malloc(sizeof(float) * (size_t)N);
    for (int j = 0; j < N; ++j) {
        nm_ctx c = { W, N, j, use_abs };
        float cs = outer_sum_threads(nm_term, &c, Nin, j, N, 1);
        if (cs == 0.f) cs = 1.0f;
        const float rc = 1.0f / cs;
        scale[j] = rc * norm;
    }
    for (int i = 0; i < Nin; ++i)
        for (int j = 0; j < N; ++j) W[(long)i * N + j] = W[(long)i * N + j] * scale[j];
    free(scale);
}

/* a11 for Conv2dConnection: normalize(), bindsnet/network/topology.py:824-837 -- W viewed as [F = Cout*Cin, K = KH*KW]; every filter
 * is scaled by norm / w[f].sum(0).  The 1-D sum of a contiguous row is SumKernel.cpp's vectorized_inner_sum with 8-float vectors
 * (what torch 2.10 runs under every ATEN_CPU_CAPABILITY; probed): lane l of 8 takes x[l], x[8+l], ... through row_sum over the
 * K / 8 vectors; a fresh accumulator then takes the K mod 8 leftover elements in order and after them the 8 lane sums in lane order;
 * rows shorter than one vector take the scalar row_sum.  float / tensor is reciprocal(tensor) * float in torch; no zero guard. */
typedef struct { const float *x; long stride, off; } lane_ctx;
static float lane_term(const void *c, long i)
{
    const lane_ctx *p = (const lane_ctx *)c;
    return p->x[i * p->stride + p->off];
}

static float inner_sum8(const float *x, long n)
{
    if (n < 8) { lane_ctx c = { x, 1, 0 }; return row_sum4(lane_term, &c, n); }
    const long vs = n / 8;
    float part[8];
    for (int l = 0; l < 8; ++l) { lane_ctx c = { x, 8, l }; part[l] = row_sum4(lane_term, &c, vs); }
    float fin = 0.f;
    for (long k = vs * 8; k < n; ++k) fin += x[k];
    for (int l = 0; l < 8; ++l) fin += part[l];
    return fin;
}

ORC_API float orc_inner_sum(const float *x, long n) { return inner_sum8(x, n); }

ORC_API void orc_normalize_conv2d(float *W, int F, int K, float norm)
{
    for (int f = 0; f < F; ++f) {
        float *w = W + (long)f * K;
        const float rc = 1.0f / inner_sum8(w, K);
        const float scale = rc * norm;
        for (int k = 0; k < K; ++k) w[k] = w[k] * scale;
    }
}

/* ------------------------------------------------------------------------------------------
 * a1: Network.run for the DiehlAndCook2015 graph, bindsnet/network/network.py:380-465 with
 * the wiring of bindsnet/models/models.py:156-244 (layers X, Ae, Ai in that order;
 * connections X->Ae (PostPre), Ae->Ai, Ai->Ae in that order).
 * State in/out: everything the reference keeps between run() calls.
 * rasters (optional): [T,B,N] u8 for Ae and Ai.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int B, Nin, N, T;
    float dt;
    /* X (Input, traces) */
    float x_trace_decay, x_trace_scale;
    /* Ae (DiehlAndCookNodes) */
    float e_decay, e_rest, e_reset, e_thresh, e_refrac, e_theta_decay, e_theta_plus,
          e_trace_decay, e_trace_scale;
    int e_one_spike;
    /* Ai (LIFNodes, no traces) */
    float i_decay, i_rest, i_reset, i_thresh, i_refrac;
    /* X->Ae PostPre (MCC) */
    float nu0, nu1, wmin, wmax, norm;
    int learning;
} orc_dc_params;

ORC_API int orc_run_dc2015(const orc_dc_params *P,
                           float *W_xe, const float *W_ei, const float *W_ie,
                           const uint8_t *inputs,        /* [T,B,Nin] */
                           uint8_t *sX_prev,             /* [B,Nin] X.s at entry / exit */
                           float *xX,                    /* [B,Nin] X trace */
                           float *vE, float *rE, uint8_t *sE, float *xE, float *theta,
                           float *vI, float *rI, uint8_t *sI,
                           const float *Q, long Qlen, long *cursor,
                           uint8_t *rasterE, uint8_t *rasterI)
{
    const int B = P->B, Nin = P->Nin, N = P->N;
    float *IE = (float *)malloc(sizeof(float) * (size_t)B * N);
    float *II = (float *)malloc(sizeof(float) * (size_t)B * N);
    const uint8_t *sX = sX_prev;
    int rc = 0;
    for (int t = 0; t < P->T; ++t) {
        /* network.py:384 _get_inputs(): connection insertion order, previous-step spikes */
        orc_prop_mcc(W_xe, sX, IE, B, Nin, N, 0);
        orc_prop_mcc(W_ei, sE, II, B, N, N, 0);
        orc_prop_mcc(W_ie, sI, IE, B, N, N, 1);
        /* network.py:386-413 layers in insertion order */
        sX = inputs + (long)t * B * Nin;
        orc_input_step(sX, xX, (long)B * Nin, 1, P->x_trace_decay, P->x_trace_scale, 0);
        const int r = orc_dc_step(vE, rE, sE, xE, theta, IE, B, N, P->e_decay, P->e_rest,
                                  P->e_reset, P->e_thresh, P->e_refrac, P->dt,
                                  P->e_theta_decay, P->e_theta_plus, P->learning,
                                  P->e_one_spike, 0, 0.f, 1, P->e_trace_decay,
                                  P->e_trace_scale, 0, Q, Qlen, cursor);
        if (r < 0) { rc = -1; break; }
        orc_lif_step(vI, rI, sI, NULL, II, B, N, P->i_decay, P->i_rest, P->i_reset,
                     P->i_thresh, P->i_refrac, P->dt, 0, 0.f, 0, 0.f, 0.f, 0);
        /* network.py:431-454 connection updates (only X->Ae has a rule) */
        if (P->learning)
            orc_postpre(W_xe, sX, xX, sE, xE, B, Nin, N, P->nu0, P->nu1, 1, P->dt, 1.0f,
                        1, P->wmin, 1, P->wmax);
        if (rasterE) memcpy(rasterE + (long)t * B * N, sE, (size_t)B * N);
        if (rasterI) memcpy(rasterI + (long)t * B * N, sI, (size_t)B * N);
    }
    if (rc == 0) {
        /* network.py:464-465: normalize every connection (only X->Ae has a norm) */
        if (P->norm > 0.0f) orc_normalize(W_xe, Nin, N, P->norm, 0);   /* norm <= 0: a run whose caller normalises later
                                                                          (the batch-sharded schedule merges first) */
        if (P->T > 0) memcpy(sX_prev, inputs + (long)(P->T - 1) * B * Nin, (size_t)B * Nin);
    }
    free(IE); free(II);
    return rc;
}

/* a1 (dense family): Input -> Connection -> LIFNodes with PostPre or MSTDP, the graph of
 * TwoLayerNetwork (bindsnet/models/models.py:21-91) and of cfg5.
 * rule: 0 none, 1 PostPre (learning.py:390-420), 2 MSTDP (learning.py:1504-1574), 3 Hebbian (:1110-1135),
 *       4 WeightDependentPostPre (:626-653), 5 MSTDPET (:2187-2248, batch 1; `elig` / `e_trace` are [Nin,N]). */
typedef struct {
    int B, Nin, N, T, rule;
    float dt;
    float x_trace_decay, x_trace_scale; int x_traces;
    float decay, rest, reset, thresh, refrac; int y_traces; float y_trace_decay, y_trace_scale;
    float nu0, nu1; int has_min, has_max; float wmin, wmax; int has_norm; float norm;
    float reward, a_plus, a_minus, decay_plus, decay_minus;
    int learning;
    float decay_e, tc_e;   /* MSTDPET: exp(-dt / tc_e_trace), tc_e_trace */
    int mcc;    /* 1: MulticompartmentConnection + Weight instead of a dense Connection: propagation in ATen's sum(dim=1)
                   order (topology.py:437-479), PostPre scaled by dt (MCC_learning.py:224-302), MSTDP as
                   MCC_learning.py:468-551 (same arithmetic as learning.py:1504-1574), signed column sums in normalize
                   (topology_features.py:250-266) */
} orc_two_params;

ORC_API void orc_run_two_layer(const orc_two_params *P, float *W, const float *bias,
                               const uint8_t *inputs, uint8_t *sX_prev, float *xX,
                               float *vY, float *rY, uint8_t *sY, float *xY,
                               float *elig, float *p_plus, float *p_minus,
                               const float *I_forced,   /* optional [T,B,N] teacher-forced currents */
                               uint8_t *rasterY, float *e_trace)
{
    const int B = P->B, Nin = P->Nin, N = P->N;
    float *I = (float *)malloc(sizeof(float) * (size_t)B * N);
    const uint8_t *sX = sX_prev;
    for (int t = 0; t < P->T; ++t) {
        if (I_forced) memcpy(I, I_forced + (long)t * B * N, sizeof(float) * (size_t)B * N);
        else if (P->mcc) orc_prop_mcc(W, sX, I, B, Nin, N, 0);
        else orc_prop_dense(W, bias, sX, I, B, Nin, N, 0);
        sX = inputs + (long)t * B * Nin;
        orc_input_step(sX, xX, (long)B * Nin, P->x_traces, P->x_trace_decay, P->x_trace_scale, 0);
        orc_lif_step(vY, rY, sY, xY, I, B, N, P->decay, P->rest, P->reset, P->thresh,
                     P->refrac, P->dt, 0, 0.f, P->y_traces, P->y_trace_decay,
                     P->y_trace_scale, 0);
        if (P->learning && P->rule == 1)
            orc_postpre(W, sX, xX, sY, xY, B, Nin, N, P->nu0, P->nu1, P->mcc, P->dt, 1.0f,
                        P->has_min, P->wmin, P->has_max, P->wmax);
        else if (P->learning && P->rule == 2)
            orc_mstdp(W, elig, p_plus, p_minus, sX, sY, B, Nin, N, P->reward, NULL, P->nu0,
                      P->a_plus, P->a_minus, P->decay_plus, P->decay_minus, 1.0f,
                      P->has_min, P->wmin, P->has_max, P->wmax);
        else if (P->learning && (P->rule == 3 || P->rule == 4))
            orc_hebbian_wdpp(W, sX, xX, sY, xY, B, Nin, N, P->nu0, P->nu1, P->rule == 4, 1.0f,
                             P->has_min, P->wmin, P->has_max, P->wmax);
        else if (P->learning && P->rule == 5)
            orc_mstdpet(W, elig, e_trace, p_plus, p_minus, sX, sY, Nin, N, P->reward, P->nu0, P->dt, P->a_plus, P->a_minus,
                        P->decay_plus, P->decay_minus, P->decay_e, P->tc_e, 1.0f, P->has_min, P->wmin, P->has_max, P->wmax);
        if (rasterY) memcpy(rasterY + (long)t * B * N, sY, (size_t)B * N);
    }
    if (P->has_norm) orc_normalize(W, Nin, N, P->norm, P->mcc ? 0 : 1);
    if (P->T > 0) memcpy(sX_prev, inputs + (long)(P->T - 1) * B * Nin, (size_t)B * Nin);
    free(I);
}

/* ------------------------------------------------------------------------------------------
 * Host RNG stream used by one_spike: torch CPU generator = mt19937; exponential_(1) on a
 * float tensor draws random64() per element, u = (r & (2^53-1)) * 2^-53 and returns
 * (float)(-log1p(-u))  (ATen core/TransformationHelper.h, DistributionTemplates.h; verified
 * against torch 2.10 in tests/test_oracle_golden.py).  state[624], *pos in [0,624] (624 =>
 * twist before next output), exactly at::mt19937's (state_, next_) with left_ = 624 - pos...
 * expressed as a plain index.
 * ---------------------------------------------------------------------------------------- */
static void mt_twist(uint32_t *mt)
{
    for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}
static uint32_t mt_next(uint32_t *mt, int *pos)
{
    if (*pos >= 624) { mt_twist(mt); *pos = 0; }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
ORC_API void orc_mt_exponential(uint32_t *mt, int *pos, float *out, long n)
{
    for (long k = 0; k < n; ++k) {
        const uint64_t hi = mt_next(mt, pos), lo = mt_next(mt, pos);
        const uint64_t r = (hi << 32) | lo;
        const double u = (double)(r & ((1ULL << 53) - 1)) * 0x1.0p-53;
        out[k] = (float)(-1.0 * log1p(-u));
    }
}


/* ======================================================================================================================
 * orc_encode_poisson -- the CPU statement of libsnnhip's snn_encode_poisson (csrc/snn_encode.hip): Poisson spike trains in the construction
 * of bindsnet/encoding/encodings.py:101-152 (intervals ~ Poisson(lambda), lambda = 1 / x * (1000 / dt) in f32, zero intervals bumped to one,
 * cumulated; time index 0 dropped) from a SPECIFIED counter-based stream.  The reference's own stream cannot be produced in parallel
 * (ATen's sampler consumes a data-dependent number of generator outputs per element), so this is the one place where the oracle restates
 * the library's specification instead of the reference's arithmetic; what ties it to the reference is the distribution
 * (tests/test_gpu_encoding.py: rates and inter-spike intervals against bindsnet.encoding.poisson).  Every operation is an IEEE f32 / f64
 * add, multiply, divide, f32 sqrt or an integer conversion (built with -ffp-contract=off): exp / log / log k! are the fixed series
 * below, not libm.
 *   uniforms: Philox-4x32-10, key = seed, counter = (block, 0, element lo, element hi); a block's four words are used last first;
 *             u = ((r >> 8) + 1) * 2^-24.   lambda < 30: multiplication method (running f64 product against exp(-lambda));
 *             otherwise Hoermann's PTRS in f64 with sqrt(lambda) taken in f32.
 * ====================================================================================================================== */
typedef struct { uint32_t key[2], ctr[4], out[4]; int have; } pz_philox;

static void pz_block(pz_philox *g)
{
    uint32_t c0 = g->ctr[0], c1 = g->ctr[1], c2 = g->ctr[2], c3 = g->ctr[3], k0 = g->key[0], k1 = g->key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    g->out[0] = c0; g->out[1] = c1; g->out[2] = c2; g->out[3] = c3;
}

static float pz_uniform(pz_philox *g)
{
    if (!g->have) { pz_block(g); if (++g->ctr[0] == 0) ++g->ctr[1]; g->have = 4; }
    const uint32_t r = g->out[--g->have];
    return ((float)(r >> 8) + 1.0f) * 5.9604644775390625e-08f;
}

static double pz_floor(double x) { double t = (double)(long long)x; if (t > x) t -= 1.0; return t; }

static double pz_scale(double x, int k)
{
    int64_t b; memcpy(&b, &x, 8); b += (int64_t)k << 52; memcpy(&x, &b, 8); return x;
}

static double pz_exp(double y)
{
    const double k = pz_floor(y * 1.4426950408889634 + 0.5);
    const double r = (y - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0; p = p * r + 1.0 / 3628800.0; p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0; p = p * r + 1.0 / 5040.0; p = p * r + 1.0 / 720.0; p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0; p = p * r + 1.0 / 6.0; p = p * r + 0.5; p = p * r + 1.0; p = p * r + 1.0;
    return pz_scale(p, (int)k);
}

static double pz_log(double x)
{
    int64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    int64_t mb = (bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll;
    double m; memcpy(&m, &mb, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0; p = p * z + 1.0 / 7.0; p = p * z + 1.0 / 5.0; p = p * z + 1.0 / 3.0; p = p * z + 1.0;
    return (double)e * 6.93147180559945286227e-01 + 2.0 * s * p;
}

static double pz_lfact(double k)
{
    if (k < 10.0) {
        double f = 1.0;
        for (double i = 2.0; i <= k; i += 1.0) f = f * i;
        return pz_log(f);
    }
    const double z = k + 1.0, zi = 1.0 / z, z2 = zi * zi;
    double c = -1.0 / 1680.0;
    c = c * z2 + 1.0 / 1260.0; c = c * z2 - 1.0 / 360.0; c = c * z2 + 1.0 / 12.0;
    return ((z - 0.5) * pz_log(z) - z) + 0.91893853320467278056 + c * zi;
}

static double pz_sample(pz_philox *g, float lamf)
{
    if (!(lamf > 0.f)) return 0.0;
    const double lam = (double)lamf;
    if (lamf < 30.f) {
        const double limit = pz_exp(-lam);
        double prod = (double)pz_uniform(g), k = 0.0;
        while (prod > limit) { prod = prod * (double)pz_uniform(g); k += 1.0; }
        return k;
    }
    const double slam = (double)sqrtf(lamf), loglam = pz_log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        const double U = (double)pz_uniform(g) - 0.5, V = (double)pz_uniform(g);
        const double us = 0.5 - (U < 0.0 ? -U : U);
        const double k = pz_floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (pz_log(V) + pz_log(invalpha) - pz_log(a / (us * us) + b) <= (k * loglam - lam) - pz_lfact(k)) return k;
    }
}

/* out: u8 [steps][n] */
ORC_API void orc_encode_poisson(const float *datum, long n, int steps, float dt, uint64_t seed, uint8_t *out)
{
    for (long i = 0; i < n; ++i) {
        const float x = datum[i];
        const float lam = x != 0.f ? 1.0f / x * (1000.0f / dt) : 0.f;
        pz_philox g;
        g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32);
        g.ctr[0] = 0; g.ctr[1] = 0; g.ctr[2] = (uint32_t)(uint64_t)i; g.ctr[3] = (uint32_t)((uint64_t)i >> 32);
        g.have = 0;
        long long next = 0;
        double k = pz_sample(&g, lam);
        if (x != 0.f && k == 0.0) k = 1.0;
        next += (long long)k;
        for (int t = 1; t <= steps; ++t) {
            uint8_t s = 0;
            if (x != 0.f) {
                while (next < t) { k = pz_sample(&g, lam); if (k == 0.0) k = 1.0; next += (long long)k; }
                if (next == t) s = 1;
            }
            out[(size_t)(t - 1) * (size_t)n + (size_t)i] = s;
        }
    }
}

// snn_common.hpp -- elementwise neuron updates shared by the per-op kernels and the fused drivers.
// Each function restates one reference forward() in its exact f32 op order (no FMA: the library is
// built with -ffp-contract=off, so every * and + below is a separately rounded instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/snnhip.h"

int snn_check_launch();            // snn_api.hip: hipGetLastError() -> SNN_* code
int snn_check(hipError_t e);

namespace snn {

// Nodes.forward trace, bindsnet/network/nodes.py:96-103.
__host__ __device__ __forceinline__ float trace_next(float x, uint8_t s, float decay, float scale, int additive) {
    float t = x * decay;
    if (additive) t = t + scale * (float)s;
    else if (s) t = scale;
    return t;
}

// LIFNodes.forward, bindsnet/network/nodes.py:508-527.  `cur` must already be zeroed by the
// caller where rc > 0 (nodes.py:511 masks with the refractory counter BEFORE it is decremented).
__host__ __device__ __forceinline__ uint8_t lif_update(float &v, float &rc, float cur, const snn_lif_params &p) {
    float vv = v - p.rest;              // :508  decay * (v - rest) + rest, three roundings
    vv = p.decay * vv;
    vv = vv + p.rest;
    rc = rc - p.dt;                     // :514
    vv = vv + cur;                      // :516
    const uint8_t sp = vv >= p.thresh;  // :519
    if (sp) { rc = p.refrac; vv = p.reset; }                 // :522-523
    if (p.has_lbound && vv < p.lbound) vv = p.lbound;        // :526-527
    v = vv;
    return sp;
}

// DiehlAndCookNodes.forward membrane part, bindsnet/network/nodes.py:1077-1092 (+ :1108-1109).
// thr = thresh + theta[j] (already decayed), computed once per neuron by the caller.
__host__ __device__ __forceinline__ uint8_t dc_update(float &v, float &rc, float cur, float thr, const snn_lif_params &p) {
    float vv = v - p.rest;              // :1077
    vv = p.decay * vv;
    vv = vv + p.rest;
    const float gate = (rc <= 0.f) ? 1.0f : 0.0f;            // :1082 (refrac_count <= 0).float() * x
    const float gx = gate * cur;
    vv = vv + gx;
    rc = rc - p.dt;                     // :1085
    const uint8_t sp = vv >= thr;       // :1088
    if (sp) { rc = p.refrac; vv = p.reset; }                 // :1091-1092
    if (p.has_lbound && vv < p.lbound) vv = p.lbound;        // :1108-1109
    v = vv;
    return sp;
}

}  // namespace snn

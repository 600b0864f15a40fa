// Host-side check of bindsnet_amd/csrc/snn_conv_events.hpp: the build container has no GPU, so the __host__ __device__ bodies of
// the event-driven Conv2d-PostPre partial sums are run HERE on the CPU (compiled by hipcc like the kernels, no device code is
// executed), one weight element at a time, exactly as k_conv_pp_partial_ev's threads call them.  Test infrastructure only
// (tests/test_conv_events_host.py); not part of libsnnhip.
#include <stdint.h>
#include <vector>
#include "../../bindsnet_amd/csrc/snn_conv_events.hpp"

extern "C" int hostcheck_conv_pp_partials(const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt, int B, int Cin,
                                          int H, int Wd, int Cout, int KH, int KW, int stride, int pad, int use_events, float *part) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0 || Wd > 32 || OW > 32) return -1;
    const snn::ConvGeom g{Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW};
    const int KK = KH * KW;
    const long K = (long)Cin * KK, E = (long)Cout * K;
    int multi_any = 0;
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cin; ++ci) {                    // = one workgroup of k_conv_pp_partial_ev
            std::vector<uint32_t> srow(H), trow((size_t)Cout * OH);
            int multi = 0;
            for (int r = 0; r < H; ++r) srow[r] = snn::conv_pack_row(s_src + (((size_t)b * Cin + ci) * H + r) * Wd, Wd, &multi);
            for (int r = 0; r < Cout * OH; ++r) trow[r] = snn::conv_pack_row(s_tgt + ((size_t)b * Cout * OH + r) * OW, OW, &multi);
            multi_any |= multi;
            for (int e = 0; e < Cout * KK; ++e) {             // = one thread
                const int co = e / KK, kk = e - co * KK, ky = kk / KW, kx = kk - ky * KW;
                const uint8_t *ss = s_src + ((size_t)b * Cin + ci) * H * Wd, *st = s_tgt + ((size_t)b * Cout + co) * OH * OW;
                const float *xs = x_src + ((size_t)b * Cin + ci) * H * Wd, *xt = x_tgt + ((size_t)b * Cout + co) * OH * OW;
                float a, p;
                if (use_events == 2 && !multi && stride == 1) {   // the list form of k_convpp_run: one list per image, every element walks it
                    std::vector<uint16_t> evS((size_t)H * 32 + 8), evT((size_t)OH * 32 + 8);
                    const int nS = snn::conv_event_list(srow.data(), H, evS.data());
                    const int nT = snn::conv_event_list(trow.data() + (size_t)co * OH, OH, evT.data());
                    a = snn::conv_pp_list_sum(evS.data(), nS, xt, pad - ky, pad - kx, OW, OH);
                    p = snn::conv_pp_list_sum(evT.data(), nT, xs, ky - pad, kx - pad, Wd, H);
                } else
                if (use_events && !multi) snn::conv_pp_events(g, ky, kx, srow.data(), trow.data() + (size_t)co * OH, xs, xt, &a, &p);
                else snn::conv_pp_dense(g, ky, kx, ss, xs, st, xt, &a, &p);
                const long id = (long)b * E + (long)co * K + (long)ci * KK + kk;
                part[id] = a;
                part[(size_t)B * E + id] = p;
            }
        }
    return multi_any;
}

// conv_bench.hip -- stand-alone timing of k_conv3x3 (mz_conv.h) on synthetic data, with diagnostic variants selected at
// compile time (-DMZC_NO_FETCH / -DMZC_NO_XS / -DMZC_NO_STORE / -DMZC_NO_EPI remove one phase each; their outputs are wrong,
// only the time matters).  Build + run: tools/micro/run_conv_bench.sh (on the GPU box).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../muzero_amd/csrc/mz_convnet.h"

using namespace mz;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, C = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 15, W = argc > 4 ? atoi(argv[4]) : 15;
    const int cin = argc > 5 ? atoi(argv[5]) : C, reps = argc > 6 ? atoi(argv[6]) : 20, residual = argc > 7 ? atoi(argv[7]) : 0;
    const int n_cb = (cin + 15) / 16, co_tiles = (C + 15) / 16;
    const size_t in_n = (size_t)B * cin * H * W, out_n = (size_t)B * C * H * W, w_n = (size_t)co_tiles * n_cb * 9 * 256;
    std::vector<float> h_in(in_n), h_w(w_n), h_b(co_tiles * 16, 0.1f);
    for (size_t i = 0; i < in_n; i++) h_in[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.0f;
    for (size_t i = 0; i < w_n; i++) h_w[i] = (float)((i * 40503u) >> 7 & 255) / 65536.0f - 0.001f;
    float *d_in, *d_out, *d_res, *d_w, *d_b;
    CK(hipMalloc(&d_in, in_n * 4)); CK(hipMalloc(&d_out, out_n * 4)); CK(hipMalloc(&d_res, out_n * 4)); CK(hipMalloc(&d_w, w_n * 4)); CK(hipMalloc(&d_b, h_b.size() * 4));
    CK(hipMemcpy(d_in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), w_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), h_b.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_res, 0, out_n * 4));
    ConvLayerDev L;
    L.w = d_w; L.b = d_b; L.cin = cin; L.cin_real = cin; L.cout = C; L.stride = 1;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) conv_run(st, L, B, d_in, nullptr, nullptr, 0, H, W, residual ? d_res : nullptr, d_out, true);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) conv_run(st, L, B, d_in, nullptr, nullptr, 0, H, W, residual ? d_res : nullptr, d_out, true);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, flop = 2.0 * B * C * (double)cin * 9 * H * W;
    const ConvGeom g = conv_geometry(B, H, W, 1, C);
#ifdef MZC_STAMPS
    {
        long long* d_st;
        CK(hipMalloc(&d_st, 64));
        CK(hipMemset(d_st, 0, 64));
        g_conv_stamps = d_st;
        conv_run(st, L, B, d_in, nullptr, nullptr, 0, H, W, residual ? d_res : nullptr, d_out, true);
        CK(hipStreamSynchronize(st));
        long long h[8];
        CK(hipMemcpy(h, d_st, 64, hipMemcpyDeviceToHost));
        long long tot = 0;
        for (int i = 0; i < 6; i++) tot += h[i];
        printf("  stamps (cycles, wave 0 of WG 0): prologue %lld | per cb: fetch-issue+first-B %lld, taps %lld, store %lld, barrier %lld | epilogue %lld | total %lld\n", h[0],
               h[1] / n_cb, h[2] / n_cb, h[3] / n_cb, h[4] / n_cb, h[5], tot);
    }
#endif
    printf("B=%d C=%d cin=%d %dx%d G=%d npt=%d nct=%d res=%d : %8.1f us  %6.1f TFLOP/s  (%.1f%% of 157.3)\n", B, C, cin, H, W, g.G, g.npt, g.nct, residual, us,
           flop / us / 1e6, 100.0 * flop / us / 1e6 / 157.3);
    return 0;
}

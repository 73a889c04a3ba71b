// conv_bench.hip -- stand-alone timing of k_conv3x3 (mz_conv.h) on synthetic data, with diagnostic variants selected at
// compile time (-DMZC_NO_FETCH / -DMZC_NO_XS / -DMZC_NO_STORE / -DMZC_NO_EPI remove one phase each; their outputs are wrong,
// only the time matters).  Build + run: tools/micro/run_conv_bench.sh (on the GPU box).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../muzero_amd/csrc/mz_convnet.h"

using namespace mz;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// tower mode: conv_bench tower B P H W R -> times k_res_tower (R residual blocks) against R * 2 separate conv launches
static int tower_main(int argc, char** argv) {
    const int B = atoi(argv[2]), P = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), R = atoi(argv[6]);
    const int n_cb = P / 16, co_tiles = P / 16, reps = 10;
    const size_t act_n = (size_t)B * P * H * W, w_n = (size_t)co_tiles * n_cb * 9 * 256;
    std::vector<float> h_in(act_n), h_w(w_n), h_b(P, 0.01f);
    for (size_t i = 0; i < act_n; i++) h_in[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.0f;
    for (size_t i = 0; i < w_n; i++) h_w[i] = (float)((i * 40503u) >> 7 & 255) / 65536.0f - 0.0018f;
    float *d_x, *d_t1, *d_t2, *d_w, *d_b;
    CK(hipMalloc(&d_x, act_n * 4 + 256)); CK(hipMalloc(&d_t1, act_n * 4 + 256)); CK(hipMalloc(&d_t2, act_n * 4 + 256));
    CK(hipMalloc(&d_w, w_n * 4)); CK(hipMalloc(&d_b, P * 4));
    CK(hipMemcpy(d_x, h_in.data(), act_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), w_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, h_b.data(), P * 4, hipMemcpyHostToDevice));
    std::vector<ResBlockDev> blocks(R);
    for (int i = 0; i < R; i++)
        for (ConvLayerDev* c : {&blocks[i].c1, &blocks[i].c2}) { c->w = d_w; c->b = d_b; c->cin = c->cin_real = c->cout = P; c->stride = 1; }
    float *tw, *tb;
    CK(hipMalloc(&tw, (size_t)2 * R * w_n * 4)); CK(hipMalloc(&tb, (size_t)2 * R * P * 4));
    for (int i = 0; i < 2 * R; i++) {
        CK(hipMemcpy(tw + (size_t)i * w_n, d_w, w_n * 4, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(tb + (size_t)i * P, d_b, P * 4, hipMemcpyDeviceToDevice));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * R * 2.0 * B * P * (double)P * 9 * H * W;
    for (int fused = 1; fused >= 0; fused--) {
        for (int i = 0; i < 2; i++) tower_run(st, blocks, 0, R, B, d_x, d_t1, d_t2, H, W, fused ? tw : nullptr, fused ? tb : nullptr);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) tower_run(st, blocks, 0, R, B, d_x, d_t1, d_t2, H, W, fused ? tw : nullptr, fused ? tb : nullptr);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        const TowerGeom g = tower_geometry(B, P, H, W);
        printf("tower B=%d P=%d %dx%d R=%d %s (G=%d npt=%d): %9.1f us  %6.1f us/conv  %6.1f TFLOP/s (%.1f%% of 157.3)\n", B, P, H, W, R,
               fused ? "fused   " : "separate", g.G, g.npt, us, us / (2 * R), flop / us / 1e6, 100.0 * flop / us / 1e6 / 157.3);
    }
#ifdef MZC_STAMPS
    {
        long long* d_st;
        CK(hipMalloc(&d_st, 64));
        CK(hipMemset(d_st, 0, 64));
        const TowerGeom g = tower_geometry(B, P, H, W);
        TowerLaunch L{};
        L.in = d_x; L.out = d_t1; L.w = tw; L.bias = tb; L.n_convs = 2 * R; L.P = P; L.h = H; L.w_img = W; L.G = g.G; L.B = B; L.nposp = g.nposp; L.stamps = d_st;
        if (g.npt == 5) tower_launch_npt<5>(st, L, (B + g.G - 1) / g.G, g.lds);
        CK(hipStreamSynchronize(st));
        long long h[8];
        CK(hipMemcpy(h, d_st, 64, hipMemcpyDeviceToHost));
        const int nc = 2 * R;
        printf("  tower stamps (cycles, wave 0 of WG 0): prologue %lld | per conv: block-start reads %lld, taps %lld, epilogue %lld, barrier %lld\n", h[0], h[1] / nc,
               h[2] / nc, h[3] / nc, h[4] / nc);
    }
#endif
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 6 && std::string(argv[1]) == "tower") return tower_main(argc, argv);
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

// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 with NACC independent accumulators, operands in registers,
// one wave per SIMD (256 threads per block, 256 blocks).  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < NACC; j++) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(float* out, long long* cyc, int blocks) {
    const int iters = 200;
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; i++) m += h[i];
    m /= blocks;
    printf("NACC=%d blocks=%d: %.1f cycles per MFMA (per wave)\n", NACC, blocks, m / (iters * 4.0 * NACC));
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int blocks : {1, 256, 512}) { run<1>(out, cyc, blocks); run<2>(out, cyc, blocks); run<8>(out, cyc, blocks); }
    return 0;
}

// Microbenchmark: does independent VALU work ride for free in the shadow of a LONE wave's MFMA stream?
// One wave per SIMD (256 threads per block, 1 block per CU through a 100 KiB LDS allocation), 8 independent accumulators,
// NV VALU instructions of kind KIND between consecutive MFMAs (inline asm, so nothing is folded away; s_nop-free by
// construction: the VALU registers are never touched by an MFMA).
//   KIND 0: v_fma_f32 on independent registers    1: a DEPENDENT v_fma_f32 chain    2: v_exp_f32 (transcendental)
//   KIND 3: v_mov_b32 with a DPP row_shr (cross-lane)    4: v_fma_f64 (dependent chain)   5: ds_read_b32 + s_waitcnt at the end of the group
// hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void valu(float (&v)[8], double& d, int i, const float* lds) {
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 3) & 7] ));
    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[0]));
    if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
    if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i & 7]) : "v"(v[(i + 3) & 7]));
    if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d));
    if (KIND == 5) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((int)(threadIdx.x * 4 + (i & 7) * 1024))); v[i & 7] = t; }
}

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[8];
    for (int j = 0; j < 8; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 1e-4f + j;
    double d = threadIdx.x * 1e-3;
    lds[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
                for (int n = 0; n < NV; n++) valu<KIND>(v, d, j * NV + n, lds);
            }
        }
        if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    float s = (float)d;
    for (int j = 0; j < 8; j++) s += acc[j][0] + acc[j][3] + v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND>
void run(float* out, long long* cyc) {
    const int iters = 100, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NV, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 100 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; i++) m += h[i];
    m /= blocks;
    printf("KIND=%d NV=%d: %.1f cycles per MFMA (+%d VALU each)\n", KIND, NV, m / (iters * 32.0), NV);
}
#define ROW(K) run<0, K>(out, cyc); run<1, K>(out, cyc); run<2, K>(out, cyc); run<4, K>(out, cyc); run<6, K>(out, cyc); run<8, K>(out, cyc);
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    return 0;
}

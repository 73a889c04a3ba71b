// Microbenchmark (round 6, follow-up of issue.hip): why does a VALU instruction of the search kernel's tree phases cost ~9 cycles when
// the same instruction costs 4.1 in issue.hip?  Same straight-line dependent v_add_f32 block under kernel-like conditions, one at a time:
//   A  baseline: 256 threads, low registers            B  + a register allocation of 240 VGPRs (clobber of v239)
//   C  + operands in high registers (v200..)           D  512 threads, waves 4-7 parked at a barrier, 240 VGPRs
//   E  D + 150 KiB of LDS + amdgpu_waves_per_eu(2, 2)   F  E, after a burst of 64 MFMAs (does the MFMA -> VALU transition cost anything?)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue2.hip -o tools/micro/issue2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void body(float* out, long long* cyc, int iters) {
    float v0 = threadIdx.x * 1e-3f, v4 = 4.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long t0 = __builtin_readcyclecounter(), tm = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 5) {
            long long a = __builtin_readcyclecounter();
#pragma unroll
            for (int r = 0; r < 64; r++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(v4), "v"(v4));
            asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
            tm += __builtin_readcyclecounter() - a;
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) asm volatile(REP64("v_add_f32_e32 v200, v200, v201\n") ::: "v200", "v201");
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) asm volatile(REP64("v_add_f32_e32 %0, %0, %1\n") : "+v"(v0) : "v"(v4));
        }
        if (MODE >= 1) asm volatile("" ::: "v239");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + (threadIdx.x & 255)] = v0 + acc[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0 - tm;
}
template <int MODE>
__global__ __launch_bounds__(256) void k256(float* out, long long* cyc, int iters) { body<MODE>(out, cyc, iters); }
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k512(float* out, long long* cyc, int iters) {
    extern __shared__ int lds[];
    if (threadIdx.x >= 256) { __syncthreads(); return; }
    if (iters < 0) lds[threadIdx.x] = 1;
    body<MODE>(out, cyc, iters);
    __syncthreads();
}
static void report(const char* name, long long* cyc) {
    long long h[256];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; i++) m += h[i];
    printf("%-90s %6.2f cycles per v_add_f32\n", name, m / 256 / (50 * 1024.0));
    fflush(stdout);
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k256<0>), dim3(256), dim3(256), 0, 0, out, cyc, 50);
    report("A 256 threads, low registers", cyc);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k256<1>), dim3(256), dim3(256), 0, 0, out, cyc, 50);
    report("B + allocation of 240 VGPRs", cyc);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k256<2>), dim3(256), dim3(256), 0, 0, out, cyc, 50);
    report("C + operands v200 / v201", cyc);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k512<1>), dim3(256), dim3(512), 0, 0, out, cyc, 50);
    report("D 512 threads (waves 4-7 parked at the barrier), 240 VGPRs", cyc);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k512<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k512<3>), dim3(256), dim3(512), 150 * 1024, 0, out, cyc, 50);
    report("E D + 150 KiB LDS", cyc);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k512<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k512<5>), dim3(256), dim3(512), 150 * 1024, 0, out, cyc, 50);
    report("F E, each block of 1024 adds after a burst of 64 MFMAs (burst time subtracted)", cyc);
    return 0;
}

// Microbenchmark (round 6): what does ONE instruction of the tree phases cost a wave that runs alone on its SIMD?
// The search kernels' select / backup run with one wave per SIMD (the workgroup owns the CU's LDS), so nothing hides issue gaps or
// latencies.  Every pattern is a block of REP inline-asm instructions inside a loop, timed with the cycle counter; printed: cycles per
// instruction.  256 threads per block, one block per CU (100 KiB of LDS), 256 blocks.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/issue.hip -o tools/micro/issue && tools/micro/issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

enum {
    K_VALU_IND, K_VALU_DEP, K_F64_IND, K_F64_DEP, K_SALU_DEP, K_CNDMASK_VCC, K_CMP_BRANCH_NT, K_DPP_DEP, K_DPP_IND, K_LDS_CHAIN128, K_LDS_CHAIN32,
    K_LDS_IND128, K_CVT_DEP, K_READLANE, K_CMP_SALU_VALU, K_BRANCH_TAKEN, K_VALU_PAIRS, K_F64_MULADD_DEP, K_RCP_DEP, K_BCAST64_IND, K_CHAIN_OLD, K_CHAIN_NEW, K_STRAIGHT4, K_STRAIGHT8, K_STRAIGHT12, K_COUNT
};
static const char* NAMES[] = {"v_add_f32 independent", "v_add_f32 dependent chain", "v_fma_f64 independent", "v_fma_f64 dependent chain", "s_add_u32 dependent",
                              "v_cmp -> v_cndmask (vcc) pairs, per pair", "v_cmp + s_cbranch_vccnz (not taken), per pair", "v_mov_dpp row_shr dependent chain",
                              "v_mov_dpp row_shr independent", "ds_read_b128 dependent chain (address = loaded word)", "ds_read_b32 dependent chain",
                              "ds_read_b128 independent (16 in flight)", "v_cvt_f32_f64 -> v_cvt_f64_f32 dependent, per instruction", "v_readlane_b32 -> s_add dependent, per pair",
                              "v_cmp -> s_and_b64 -> v_cndmask, per triple", "taken s_branch (per branch)", "two interleaved dependent v_add chains, per instruction",
                              "v_mul_f64 -> v_add_f64 dependent, per instruction", "v_rcp_f32 dependent chain",
                              "v_mov_b64_dpp row_newbcast independent", "value-chain level, round 5 form (2 mov, nop, 2 dpp, mul, add, 2 cndmask), per level",
                              "value-chain level, round 6 form (bcast64, mul, add, 2 cndmask), per level",
                              "1024 straight-line v_add_f32, 4-byte encoding (e32), per instruction", "1024 straight-line v_add_f32, 8-byte encoding (e64), per instruction",
                              "1024 straight-line v_add_f32 with a 32-bit literal (8 bytes, e32 + literal), per instruction"};

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    extern __shared__ int lds[];
    if (threadIdx.x >= 256) { __syncthreads(); __syncthreads(); return; }  // PARKED mode (512 threads): a second wave per SIMD asleep at the barrier
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = ((i * 16) & 0x3ff0);  // a word that is a valid 16-byte aligned LDS byte address
    __syncthreads();
    float v0 = threadIdx.x * 1e-3f, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    double d0 = threadIdx.x * 1e-3, d1 = 1.0, d2 = 2.0, d3 = 3.0, dk = 0.999;
    int s0 = 1, addr = (threadIdx.x * 16) & 0x3ff0;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 q = {0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == K_VALU_IND) asm volatile(REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (KIND == K_VALU_DEP) asm volatile(REP16("v_add_f32 %0, %0, %1\n") : "+v"(v0) : "v"(v4));
        if (KIND == K_F64_IND) asm volatile(REP4("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0));
        if (KIND == K_F64_DEP) asm volatile(REP16("v_fma_f64 %0, %0, %1, %1\n") : "+v"(d0) : "v"(1.0));
        if (KIND == K_SALU_DEP) asm volatile(REP16("s_add_u32 %0, %0, 1\n") : "+s"(s0) : : "scc");
        if (KIND == K_CNDMASK_VCC) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(v0) : "v"(v4) : "vcc");
        if (KIND == K_CMP_BRANCH_NT) asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %0\n s_cbranch_vccnz 1f\n 1:\n") : "+v"(v0) : : "vcc");
        if (KIND == K_DPP_DEP) asm volatile(REP16("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v0));
        if (KIND == K_DPP_IND) asm volatile(REP4("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (KIND == K_LDS_CHAIN128) {
#pragma unroll
            for (int r = 0; r < 16; r++) { asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr) : "memory"); addr = q[0]; }
        }
        if (KIND == K_LDS_CHAIN32) asm volatile(REP16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(addr) : : "memory");
        if (KIND == K_LDS_IND128) asm volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr) : "memory");
        if (KIND == K_CVT_DEP) asm volatile(REP16("v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n") : "+v"(d0), "+v"(v0));
        if (KIND == K_READLANE) asm volatile(REP16("v_readlane_b32 %1, %0, 3\n s_add_u32 %1, %1, 1\n") : "+v"(v0), "+s"(s0) : : "scc");
        if (KIND == K_CMP_SALU_VALU) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(v0) : "v"(v4) : "vcc");
        if (KIND == K_BRANCH_TAKEN) asm volatile(REP16("s_branch 1f\n s_nop 0\n 1:\n"));
        if (KIND == K_VALU_PAIRS) asm volatile(REP16("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n") : "+v"(v0), "+v"(v1) : "v"(v4));
        if (KIND == K_F64_MULADD_DEP) asm volatile(REP16("v_mul_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n") : "+v"(d0) : "v"(1.0));
        if (KIND == K_RCP_DEP) asm volatile(REP16("v_rcp_f32 %0, %0\n") : "+v"(v0));
        if (KIND == K_STRAIGHT4) _Pragma("unroll") for (int r = 0; r < 16; r++) asm volatile(REP64("v_add_f32_e32 %0, %0, %1\n") : "+v"(v0) : "v"(v4));
        if (KIND == K_STRAIGHT8) _Pragma("unroll") for (int r = 0; r < 16; r++) asm volatile(REP64("v_add_f32_e64 %0, %0, %1\n") : "+v"(v0) : "v"(v4));
        if (KIND == K_STRAIGHT12) _Pragma("unroll") for (int r = 0; r < 16; r++) asm volatile(REP64("v_add_f32_e32 %0, 0x3f800123, %0\n") : "+v"(v0));
        if (KIND == K_BCAST64_IND) asm volatile("s_nop 1\n" REP4("v_mov_b64_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n") : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(dk));
        if (KIND == K_CHAIN_OLD) asm volatile(REP16("v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n s_nop 0\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %3, %4, %3\n v_add_f64 %3, %3, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %5, %5, %2, vcc\n") : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(d0), "+v"(dk), "+v"(v3) : : "vcc");
        if (KIND == K_CHAIN_NEW) asm volatile(REP16("v_mov_b64_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f64 %0, %2, %0\n v_add_f64 %0, %0, %1\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %3, vcc\n") : "+v"(d0), "+v"(d1), "+v"(dk), "+v"(v0), "+v"(v1) : : "vcc");
    }
    long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v5 + v6 + v7 + (float)(d0 + d1 + d2 + d3) + s0 + addr + q[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_threads = 256;
template <int KIND>
void run(float* out, long long* cyc) {
    const int iters = 200, blocks = 256;
    printf("[%d] ", KIND); fflush(stdout);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(g_threads), 100 * 1024, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; i++) m += h[i];
    m /= blocks;
    const int per = (KIND == K_STRAIGHT4 || KIND == K_STRAIGHT8 || KIND == K_STRAIGHT12) ? 1024 : (KIND == K_CVT_DEP || KIND == K_VALU_PAIRS || KIND == K_F64_MULADD_DEP) ? 32 : 16;
    printf("%-70s %7.1f cycles\n", NAMES[KIND], m / (iters * (double)per));
    fflush(stdout);
}
template <int K0>
void run_all(float* out, long long* cyc) {
    if constexpr (K0 < K_COUNT) { if (K0 != K_CMP_SALU_VALU) run<K0>(out, cyc); run_all<K0 + 1>(out, cyc); }  // (the v_cmp -> s_and vcc -> v_cndmask pattern hangs as inline asm: skipped)
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    printf("one wave per SIMD:\n");
    run_all<0>(out, cyc);
    g_threads = 512;
    printf("one working wave per SIMD + one wave parked at s_barrier:\n");
    run_all<0>(out, cyc);
    return 0;
}

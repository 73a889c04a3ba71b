// Diagnostic micro-benchmark: what does one level of an LDS pointer chase cost a wave that runs alone on its SIMD?
//   (a) bare dependent ds_read_b64 chain;  (b) the phase-A loop body of tree2_select (epoch / margin test, any, selects).
// hipcc --offload-arch=gfx950 -O3 -o chase tools/micro/chase.hip && ./chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct __attribute__((aligned(8))) Cache { int packed; float t; };

__device__ __forceinline__ long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (long long)t;
}

__global__ void k_chase(const Cache* g, int n_nodes, int iters, long long* out, int* sink) {
    __shared__ Cache c[1024];
    for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) c[i] = g[i];
    __syncthreads();
    const int a0 = threadIdx.x & 15;
    // (a) bare chain
    int n = threadIdx.x >> 4;
    long long t0 = now();
    for (int i = 0; i < iters; i++) n = c[n].packed >> 16;
    long long t1 = now();
    // (b) phase-A body
    int m = threadIdx.x >> 4, k = 0, lp = 0, la = 0, mypath = 0;
    bool done = false;
    const float thr = 1e-4f;
    const int ep = 0;
    int it = 0;
    for (;;) {
        const Cache cc = c[m];
        const bool adv = !done & (((cc.packed >> 8) & 0xff) == ep) & (thr < cc.t);
        if (!__any(adv)) break;
        mypath = (adv & (a0 == k)) ? m : mypath;
        const int ch = cc.packed >> 16, k1 = k + 1;
        const bool stop = adv & ((ch < 0) | (k1 > iters));
        lp = stop ? m : lp;
        la = stop ? (cc.packed & 0xff) : la;
        m = (adv & !stop) ? ch : m;
        k = adv ? k1 : k;
        done |= stop;
        it++;
    }
    long long t2 = now();
    // (c) lean body: stopped segments park on a sentinel node whose margin is -inf (no `done` mask), the path goes to LDS
    // unconditionally (idempotent), exit test straight on the compare's lane mask
    __shared__ short path[16][72];
    const int SENT = 1023;
    int m2 = threadIdx.x >> 4, k2 = 0, lp2 = 0, la2 = 0, it2 = 0;
    short* prow = path[threadIdx.x >> 4];
    for (;;) {
        const Cache cc = c[m2];
        const bool adv = thr < cc.t;
        if (__builtin_amdgcn_ballot_w64(adv) == 0) break;
        if (a0 == 0) prow[k2] = (short)m2;
        const int ch = cc.packed >> 16;
        const bool stop = adv & (ch < 0);
        lp2 = stop ? m2 : lp2;
        la2 = stop ? (cc.packed & 0xff) : la2;
        m2 = adv ? (ch < 0 ? SENT : ch) : m2;
        k2 += adv ? 1 : 0;
        it2++;
        if (it2 >= iters) c[m2].t = -1.0f;  // benchmark only: end the chase
    }
    long long t3 = now();
    // (d) software-pipelined lean body: the next level's cache entry is requested before the exit test resolves
    for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) c[i] = g[i];
    __syncthreads();
    long long t4 = now();
    int m3 = threadIdx.x >> 4, k3 = 0, lp3 = 0, la3 = 0, it3 = 0;
    Cache cc3 = c[m3];
    for (;;) {
        const bool adv = thr < cc3.t;
        const int ch = cc3.packed >> 16;
        const bool stop = adv & (ch < 0);
        const int nxt = adv ? (ch < 0 ? SENT : ch) : m3;
        const Cache nc = c[nxt];
        if (a0 == 0) prow[k3] = (short)m3;
        lp3 = stop ? m3 : lp3;
        la3 = stop ? (cc3.packed & 0xff) : la3;
        k3 += adv ? 1 : 0;
        it3++;
        if (it3 >= iters) c[nxt].t = -1.0f;  // benchmark only (takes effect one level later)
        if (__builtin_amdgcn_ballot_w64(adv) == 0) break;
        m3 = nxt;
        cc3 = nc;
    }
    long long t5 = now();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = it; out[3] = t3 - t2; out[4] = it2; out[5] = t5 - t4; out[6] = it3; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = n + lp + la + mypath + k + lp2 + la2 + k2 + prow[3] + lp3 + la3 + k3 + m3;
}

int main() {
    const int N = 1024, iters = 64;
    std::vector<Cache> h(N);
    for (int i = 0; i < N; i++) { h[i].packed = (((i * 7 + 13) % N) << 16) | (0 << 8) | (i & 1); h[i].t = 1.0f; }
    Cache* d; long long* out; int* sink;
    hipMalloc(&d, N * sizeof(Cache)); hipMalloc(&out, 64); hipMalloc(&sink, 256 * 256 * 4);
    hipMemcpy(d, h.data(), N * sizeof(Cache), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_chase, dim3(256), dim3(256), 0, 0, d, N, iters, out, sink);
    long long r[7];
    hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    printf("bare chain: %.1f cycles per level; phase-A body: %.1f cycles per level (%lld levels); lean body: %.1f (%lld levels)\n", (double)r[0] / iters,
           (double)r[1] / (double)r[2], r[2], (double)r[3] / (double)r[4], r[4]);
    printf("pipelined lean body: %.1f cycles per level (%lld levels)\n", (double)r[5] / (double)r[6], r[6]);
    return 0;
}

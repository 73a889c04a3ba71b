// learner_replay.hip -- device-side replay sampling and priority updates (replay.py:81-113 of the reference) for a replay ring whose
// bookkeeping lives in HBM (muzero_amd.replay.PrioritizedReplay with a device writer attached): the learner draws its batch and writes the new
// priorities back without the host ever reading the counter or the priority array.  Part of libmzlearner_hip.so (include/mzlearner.h).
//
//   uniform (priority_exponent == 0, every launcher's default; replay.py:87-89):  index = floor(u * size), u ~ Philox, weights 1
//   proportional (replay.py:90-98): w_i = priority_i ^ alpha, inverse-CDF picks on the inclusive prefix sums (float64; what
//       np.random.choice(p = w / sum w) does with its own uniforms: searchsorted(cumsum, u, side='right')), importance weights
//       ((1 / size) / (w_i / sum w)) ^ beta, divided by their maximum over the batch
//   update (replay.py:106-113): priority[index[b]] = new[b]; of a repeated index the LAST b wins, as in the reference's loop
// Randomness: Philox4x32-10 keyed by (seed; draw number, sample) -- the stream the planner's device randomness uses (mz_device.h).  Everything is
// enqueued on the caller's stream; nothing synchronises.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/mzlearner.h"
#include "mz_device.h"

namespace {

constexpr int RB = 1024;  // elements per scan block

__device__ __forceinline__ long long live_size(const int64_t* num_added, long long capacity) {
    const long long n = *num_added;
    return n < capacity ? n : capacity;
}

// error counters (mzl_replay_set_error_counters; may be null): [0] draws from an EMPTY replay (the reference raises, replay.py:83-84; here
// the draw returns slot 0 and is counted), [1] priorities the reference's update would have raised on (replay.py:106-110: non-finite or
// negative; here they are skipped and counted)
__global__ __launch_bounds__(256) void k_rp_uniform(const int64_t* num_added, long long capacity, unsigned long long seed, unsigned long long draw, int batch,
                                                     int64_t* index, float* weights, int* errors) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const long long n = live_size(num_added, capacity);
    mz::Philox g(seed, (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)b);
    long long i = (long long)(g.uniform() * (double)n);  // np.uniform(0, n).astype(int64): truncation
    i = i >= n ? n - 1 : i;
    i = i < 0 ? 0 : i;  // (n == 0: slot 0, counted below -- never an index in front of the ring)
    if (n < 1 && b == 0 && errors) atomicAdd(&errors[0], 1);
    index[b] = i;
    if (weights) weights[b] = 1.0f;
}

// ONE reading of the item count per draw: the planner's stream advances *num_added while the scan, the pick and the importance weights
// (1 / n) run -- every kernel of a proportional draw uses this snapshot
__global__ void k_rp_snapshot(const int64_t* num_added, long long capacity, long long* snap, int* errors) {
    const long long n = live_size(num_added, capacity);
    *snap = n;
    if (n < 1 && errors) atomicAdd(&errors[0], 1);
}

// block-local inclusive prefix sums of w_i = priority_i ^ alpha (float64), block totals
__global__ __launch_bounds__(256) void k_rp_scan_blocks(const float* prio, const long long* snap, long long capacity, double alpha, double* cdf, double* block_sum) {
    __shared__ double s_part[256];
    const long long n = *snap;
    const long long base = (long long)blockIdx.x * RB + (long long)threadIdx.x * 4;
    double w[4], run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + k;
        w[k] = i < n ? pow((double)prio[i], alpha) : 0.0;
        run += w[k];
        w[k] = run;
    }
    s_part[threadIdx.x] = run;
    __syncthreads();
    // Hillis-Steele over the 256 thread totals (fixed order: bit-reproducible)
    for (int off = 1; off < 256; off <<= 1) {
        const double add = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0.0;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    const double before = threadIdx.x ? s_part[threadIdx.x - 1] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (base + k < capacity) cdf[base + k] = before + w[k];
    if (threadIdx.x == 255) block_sum[blockIdx.x] = s_part[255];
}

// prefix sums of the block totals by one workgroup (any number of blocks: chunks of 1024 with a running carry): block_off[i] = sum of blocks
// < i, block_off[nb] = total.  Every entry is the INCLUSIVE running sum of its predecessor block (round 6, ADVICE r5: the exclusive form
// `carry + inclusive - v` was neither monotone in floating point nor consistent with the total, so a draw could land in an empty
// trailing block)
__global__ __launch_bounds__(1024) void k_rp_scan_totals(const double* block_sum, double* block_off, int nb) {
    __shared__ double s[1024];
    __shared__ double s_carry;
    if (threadIdx.x == 0) s_carry = 0.0;
    __syncthreads();
    for (int c0 = 0; c0 < nb; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const double v = i < nb ? block_sum[i] : 0.0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const double add = threadIdx.x >= off ? s[threadIdx.x - off] : 0.0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        const double carry = s_carry;
        if (i < nb) block_off[i + 1] = carry + s[threadIdx.x];  // non-decreasing: sums of non-negative terms, added in one fixed order
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_off[0] = 0.0;  // (block_off[nb] was written by the last block's thread: the same sum the picks compare with)
}

__global__ __launch_bounds__(256) void k_rp_pick(const float* prio, const long long* snap, long long capacity, double alpha, double beta, const double* cdf,
                                                  const double* block_off, int nb, unsigned long long seed, unsigned long long draw, int batch, int64_t* index,
                                                  float* weights) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const long long n = *snap;
    const double total = block_off[nb];
    mz::Philox g(seed, (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)b);
    const double u = g.uniform() * total;
    // first block whose inclusive end exceeds u (searchsorted side='right' on the global inclusive prefix sums)
    int lo = 0, hi = nb - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (block_off[mid + 1] > u) hi = mid; else lo = mid + 1;
    }
    const double off = block_off[lo];
    long long a = (long long)lo * RB, e = a + RB - 1;
    if (e > n - 1) e = n - 1;
    while (a < e) {
        const long long mid = (a + e) >> 1;
        if (off + cdf[mid] > u) e = mid; else a = mid + 1;
    }
    // never an unwritten slot: inside [0, n - 1], and -- should rounding have carried u past the last item with weight -- back to the nearest
    // slot that has one (np.random.choice cannot return an entry of probability 0)
    a = a > n - 1 ? n - 1 : a;
    a = a < 0 ? 0 : a;
    double w = pow((double)prio[a], alpha);
    while (!(w > 0.0) && a > 0) { a--; w = pow((double)prio[a], alpha); }
    index[b] = a;
    const double p = w / total;
    weights[b] = (p > 0.0 && n > 0) ? (float)pow((1.0 / (double)n) / p, beta) : 0.0f;  // normalised by the batch maximum in k_rp_normalize
}

__global__ __launch_bounds__(1024) void k_rp_normalize(float* weights, int batch) {
    __shared__ float s[1024];
    float m = 0.0f;
    for (int i = threadIdx.x; i < batch; i += 1024) m = weights[i] > m ? weights[i] : m;
    s[threadIdx.x] = m;
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] = s[threadIdx.x + off] > s[threadIdx.x] ? s[threadIdx.x + off] : s[threadIdx.x];
        __syncthreads();
    }
    m = s[0];
    for (int i = threadIdx.x; i < batch; i += 1024) weights[i] = m > 0.0f ? weights[i] / m : 1.0f;  // (all priorities zero / empty replay: plain weights, not 0 / 0)
}

// priority[index[b]] = value[b], the last b of a repeated index wins: the owner array records the largest b per touched slot (max is order-free)
__global__ __launch_bounds__(256) void k_rp_owner_clear(const int64_t* index, int batch, int* owner) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) owner[index[b]] = -1;
}
__global__ __launch_bounds__(256) void k_rp_owner_max(const int64_t* index, int batch, int* owner) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) atomicMax(&owner[index[b]], b);
}
__global__ __launch_bounds__(256) void k_rp_scatter(const int64_t* index, const float* value, int batch, const int* owner, float* prio, int* errors) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float v = value[b];
    // replay.py:106-110 raises on a non-finite or negative priority before writing anything; the device cannot raise: the value is not
    // written (one NaN would poison every later prefix sum and importance weight) and the caller reads the count at its next sync point
    const bool bad = !(v >= 0.0f) || v > 3.0e38f;
    if (bad) { if (errors) atomicAdd(&errors[1], 1); return; }
    if (owner[index[b]] == b) prio[index[b]] = v;
}

}  // namespace

void mzl_internal_set_error(const std::string& msg);  // learner.hip: the text mzl_last_error() returns
#define g_rerr_set(m) mzl_internal_set_error(m)

static int* g_replay_errors = nullptr;  // device int32[2] or null (mzl_replay_set_error_counters)
extern "C" int mzl_replay_set_error_counters(int32_t* d_counters) {
    g_replay_errors = d_counters;
    return MZL_OK;
}

extern "C" int64_t mzl_replay_scratch_doubles(int64_t capacity) {
    const int64_t nb = (capacity + RB - 1) / RB;
    return capacity + 2 * nb + 8;
}

extern "C" int mzl_replay_sample(const mzl_replay_draw* d, void* stream) {
    if (!d || !d->d_num_added || !d->d_index || d->capacity < 1 || d->batch < 1) { g_rerr_set("mzl_replay_sample: bad arguments"); return MZL_E_INVALID; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int gb = (d->batch + 255) / 256;
    if (d->priority_exponent == 0.0) {
        hipLaunchKernelGGL(k_rp_uniform, dim3(gb), dim3(256), 0, st, d->d_num_added, (long long)d->capacity, (unsigned long long)d->seed,
                           (unsigned long long)d->draw, d->batch, d->d_index, d->d_weights, g_replay_errors);
    } else {
        if (!d->d_priority || !d->d_weights || !d->d_scratch) { g_rerr_set("mzl_replay_sample: proportional draws need d_priority, d_weights and d_scratch"); return MZL_E_INVALID; }
        const int nb = (int)((d->capacity + RB - 1) / RB);
        double* cdf = d->d_scratch;
        double* bsum = cdf + d->capacity;
        double* boff = bsum + nb;
        long long* snap = reinterpret_cast<long long*>(boff + nb + 1);  // (one of the 8 spare slots of mzl_replay_scratch_doubles)
        hipLaunchKernelGGL(k_rp_snapshot, dim3(1), dim3(1), 0, st, d->d_num_added, (long long)d->capacity, snap, g_replay_errors);
        hipLaunchKernelGGL(k_rp_scan_blocks, dim3(nb), dim3(256), 0, st, d->d_priority, snap, (long long)d->capacity, d->priority_exponent, cdf, bsum);
        hipLaunchKernelGGL(k_rp_scan_totals, dim3(1), dim3(1024), 0, st, bsum, boff, nb);
        hipLaunchKernelGGL(k_rp_pick, dim3(gb), dim3(256), 0, st, d->d_priority, snap, (long long)d->capacity, d->priority_exponent,
                           d->importance_sampling_exponent, cdf, boff, nb, (unsigned long long)d->seed, (unsigned long long)d->draw, d->batch, d->d_index, d->d_weights);
        hipLaunchKernelGGL(k_rp_normalize, dim3(1), dim3(1024), 0, st, d->d_weights, d->batch);
    }
    if (hipGetLastError() != hipSuccess) { g_rerr_set("mzl_replay_sample: kernel launch failed"); return MZL_E_HIP; }
    return MZL_OK;
}

extern "C" int mzl_replay_update_priorities(float* d_priority, int64_t capacity, const int64_t* d_index, const float* d_new, int32_t batch, int32_t* d_owner,
                                            void* stream) {
    if (!d_priority || !d_index || !d_new || !d_owner || capacity < 1 || batch < 1) { g_rerr_set("mzl_replay_update_priorities: bad arguments"); return MZL_E_INVALID; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int gb = (batch + 255) / 256;
    hipLaunchKernelGGL(k_rp_owner_clear, dim3(gb), dim3(256), 0, st, d_index, batch, d_owner);
    hipLaunchKernelGGL(k_rp_owner_max, dim3(gb), dim3(256), 0, st, d_index, batch, d_owner);
    hipLaunchKernelGGL(k_rp_scatter, dim3(gb), dim3(256), 0, st, d_index, d_new, batch, d_owner, d_priority, g_replay_errors);
    if (hipGetLastError() != hipSuccess) { g_rerr_set("mzl_replay_update_priorities: kernel launch failed"); return MZL_E_HIP; }
    return MZL_OK;
}

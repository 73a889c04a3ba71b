/*
 * mz_oracle.c -- CPU ORACLE (test infrastructure, see mz_oracle.h).  Plain C11, no dependencies.
 *
 * Restates, function by function, the planning path of michaelnny/muzero.  File:line citations refer to
 * /root/reference/muzero/.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 *
 * Floating-point conventions (the oracle DEFINES the summation order that the HIP kernels reproduce):
 *   - tree statistics are IEEE float64 exactly as the reference's Python floats (mcts.py:129-200);
 *   - network arithmetic is float32; every dot product is one k-ordered fmaf chain whose initial
 *     accumulator is the bias (this is what v_mfma_f32_*_f32 computes, so GPU == oracle bit for bit),
 *     or a fixed tree of such chains where the kernels split a layer (linear_mlp: K-split second layers,
 *     and the one-neuron second layer of an MSE head, which the kernels run on the vector ALUs);
 *   - exp() is mzo_expf below (own polynomial, identical code on the GPU), sqrt and division are IEEE.
 *   The reference (torch CPU kernels) differs from this only by summation order / libm exp, i.e. at
 *   the 1e-6 relative level; tests/test_oracle_nets.py pins that against recorded reference outputs.
 */
#include "mz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================ */
/* math (util.py)                                                                                */
/* ============================================================================================ */

static inline float bits2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* Cephes-style expf: n = rint(x*log2e); r = x - n*ln2 (two-step); degree-5 polynomial; scale by 2^n. */
float mzo_expf(float x) {
    if (x > 88.5f) return INFINITY;
    if (x < -103.5f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int ni = (int)n;
    if (ni < -126) {
        y = y * 5.42101086242752217e-20f; /* 2^-64, exact */
        ni += 64;
    }
    if (ni > 127) {
        y = y * 2.0f;
        ni -= 1;
    }
    return y * bits2f((uint32_t)(ni + 127) << 23);
}

/* util.py:25-28, eps = 1e-3; float32 op order of the torch expression (python scalars become float32). */
float mzo_signed_parabolic(float x) {
    float ax = fabsf(x);
    float t = 1.001f + ax;
    float u = 0.004f * t;
    float v = 1.0f + u;
    float s = sqrtf(v);
    float z = s / 2.0f / 0.001f - 500.0f;
    float sq = z * z;
    float m = sq - 1.0f;
    float sg = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    return sg * m;
}

/* Row reduction used by every softmax / expectation on the path.  The ORDER is part of the oracle's numerical
 * contract (the HIP kernels reproduce it with 16 lanes per row): 16 interleaved partial sums
 * part[j] = v[j] + v[j+16] + v[j+32] + ... (sequential), then a butterfly over the 16 partials with strides
 * 8, 4, 2, 1 (IEEE addition is commutative, so every slot ends with the same total; with descending strides the
 * xor-butterfly equals the rotate-butterfly the GPU runs as DPP row_ror:8/4/2/1). */
static float row_reduce16(const float* v, int n) {
    float part[16];
    for (int j = 0; j < 16; j++) {
        float a = 0.0f;
        for (int i = j; i < n; i += 16) a = a + v[i];
        part[j] = a;
    }
    for (int m = 8; m >= 1; m >>= 1) {
        float nxt[16];
        for (int j = 0; j < 16; j++) nxt[j] = part[j] + part[j ^ m];
        for (int j = 0; j < 16; j++) part[j] = nxt[j];
    }
    return part[0];
}

/* util.py:70-93: softmax -> expectation over linspace(-(S-1)/2, (S-1)/2, S) -> signed_parabolic */
float mzo_logits_to_value(const float* logits, int32_t S) {
    float m = logits[0];
    for (int i = 1; i < S; i++) m = logits[i] > m ? logits[i] : m;
    float e[1024], t[1024];
    for (int i = 0; i < S; i++) e[i] = mzo_expf(logits[i] - m);
    float sum = row_reduce16(e, S);
    int maxv = (S - 1) / 2;
    for (int i = 0; i < S; i++) {
        float p = e[i] / sum;
        float sup = (float)(i - maxv); /* linspace(-max, max, S) has unit spacing: exact integers */
        t[i] = p * sup;
    }
    float x = row_reduce16(t, S);
    return mzo_signed_parabolic(x);
}

/* util.py:31-36: (h - min) / (max - min + 1e-8) with min/max over dim=1 (channels), per spatial position */
void mzo_normalize_hidden(float* h, int32_t channels, int32_t spatial) {
    for (int p = 0; p < spatial; p++) {
        float mn = h[p], mx = h[p];
        for (int c = 1; c < channels; c++) {
            float v = h[c * spatial + p];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        float d = (mx - mn) + 1e-8f;
        for (int c = 0; c < channels; c++) h[c * spatial + p] = (h[c * spatial + p] - mn) / d;
    }
}

static void softmax_f32(const float* logits, int32_t n, float* out) {
    float m = logits[0];
    for (int i = 1; i < n; i++) m = logits[i] > m ? logits[i] : m;
    for (int i = 0; i < n; i++) out[i] = mzo_expf(logits[i] - m);
    float sum = row_reduce16(out, n);
    for (int i = 0; i < n; i++) out[i] = out[i] / sum;
}

/* y[n] = act(b[n] + sum_k x[k]*W[n][k]) as one k-ordered fmaf chain */
static void linear(const float* x, int K, const float* W, const float* b, int N, float* y, int relu) {
    for (int n = 0; n < N; n++) {
        float acc = b ? b[n] : 0.0f;
        const float* w = W + (size_t)n * K;
        for (int k = 0; k < K; k++) acc = fmaf(x[k], w[k], acc);
        y[n] = (relu && !(acc > 0.0f)) ? 0.0f : acc;
    }
}

/* The MLP nets' Linear layers (network.py:145-149,172-182,212-222) in the summation order of the MFMA tile pipeline
 * (muzero_amd/csrc/mz_mlp.h): inputs are taken in blocks of 16; inside a block the chain visits k = 16g + 4q + i in
 * the order i = 0..3 (outer), q = 0..3 (inner) -- the order in which v_mfma_f32_16x16x4_f32 consumes a D-layout
 * accumulator (lane (e, q) holds neurons 4q..4q+3) as the next layer's B operand.  The Ka one-hot action inputs of the
 * dynamics net (network.py:191-193) follow the Kh hidden inputs in natural order.  `split` (the second layer of every
 * two-layer net, K = num_planes): the blocks are dealt to 4 contiguous quarters, one chain each -- quarter 0 starts
 * from the bias, the others from +0 -- and the result is ((c0 + c1) + c2) + c3: the four waves of a workgroup each own
 * one quarter of the hidden layer in registers and never exchange it.
 * `split` == 2 (the ONE-neuron second layer of an MSE head, value / reward support size 1, network.py:172-182,212-222): a
 * 16-row MFMA tile would carry 15 idle rows, so the kernels run this layer on the vector ALUs, and each quarter is dealt
 * once more to the 4 lane groups that hold its inputs: group q owns k = 16g + 4q + i of every block g of the quarter
 * and runs ONE chain over them (g ascending, i ascending; the chain of quarter 0, group 0 starts from the bias, the
 * others from +0); the groups meet as (p0 + p1) + (p2 + p3), the quarters as before. */
static void linear_mlp(const float* x, int Kh, int Ka, const float* W, const float* b, int N, float* y, int relu, int split) {
    const int NB = (Kh + 15) / 16;
    const int NBq = split ? (NB + 3) / 4 : NB;
    const int chains = split ? 4 : 1;
    if (split == 2) {
        float sc[4];
        for (int c = 0; c < 4; c++) {
            float pq[4];
            const int g1 = (c + 1) * NBq < NB ? (c + 1) * NBq : NB;
            for (int q = 0; q < 4; q++) {
                float acc = (c == 0 && q == 0) ? b[0] : 0.0f;
                for (int g = c * NBq; g < g1; g++)
                    for (int i = 0; i < 4; i++) {
                        const int k = 16 * g + 4 * q + i;
                        if (k < Kh) acc = fmaf(x[k], W[k], acc);
                    }
                pq[q] = acc;
            }
            sc[c] = (pq[0] + pq[1]) + (pq[2] + pq[3]);
        }
        const float total = ((sc[0] + sc[1]) + sc[2]) + sc[3];
        y[0] = (relu && !(total > 0.0f)) ? 0.0f : total;
        return;
    }
    for (int n = 0; n < N; n++) {
        const float* w = W + (size_t)n * (Kh + Ka);
        float total = 0.0f;
        for (int c = 0; c < chains; c++) {
            float acc = (c == 0) ? b[n] : 0.0f;
            const int g1 = (c + 1) * NBq < NB ? (c + 1) * NBq : NB;
            for (int g = c * NBq; g < g1; g++)
                for (int i = 0; i < 4; i++)
                    for (int q = 0; q < 4; q++) {
                        const int k = 16 * g + 4 * q + i;
                        if (k < Kh) acc = fmaf(x[k], w[k], acc);
                    }
            total = (c == 0) ? acc : total + acc;
        }
        for (int a = 0; a < Ka; a++) total = fmaf(x[Kh + a], w[Kh + a], total);
        y[n] = (relu && !(total > 0.0f)) ? 0.0f : total;
    }
}

/* ============================================================================================ */
/* networks                                                                                     */
/* ============================================================================================ */
enum { NET_MLP = 0, NET_CONV = 1, NET_SCRIPTED = 2 };

typedef struct {
    /* conv (optionally followed by eval-mode BatchNorm folded into weight/bias) */
    int cin, cout, k, stride, pad;
    float* w; /* [cout][ky][kx][cin]  (tap-major, channel-minor: the k order of the fmaf chain) */
    float* b; /* [cout] */
} conv_t;

typedef struct {
    conv_t c1, c2;
} resblock_t;

typedef struct {
    conv_t conv;  /* 1x1 + BN */
    float *lw, *lb; /* Linear(cout*h*w -> n_out) */
    int n_out;
} head_t;

struct mzo_net {
    int kind;
    int A, Sv, Sr;
    /* mlp */
    int in_dim, P, H;
    const float* mp[20];
    float* own[20];
    /* conv */
    int conv_kind; /* 0 board, 1 atari */
    int in_c, in_h, in_w, planes, blocks, hh, hw;
    conv_t rep_conv;      /* board: conv+BN ; atari: conv_1 */
    conv_t rep_conv2;     /* atari conv_2 */
    resblock_t* rep_res;  /* board: blocks ; atari: 6 (2+2+2) */
    int n_rep_res;
    conv_t dyn_conv;
    resblock_t* dyn_res;
    head_t reward_head;
    resblock_t* pred_res;
    head_t policy_head, value_head;
    /* scripted */
    const float *s_pi0, *s_values, *s_rewards;
    int s_n, s_calls;
};

static float* dupf(const float* p, size_t n) {
    float* q = (float*)malloc(n * sizeof(float));
    memcpy(q, p, n * sizeof(float));
    return q;
}

mzo_net* mzo_net_create_mlp(int32_t input_dim, int32_t A, int32_t P, int32_t H, int32_t Sv, int32_t Sr, const float* const* params) {
    mzo_net* n = (mzo_net*)calloc(1, sizeof(mzo_net));
    n->kind = NET_MLP;
    n->A = A; n->Sv = Sv; n->Sr = Sr; n->in_dim = input_dim; n->P = P; n->H = H;
    size_t sz[20] = {(size_t)P * input_dim, P, (size_t)H * P, H, (size_t)P * (H + A), P, (size_t)H * P, H, (size_t)P * H, P, (size_t)Sr * P, Sr,
                     (size_t)P * H, P, (size_t)A * P, A, (size_t)P * H, P, (size_t)Sv * P, Sv};
    for (int i = 0; i < 20; i++) {
        n->own[i] = dupf(params[i], sz[i]);
        n->mp[i] = n->own[i];
    }
    return n;
}

mzo_net* mzo_net_create_scripted(int32_t A, const float* pi0, const float* values, const float* rewards, int32_t cnt) {
    mzo_net* n = (mzo_net*)calloc(1, sizeof(mzo_net));
    n->kind = NET_SCRIPTED;
    n->A = A; n->H = 1; n->in_dim = 1;
    n->s_pi0 = pi0; n->s_values = values; n->s_rewards = rewards; n->s_n = cnt; n->s_calls = 0;
    return n;
}

/* fold eval-mode BatchNorm2d (eps 1e-5) into the preceding bias-free conv: alpha = gamma/sqrt(var+eps),
 * w' = w*alpha, b' = beta - mean*alpha   (network.py:283-291 etc.; torch evaluates in*alpha + b') */
static void conv_init(conv_t* c, int cin, int cout, int k, int stride, int pad, const float* w /*[cout][cin][k][k]*/, const float* bn_w,
                      const float* bn_b, const float* bn_m, const float* bn_v) {
    c->cin = cin; c->cout = cout; c->k = k; c->stride = stride; c->pad = pad;
    c->w = (float*)malloc(sizeof(float) * (size_t)cout * cin * k * k);
    c->b = (float*)calloc(cout, sizeof(float));
    for (int co = 0; co < cout; co++) {
        float alpha = 1.0f;
        if (bn_w) {
            float invstd = 1.0f / sqrtf(bn_v[co] + 1e-5f);
            alpha = invstd * bn_w[co];
            float t = bn_m[co] * alpha;
            c->b[co] = bn_b[co] - t;
        }
        for (int ci = 0; ci < cin; ci++)
            for (int ky = 0; ky < k; ky++)
                for (int kx = 0; kx < k; kx++) {
                    float v = w[(((size_t)co * cin + ci) * k + ky) * k + kx];
                    if (bn_w) v = v * alpha;
                    c->w[(((size_t)co * k + ky) * k + kx) * cin + ci] = v;
                }
    }
}

static void conv_free(conv_t* c) {
    free(c->w);
    free(c->b);
}

static int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

/* y[co][oy][ox] = b[co] + sum x[ci][iy][ix]*w as ONE fmaf chain in the order: 16-channel block (major), tap (ky, kx),
 * channel inside the block (minor) -- the order in which the HIP implicit-GEMM kernel stages 16-channel input slabs in LDS.
 * Zero padding contributes fmaf(0, w, acc) == acc, i.e. nothing. */
static void conv_fwd(const conv_t* c, const float* x, int h, int w, float* y, int relu, const float* residual) {
    int oh = conv_out(h, c->k, c->stride, c->pad), ow = conv_out(w, c->k, c->stride, c->pad);
    for (int co = 0; co < c->cout; co++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++) {
                float acc = c->b[co];
                for (int cb = 0; cb < c->cin; cb += 16)
                    for (int ky = 0; ky < c->k; ky++)
                        for (int kx = 0; kx < c->k; kx++) {
                            int iy = oy * c->stride + ky - c->pad, ix = ox * c->stride + kx - c->pad;
                            if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                            const float* wp = c->w + (((size_t)co * c->k + ky) * c->k + kx) * c->cin;
                            const float* xp = x + (size_t)iy * w + ix;
                            int ce = cb + 16 < c->cin ? cb + 16 : c->cin;
                            for (int ci = cb; ci < ce; ci++) acc = fmaf(xp[(size_t)ci * h * w], wp[ci], acc);
                        }
                if (residual) acc = acc + residual[((size_t)co * oh + oy) * ow + ox];
                if (relu && !(acc > 0.0f)) acc = 0.0f;
                y[((size_t)co * oh + oy) * ow + ox] = acc;
            }
}

/* network.py:293-299 */
static void resblock_fwd(const resblock_t* r, float* x, int h, int w, float* tmp) {
    size_t n = (size_t)r->c1.cout * h * w;
    float* t2 = tmp + n;
    conv_fwd(&r->c1, x, h, w, tmp, 1, NULL);
    conv_fwd(&r->c2, tmp, h, w, t2, 1, x);
    memcpy(x, t2, n * sizeof(float));
}

/* nn.AvgPool2d(kernel_size=3, stride=2, padding=1), count_include_pad=True (network.py:337,342) */
static void avgpool_3_2_1(const float* x, int c, int h, int w, float* y) {
    int oh = conv_out(h, 3, 2, 1), ow = conv_out(w, 3, 2, 1);
    for (int ch = 0; ch < c; ch++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++) {
                float acc = 0.0f;
                for (int ky = 0; ky < 3; ky++)
                    for (int kx = 0; kx < 3; kx++) {
                        int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
                        if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                        acc = acc + x[((size_t)ch * h + iy) * w + ix];
                    }
                y[((size_t)ch * oh + oy) * ow + ox] = acc / 9.0f;
            }
}

static const float* const* take_res(resblock_t* r, int planes, const float* const* p) {
    conv_init(&r->c1, planes, planes, 3, 1, 1, p[0], p[1], p[2], p[3], p[4]);
    conv_init(&r->c2, planes, planes, 3, 1, 1, p[5], p[6], p[7], p[8], p[9]);
    return p + 10;
}

static const float* const* take_head(head_t* hd, int planes, int out_planes, int hw, int n_out, const float* const* p) {
    conv_init(&hd->conv, planes, out_planes, 1, 1, 0, p[0], p[1], p[2], p[3], p[4]);
    hd->lw = dupf(p[5], (size_t)n_out * out_planes * hw);
    hd->lb = dupf(p[6], n_out);
    hd->n_out = n_out;
    return p + 7;
}

mzo_net* mzo_net_create_conv(int32_t kind, int32_t in_c, int32_t in_h, int32_t in_w, int32_t A, int32_t blocks, int32_t planes, int32_t Sv,
                             int32_t Sr, const float* const* params, int32_t n_params) {
    mzo_net* n = (mzo_net*)calloc(1, sizeof(mzo_net));
    n->kind = NET_CONV;
    n->conv_kind = kind;
    n->A = A; n->Sv = Sv; n->Sr = Sr;
    n->in_c = in_c; n->in_h = in_h; n->in_w = in_w; n->planes = planes; n->blocks = blocks;
    const float* const* p = params;
    if (kind == 0) {
        n->hh = in_h; n->hw = in_w;
        conv_init(&n->rep_conv, in_c, planes, 3, 1, 1, p[0], p[1], p[2], p[3], p[4]);
        p += 5;
        n->n_rep_res = blocks;
        n->rep_res = (resblock_t*)calloc(blocks > 0 ? blocks : 1, sizeof(resblock_t));
        for (int i = 0; i < blocks; i++) p = take_res(&n->rep_res[i], planes, p);
    } else {
        n->hh = 6; n->hw = 6;
        conv_init(&n->rep_conv, in_c, 128, 3, 2, 1, p[0], NULL, NULL, NULL, NULL);
        p += 1;
        n->n_rep_res = 6;
        n->rep_res = (resblock_t*)calloc(6, sizeof(resblock_t));
        for (int i = 0; i < 2; i++) p = take_res(&n->rep_res[i], 128, p);
        conv_init(&n->rep_conv2, 128, planes, 3, 2, 1, p[0], NULL, NULL, NULL, NULL);
        p += 1;
        for (int i = 2; i < 6; i++) p = take_res(&n->rep_res[i], planes, p);
    }
    int hw = n->hh * n->hw;
    conv_init(&n->dyn_conv, planes + A, planes, 3, 1, 1, p[0], p[1], p[2], p[3], p[4]);
    p += 5;
    n->dyn_res = (resblock_t*)calloc(blocks > 0 ? blocks : 1, sizeof(resblock_t));
    for (int i = 0; i < blocks; i++) p = take_res(&n->dyn_res[i], planes, p);
    p = take_head(&n->reward_head, planes, 1, hw, Sr, p);
    n->pred_res = (resblock_t*)calloc(blocks > 0 ? blocks : 1, sizeof(resblock_t));
    for (int i = 0; i < blocks; i++) p = take_res(&n->pred_res[i], planes, p);
    p = take_head(&n->policy_head, planes, 2, hw, A, p);
    p = take_head(&n->value_head, planes, 1, hw, Sv, p);
    if ((int)(p - params) != n_params) {
        free(n);
        return NULL;
    }
    n->H = planes * hw;
    n->in_dim = in_c * in_h * in_w;
    return n;
}

void mzo_net_destroy(mzo_net* n) {
    if (!n) return;
    if (n->kind == NET_MLP)
        for (int i = 0; i < 20; i++) free(n->own[i]);
    if (n->kind == NET_CONV) {
        conv_free(&n->rep_conv);
        if (n->conv_kind == 1) conv_free(&n->rep_conv2);
        for (int i = 0; i < n->n_rep_res; i++) { conv_free(&n->rep_res[i].c1); conv_free(&n->rep_res[i].c2); }
        for (int i = 0; i < n->blocks; i++) {
            conv_free(&n->dyn_res[i].c1); conv_free(&n->dyn_res[i].c2);
            conv_free(&n->pred_res[i].c1); conv_free(&n->pred_res[i].c2);
        }
        conv_free(&n->dyn_conv);
        head_t* hs[3] = {&n->reward_head, &n->policy_head, &n->value_head};
        for (int i = 0; i < 3; i++) { conv_free(&hs[i]->conv); free(hs[i]->lw); free(hs[i]->lb); }
        free(n->rep_res); free(n->dyn_res); free(n->pred_res);
    }
    free(n);
}

int32_t mzo_net_hidden_size(const mzo_net* n) { return n->H; }
int32_t mzo_net_obs_size(const mzo_net* n) { return n->in_dim; }

/* head: 1x1 conv + BN + ReLU + flatten + Linear (network.py:424-430,472-486) */
static void head_fwd(const head_t* hd, const float* x, int h, int w, float* out, float* tmp) {
    conv_fwd(&hd->conv, x, h, w, tmp, 1, NULL);
    linear(tmp, hd->conv.cout * h * w, hd->lw, hd->lb, hd->n_out, out, 0);
}

static float scalar_from_logits(const float* logits, int S) { return S == 1 ? logits[0] : mzo_logits_to_value(logits, S); }

/* prediction + softmax + value transform: shared tail of network.py:70-75 and :98-103 */
static void mlp_prediction(mzo_net* n, const float* hidden, float* pi_out, float* value_out) {
    float t[4096], lg[1024];
    if (pi_out) {
        linear_mlp(hidden, n->H, 0, n->mp[12], n->mp[13], n->P, t, 1, 0);
        linear_mlp(t, n->P, 0, n->mp[14], n->mp[15], n->A, lg, 0, 1);
        softmax_f32(lg, n->A, pi_out);
    }
    linear_mlp(hidden, n->H, 0, n->mp[16], n->mp[17], n->P, t, 1, 0);
    linear_mlp(t, n->P, 0, n->mp[18], n->mp[19], n->Sv, lg, 0, n->Sv == 1 ? 2 : 1);
    *value_out = scalar_from_logits(lg, n->Sv);
}

static void conv_prediction(mzo_net* n, const float* hidden, float* pi_out, float* value_out) {
    int h = n->hh, w = n->hw;
    size_t sz = (size_t)n->planes * h * w;
    float* feat = (float*)malloc(sizeof(float) * sz * 4);
    float* tmp = feat + sz;
    memcpy(feat, hidden, sz * sizeof(float));
    for (int i = 0; i < n->blocks; i++) resblock_fwd(&n->pred_res[i], feat, h, w, tmp);
    float lg[1024];
    if (pi_out) {
        head_fwd(&n->policy_head, feat, h, w, lg, tmp);
        softmax_f32(lg, n->A, pi_out);
    }
    head_fwd(&n->value_head, feat, h, w, lg, tmp);
    *value_out = scalar_from_logits(lg, n->Sv);
    free(feat);
}

/* network.py:62-84 */
void mzo_initial_inference(mzo_net* n, const float* obs, float* hidden_out, float* pi_out, float* value_out) {
    if (n->kind == NET_SCRIPTED) {
        hidden_out[0] = 0.0f;
        if (pi_out) memcpy(pi_out, n->s_pi0, sizeof(float) * n->A);
        *value_out = 0.123f;
        return;
    }
    if (n->kind == NET_MLP) {
        float t[4096];
        linear_mlp(obs, n->in_dim, 0, n->mp[0], n->mp[1], n->P, t, 1, 0); /* network.py:151-156 */
        linear_mlp(t, n->P, 0, n->mp[2], n->mp[3], n->H, hidden_out, 0, 1);
        mzo_normalize_hidden(hidden_out, n->H, 1);               /* network.py:256-259 */
        mlp_prediction(n, hidden_out, pi_out, value_out);
        return;
    }
    /* conv representation */
    if (n->conv_kind == 0) { /* network.py:389-393 */
        int h = n->in_h, w = n->in_w;
        size_t sz = (size_t)n->planes * h * w;
        float* tmp = (float*)malloc(sizeof(float) * sz * 3);
        conv_fwd(&n->rep_conv, obs, h, w, hidden_out, 1, NULL);
        for (int i = 0; i < n->blocks; i++) resblock_fwd(&n->rep_res[i], hidden_out, h, w, tmp);
        free(tmp);
    } else { /* network.py:344-353 */
        int h = n->in_h, w = n->in_w;
        int h1 = conv_out(h, 3, 2, 1), w1 = conv_out(w, 3, 2, 1);
        int h2 = conv_out(h1, 3, 2, 1), w2 = conv_out(w1, 3, 2, 1);
        int h3 = conv_out(h2, 3, 2, 1), w3 = conv_out(w2, 3, 2, 1);
        size_t s1 = (size_t)128 * h1 * w1;
        size_t s2 = (size_t)n->planes * h2 * w2;
        size_t s3 = (size_t)n->planes * h3 * w3;
        float* a = (float*)malloc(sizeof(float) * (s1 * 3 + s2 * 3 + s3 * 3));
        float* tmpa = a + s1;      /* 2*s1 */
        float* b = tmpa + 2 * s1;  /* s2 */
        float* tmpb = b + s2;      /* 2*s2 */
        float* c = tmpb + 2 * s2;  /* s3 */
        float* tmpc = c + s3;      /* 2*s3 */
        conv_fwd(&n->rep_conv, obs, h, w, a, 1, NULL);
        for (int i = 0; i < 2; i++) resblock_fwd(&n->rep_res[i], a, h1, w1, tmpa);
        conv_fwd(&n->rep_conv2, a, h1, w1, b, 1, NULL);
        for (int i = 2; i < 4; i++) resblock_fwd(&n->rep_res[i], b, h2, w2, tmpb);
        avgpool_3_2_1(b, n->planes, h2, w2, c);
        for (int i = 4; i < 6; i++) resblock_fwd(&n->rep_res[i], c, h3, w3, tmpc);
        avgpool_3_2_1(c, n->planes, h3, w3, hidden_out);
        free(a);
    }
    mzo_normalize_hidden(hidden_out, n->planes, n->hh * n->hw);
    conv_prediction(n, hidden_out, pi_out, value_out);
}

/* network.py:86-111 */
void mzo_recurrent_inference(mzo_net* n, const float* hidden_in, int32_t action, float* hidden_out, float* reward_out, float* pi_out,
                             float* value_out) {
    if (n->kind == NET_SCRIPTED) {
        int s = n->s_calls++;
        hidden_out[0] = (float)(s + 1);
        *reward_out = n->s_rewards[s];
        *value_out = n->s_values[s];
        if (pi_out) memcpy(pi_out, n->s_pi0, sizeof(float) * n->A);
        return;
    }
    if (n->kind == NET_MLP) {
        float x[4096], t[4096], lg[1024];
        memcpy(x, hidden_in, sizeof(float) * n->H);
        for (int a = 0; a < n->A; a++) x[n->H + a] = (a == action) ? 1.0f : 0.0f; /* network.py:191-193 */
        linear_mlp(x, n->H, n->A, n->mp[4], n->mp[5], n->P, t, 1, 0);
        linear_mlp(t, n->P, 0, n->mp[6], n->mp[7], n->H, hidden_out, 0, 1);
        /* reward head reads the UN-normalised hidden state (network.py:195-196); normalisation follows (:263) */
        linear_mlp(hidden_out, n->H, 0, n->mp[8], n->mp[9], n->P, t, 1, 0);
        linear_mlp(t, n->P, 0, n->mp[10], n->mp[11], n->Sr, lg, 0, n->Sr == 1 ? 2 : 1);
        *reward_out = scalar_from_logits(lg, n->Sr);
        mzo_normalize_hidden(hidden_out, n->H, 1);
        mlp_prediction(n, hidden_out, pi_out, value_out);
        return;
    }
    int h = n->hh, w = n->hw, hw = h * w;
    size_t cin = (size_t)n->planes + n->A;
    float* x = (float*)malloc(sizeof(float) * (cin * hw + (size_t)n->planes * hw * 3));
    float* tmp = x + cin * hw;
    memcpy(x, hidden_in, sizeof(float) * n->planes * hw);
    /* "scrambled" action planes: one_hot [1,1,A] -> repeat_interleave(h*w, dim=1) -> reshape (A,h,w)
     * (network.py:440-444) gives plane element f = c*h*w + y*w + x equal to 1 iff f % A == action */
    for (int f = 0; f < n->A * hw; f++) x[(size_t)n->planes * hw + f] = ((f % n->A) == action) ? 1.0f : 0.0f;
    conv_fwd(&n->dyn_conv, x, h, w, hidden_out, 1, NULL);
    for (int i = 0; i < n->blocks; i++) resblock_fwd(&n->dyn_res[i], hidden_out, h, w, tmp);
    float lg[1024];
    head_fwd(&n->reward_head, hidden_out, h, w, lg, tmp);
    *reward_out = scalar_from_logits(lg, n->Sr);
    free(x);
    mzo_normalize_hidden(hidden_out, n->planes, hw);
    conv_prediction(n, hidden_out, pi_out, value_out);
}

/* ============================================================================================ */
/* numpy reductions used by the reference on the path                                            */
/* ============================================================================================ */
/* np.sum on a contiguous 1-D array == numpy's pairwise_sum (loops_utils.h.src), verified empirically */
static double np_sum_f64(const double* a, long n) {
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; i++) r = r + a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum_f64(a, n2) + np_sum_f64(a + n2, n - n2);
    }
}

static float np_sum_f32(const float* a, long n) {
    if (n < 8) {
        float r = 0.0f;
        for (long i = 0; i < n; i++) r = r + a[i];
        return r;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
    }
}

/* ============================================================================================ */
/* root prior, play policy, sampling (mcts.py:220-299, 391-404)                                  */
/* ============================================================================================ */
/* mcts.py:357-365.  Self-play: prior becomes float64 through add_dirichlet_noise (:244-247), where
 * (1-eps)*prob is a float32 product (python scalar x float32 array) and eps*noise is float64.
 * Deterministic (or alpha/eps == 0): prior stays float32 (numpy-2 promotion rules). */
void mzo_prepare_root_prior(const float* pi0, int32_t A, const double* noise, double eps, const uint8_t* mask, int32_t deterministic,
                            double* prior64, float* prior32) {
    int use_noise = (!deterministic) && noise != NULL;
    if (use_noise) {
        float om = (float)(1.0 - eps);
        for (int a = 0; a < A; a++) {
            float t = om * pi0[a];
            double e = eps * noise[a];
            prior64[a] = (double)t + e;
        }
        if (mask) {
            for (int a = 0; a < A; a++)
                if (!mask[a]) prior64[a] = 0.0;
            double s = np_sum_f64(prior64, A); /* mcts.py:295-298 */
            if (s > 0)
                for (int a = 0; a < A; a++) prior64[a] = prior64[a] / s;
        }
        for (int a = 0; a < A; a++) prior32[a] = (float)prior64[a];
    } else {
        for (int a = 0; a < A; a++) prior32[a] = pi0[a];
        if (mask) {
            for (int a = 0; a < A; a++)
                if (!mask[a]) prior32[a] = 0.0f;
            float s = np_sum_f32(prior32, A);
            if (s > 0)
                for (int a = 0; a < A; a++) prior32[a] = prior32[a] / s;
        }
        for (int a = 0; a < A; a++) prior64[a] = (double)prior32[a];
    }
}

/* mcts.py:250-280: T>0 => visits ** clip(1/T, 1, 5); then / sum.  T == 0 => linear in visits. */
void mzo_generate_play_policy(const int32_t* visits, int32_t A, double temperature, double* pi) {
    double* v = (double*)malloc(sizeof(double) * A);
    for (int a = 0; a < A; a++) v[a] = (double)visits[a];
    if (temperature > 0.0) {
        double e = 1.0 / temperature;
        e = e < 5.0 ? e : 5.0;
        e = e > 1.0 ? e : 1.0;
        for (int a = 0; a < A; a++) v[a] = pow(v[a], e);
    }
    double s = np_sum_f64(v, A);
    for (int a = 0; a < A; a++) pi[a] = v[a] / s;
    free(v);
}

/* np.random.choice(arange(A), p=pi) (mcts.py:404): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right') */
int32_t mzo_sample_action(const double* pi, int32_t A, double u) {
    double* cdf = (double*)malloc(sizeof(double) * A);
    double c = 0.0;
    for (int a = 0; a < A; a++) {
        c = c + pi[a];
        cdf[a] = c;
    }
    double last = cdf[A - 1];
    int32_t idx = 0;
    for (int a = 0; a < A; a++) {
        double q = cdf[a] / last;
        if (q <= u) idx = a + 1;
    }
    free(cdf);
    if (idx >= A) idx = A - 1;
    return idx;
}

/* ============================================================================================ */
/* the search (mcts.py:302-407), SoA tree: node 0 = root, node s+1 created by simulation s       */
/* ============================================================================================ */
mzo_search_result mzo_uct_search(const mzo_search_config* cfg, mzo_net* net, const float* obs, const uint8_t* mask, int32_t current_player,
                                 int32_t opponent_player, double temperature, int32_t deterministic, const mzo_rng_inputs* rng,
                                 double* out_pi, int32_t* out_visits, int32_t* trace_parent, int32_t* trace_action, double* out_minmax) {
    mzo_search_result res = {0, 0.0, 0, 0};
    const int A = cfg->num_actions, S = cfg->num_simulations, H = net->H;
    const int board = cfg->is_board_game;
    const double gamma = cfg->discount;
    const int nn = S + 1;

    int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * nn);
    int32_t* move = (int32_t*)malloc(sizeof(int32_t) * nn);
    int32_t* player = (int32_t*)malloc(sizeof(int32_t) * nn);
    double* reward = (double*)malloc(sizeof(double) * nn);
    int32_t* N = (int32_t*)calloc(nn, sizeof(int32_t));
    double* W = (double*)calloc(nn, sizeof(double));
    int32_t* child = (int32_t*)malloc(sizeof(int32_t) * (size_t)nn * A);
    float* hidden = (float*)malloc(sizeof(float) * (size_t)nn * H);
    float* pi0 = (float*)malloc(sizeof(float) * A);
    double* prior64 = (double*)malloc(sizeof(double) * A);
    float* prior32 = (float*)malloc(sizeof(float) * A);
    float* q = (float*)malloc(sizeof(float) * A);
    float* ucb = (float*)malloc(sizeof(float) * A);
    for (size_t i = 0; i < (size_t)nn * A; i++) child[i] = -1;

    /* MinMaxStats (mcts.py:33-48) */
    double mm_max = cfg->has_known_bounds ? cfg->kb_max : -INFINITY;
    double mm_min = cfg->has_known_bounds ? cfg->kb_min : INFINITY;

    /* root (mcts.py:355-367); the root value from the network is discarded, reward forced to 0 (network.py:76) */
    float v0;
    mzo_initial_inference(net, obs, hidden, pi0, &v0);
    int use_noise = (!deterministic) && cfg->dirichlet_alpha > 0.0 && cfg->exploration_eps > 0.0 && rng && rng->noise;
    mzo_prepare_root_prior(pi0, A, use_noise ? rng->noise : NULL, cfg->exploration_eps, mask, deterministic, prior64, prior32);
    /* child_U (mcts.py:189-197): `child.prior * (python float)`.  With noise the prior is float64.  Without, it is an np.float32 SCALAR:
     * numpy >= 2 (NEP 50) multiplies in float32 (the form the fixtures were recorded under: this container runs numpy 2.2.6); numpy 1.x -- the
     * reference pins 1.21.6, requirements.txt:21 -- promotes scalar x scalar to float64 and np.array(dtype=float32) rounds once. */
    const int prior_is_f64 = use_noise || cfg->legacy_scalar_promotion;
    parent[0] = -1; move[0] = -1; player[0] = current_player; reward[0] = 0.0;

    int tie_used = 0;
    for (int s = 0; s < S; s++) {
        int n = 0, a_sel = 0;
        int cp = current_player, op = opponent_player;
        for (;;) { /* best_child, mcts.py:104-127 */
            double pb = (log(((double)N[n] + cfg->pb_c_base + 1.0) / cfg->pb_c_base) + cfg->pb_c_init) * sqrt((double)N[n]);
            float best = -INFINITY;
            for (int a = 0; a < A; a++) {
                int c = child[(size_t)n * A + a];
                int cn = c >= 0 ? N[c] : 0;
                float qa = 0.0f; /* child_Q, mcts.py:159-178 */
                if (cn > 0) {
                    double Q = W[c] / (double)cn;
                    double v = reward[c] + (gamma * (board ? -1.0 : 1.0)) * Q;
                    if (mm_max > mm_min) v = (v - mm_min) / (mm_max - mm_min);
                    qa = (float)v;
                }
                double f = pb / (double)(cn + 1); /* child_U, mcts.py:180-200 */
                float ua = prior_is_f64 ? (float)(prior64[a] * f) : (prior32[a] * (float)f);
                ucb[a] = qa + ua;
                if (ucb[a] > best) best = ucb[a];
            }
            int ncand = 0;
            for (int a = 0; a < A; a++) ncand += (ucb[a] == best);
            int pick = 0;
            if (ncand > 1) { /* np.random.choice(np.where(ucb == max)[0]) consumes randomness only here */
                if (!rng || tie_used >= rng->n_tie) {
                    res.status = MZO_E_TIES_EXHAUSTED;
                    goto done;
                }
                double u = rng->u_tie[tie_used++];
                pick = (int)floor(u * ncand);
                if (pick >= ncand) pick = ncand - 1;
            }
            for (int a = 0, k = 0; a < A; a++)
                if (ucb[a] == best) {
                    if (k == pick) { a_sel = a; break; }
                    k++;
                }
            int t = cp; cp = op; op = t; /* mcts.py:379 */
            int c = child[(size_t)n * A + a_sel];
            if (c < 0) break;
            n = c;
        }
        /* expand + evaluate (mcts.py:382-386): every expanded node receives the ROOT prior */
        int nw = s + 1;
        float r32, v32;
        mzo_recurrent_inference(net, hidden + (size_t)n * H, a_sel, hidden + (size_t)nw * H, &r32, NULL, &v32);
        if (trace_parent) trace_parent[s] = n;
        if (trace_action) trace_action[s] = a_sel;
        child[(size_t)n * A + a_sel] = nw;
        parent[nw] = n; move[nw] = a_sel; player[nw] = cp; reward[nw] = (double)r32;
        /* backup (mcts.py:129-157) */
        double val = (double)v32;
        for (int c = nw; c >= 0; c = parent[c]) {
            W[c] += (player[c] == cp) ? val : -val;
            N[c] += 1;
            double Q = W[c] / (double)N[c];
            double x = board ? (reward[c] + gamma * -Q) : (reward[c] + gamma * Q);
            if (x > mm_max) mm_max = x;
            if (x < mm_min) mm_min = x;
            if (board && player[c] == cp) val = -reward[c] + gamma * val;
            else val = reward[c] + gamma * val;
        }
    }
    {
        /* play (mcts.py:391-407) */
        int32_t* visits = (int32_t*)malloc(sizeof(int32_t) * A);
        for (int a = 0; a < A; a++) {
            int c = child[a];
            int v = c >= 0 ? N[c] : 0;
            visits[a] = (mask && !mask[a]) ? 0 : v;
        }
        if (out_visits) memcpy(out_visits, visits, sizeof(int32_t) * A);
        mzo_generate_play_policy(visits, A, temperature, out_pi);
        if (deterministic) {
            int best = 0;
            for (int a = 1; a < A; a++)
                if (visits[a] > visits[best]) best = a;
            res.action = best;
        } else {
            res.action = mzo_sample_action(out_pi, A, rng ? rng->u_final : 0.5);
        }
        free(visits);
        res.root_value = N[0] > 0 ? W[0] / (double)N[0] : 0.0;
        res.n_tie_used = tie_used;
        if (out_minmax) { out_minmax[0] = mm_min; out_minmax[1] = mm_max; }
    }
done:
    free(parent); free(move); free(player); free(reward); free(N); free(W); free(child); free(hidden);
    free(pi0); free(prior64); free(prior32); free(q); free(ucb);
    return res;
}

int32_t mzo_uct_search_batch(const mzo_search_config* cfg, mzo_net* net, int32_t batch, const float* obs, const uint8_t* mask,
                             const int32_t* cur_player, const int32_t* opp_player, const double* temperature, int32_t deterministic,
                             const double* noise, const double* u_tie, int32_t n_tie, const double* u_final, int32_t* out_action,
                             double* out_pi, double* out_root_value, int32_t* out_visits, int32_t num_threads) {
    const int A = cfg->num_actions;
    const size_t od = (size_t)net->in_dim;
    int32_t status = 0;
    if (net->kind == NET_SCRIPTED) return MZO_E_BAD_ARG; /* scripted nets carry per-search state */
#ifdef _OPENMP
    if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < batch; b++) {
        mzo_rng_inputs r;
        r.noise = noise ? noise + (size_t)b * A : NULL;
        r.u_tie = u_tie + (size_t)b * n_tie;
        r.n_tie = n_tie;
        r.u_final = u_final[b];
        mzo_search_result sr = mzo_uct_search(cfg, net, obs + (size_t)b * od, mask ? mask + (size_t)b * A : NULL, cur_player[b], opp_player[b],
                                              temperature[b], deterministic, &r, out_pi + (size_t)b * A,
                                              out_visits ? out_visits + (size_t)b * A : NULL, NULL, NULL, NULL);
        out_action[b] = sr.action;
        out_root_value[b] = sr.root_value;
        if (sr.status != 0) {
#pragma omp atomic write
            status = sr.status;
        }
    }
    return status;
}

/* ============================================================================================ */
/* environments                                                                                 */
/* ============================================================================================ */
/* gym 0.23.1 classic_control/cartpole.py (un-vendored dependency, requirements.txt:7): Euler integration,
 * gravity 9.8, masscart 1.0, masspole 0.1, length 0.5 (half), force_mag 10, tau 0.02; terminate when
 * |x| > 2.4 or |theta| > 12 deg; reward 1.0 every step; TimeLimit(max_episode_steps=500) for CartPole-v1. */
void mzo_cartpole_reset(mzo_cartpole* e, const double init[4]) {
    for (int i = 0; i < 4; i++) e->s[i] = init[i];
    e->steps = 0;
    e->done = 0;
}

double mzo_cartpole_step(mzo_cartpole* e, int32_t action, float obs_out[4]) {
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, total_mass = masspole + masscart, length = 0.5;
    const double polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
    const double theta_threshold = 12.0 * 2.0 * M_PI / 360.0, x_threshold = 2.4;
    double x = e->s[0], x_dot = e->s[1], theta = e->s[2], theta_dot = e->s[3];
    double force = action == 1 ? force_mag : -force_mag;
    double costheta = cos(theta), sintheta = sin(theta);
    double temp = (force + polemass_length * theta_dot * theta_dot * sintheta) / total_mass;
    double thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * costheta * costheta / total_mass));
    double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    e->s[0] = x; e->s[1] = x_dot; e->s[2] = theta; e->s[3] = theta_dot;
    e->steps += 1;
    int term = (x < -x_threshold) || (x > x_threshold) || (theta < -theta_threshold) || (theta > theta_threshold);
    e->done = term || (e->steps >= 500);
    for (int i = 0; i < 4; i++) obs_out[i] = (float)e->s[i];
    return 1.0;
}

/* gym_env.py:326-353: reset fills every slot with the first observation and the bias (0+1)/A */
void mzo_stack_reset(float* stacked, int32_t stack, int32_t dim, const float* obs, int32_t num_actions) {
    float bias = (float)((0 + 1) / (double)num_actions);
    for (int k = 0; k < stack; k++) {
        for (int i = 0; i < dim; i++) stacked[k * (dim + 1) + i] = obs[i];
        stacked[k * (dim + 1) + dim] = bias;
    }
}

/* gym_env.py:317-324: appendleft => row 0 is the newest (obs_t, (a_t+1)/A) */
void mzo_stack_push(float* stacked, int32_t stack, int32_t dim, const float* obs, int32_t action, int32_t num_actions) {
    for (int k = stack - 1; k > 0; k--) memcpy(stacked + k * (dim + 1), stacked + (k - 1) * (dim + 1), sizeof(float) * (dim + 1));
    for (int i = 0; i < dim; i++) stacked[i] = obs[i];
    stacked[dim] = (float)((action + 1) / (double)num_actions);
}

/* games/env.py:98-115 */
void mzo_board_reset(mzo_board* b, int32_t board_size, int32_t stack, int32_t num_to_win) {
    memset(b, 0, sizeof(*b));
    b->board_size = board_size; b->stack = stack; b->num_to_win = num_to_win;
    b->num_actions = board_size * board_size + 1;
    for (int a = 0; a < b->num_actions; a++) b->mask[a] = 1;
    b->current_player = 1;
    b->last_action[0] = b->last_action[1] = -1;
}

int32_t mzo_board_game_over(const mzo_board* b) {
    if (b->winner) return 1;
    for (int i = 0; i < b->board_size * b->board_size; i++)
        if (b->board[i] == 0) return 0;
    return 1;
}

static int count_dir(const mzo_board* b, int r, int c, int dr, int dc, int colour) {
    int n = 0, N = b->board_size;
    r += dr; c += dc;
    while (r >= 0 && r < N && c >= 0 && c < N && b->board[r * N + c] == colour) { n++; r += dr; c += dc; }
    return n;
}

/* games/tictactoe.py:33-77 == games/gomoku.py:72-116: test the 4 lines through the last move only */
static int current_player_won(const mzo_board* b, int action) {
    if (b->steps < (b->num_to_win - 1) * 2) return 0;
    int N = b->board_size, r = action / N, c = action % N, colour = b->current_player;
    const int dirs[4][2] = {{0, 1}, {1, 0}, {1, 1}, {-1, 1}};
    for (int d = 0; d < 4; d++) {
        int len = 1 + count_dir(b, r, c, dirs[d][0], dirs[d][1], colour) + count_dir(b, r, c, -dirs[d][0], -dirs[d][1], colour);
        if (len >= b->num_to_win) return 1;
    }
    return 0;
}

/* games/env.py:117-154 */
int32_t mzo_board_step(mzo_board* b, int32_t action, double* reward, int32_t* done) {
    if (action < 0 || action >= b->num_actions) return -1;
    if (!b->mask[action]) return -2;
    if (mzo_board_game_over(b)) return -3;
    *reward = 0.0;
    b->mask[action] = 0;
    int me = b->current_player, opp = 3 - me;
    b->last_action[me - 1] = action;
    if (action == b->num_actions - 1) { /* resign */
        *reward = -1.0;
        b->winner = opp;
    } else {
        int nn = b->board_size * b->board_size;
        b->board[action] = (int8_t)me; /* colour id == player id (1 black, 2 white) */
        /* _update_feature_planes (games/env.py:294-302): push the mover's own-stone plane */
        for (int k = b->stack - 1; k > 0; k--) memcpy(b->planes[me - 1][k], b->planes[me - 1][k - 1], nn);
        for (int i = 0; i < nn; i++) b->planes[me - 1][0][i] = (b->board[i] == me);
        if (current_player_won(b, action)) {
            *reward = 1.0;
            b->winner = me;
        }
    }
    *done = mzo_board_game_over(b);
    if (!*done) b->current_player = opp;
    b->steps += 1;
    return 0;
}

/* games/env.py:242-271: [X_t, Y_t, X_t-1, Y_t-1, ..., C] from the side to move */
void mzo_board_observation(const mzo_board* b, int8_t* obs) {
    int nn = b->board_size * b->board_size, me = b->current_player, opp = 3 - me;
    for (int t = 0; t < b->stack; t++) {
        memcpy(obs + (size_t)(2 * t) * nn, b->planes[me - 1][t], nn);
        memcpy(obs + (size_t)(2 * t + 1) * nn, b->planes[opp - 1][t], nn);
    }
    memset(obs + (size_t)(2 * b->stack) * nn, me == 1 ? 1 : 0, nn);
}

/* pipeline.py:632-673 */
void mzo_n_step_target(const double* rewards, const double* root_values, int32_t T, int32_t td_steps, double discount, double* out) {
    for (int t = 0; t < T; t++) {
        int bi = t + td_steps;
        double value = 0.0;
        for (int i = 0; t + i < bi; i++) {
            double r = (t + i < T) ? rewards[t + i] : 0.0;
            value = value + pow(discount, (double)i) * r;
        }
        double rv = bi < T ? root_values[bi] : 0.0;
        value = value + pow(discount, (double)td_steps) * rv;
        out[t] = value;
    }
}

/* pipeline.py:676-707 */
void mzo_mc_return_target(const double* rewards, const int32_t* player_ids, int32_t T, double* out) {
    for (int t = 0; t < T; t++) out[t] = 0.0;
    if (T == 0) return;
    double fr = rewards[T - 1];
    int fp = player_ids[T - 1];
    if (fr != 0.0)
        for (int t = 0; t < T; t++) out[t] = (player_ids[t] == fp) ? fr : -fr;
}

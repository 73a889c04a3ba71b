// mz_env.h -- device-resident environments and the self-play record ring (run_self_play's bookkeeping,
// pipeline.py:83-113) so that a whole lock-step move -- search, action sampling, env.step, trajectory record,
// auto-reset -- stays in HBM.  One thread per environment: these kernels move a few hundred bytes per env per move,
// they are launch-latency-sized, not bandwidth-sized.
//
//   CartPole-v1   : gym 0.23.1 equations (un-vendored dependency of the reference, requirements.txt:7),
//                   float64 state, float32 observation, StackFrameAndAction(4) rows [obs_t-k, (a_t-k + 1)/A] newest
//                   first (gym_env.py:306-353), players 1/1 and an all-true mask (gym_env.py:356-365), TimeLimit 500.
//   TicTacToe     : BoardGameEnv semantics (games/env.py:117-154,242-310; games/tictactoe.py:33-77): per-player
//                   own-stone history planes, resign action 9, win test through the last move, player switches only
//                   if the game is not over.
#pragma once
#include "mz_device.h"

namespace mz {

constexpr int ENV_CARTPOLE = 1;
constexpr int ENV_TICTACTOE = 2;

struct EnvState {
    int kind, B, A, D, ring_len;
    double* cp_state;      // [B][4]
    int* steps;            // [B] steps in the current episode
    unsigned int* episode; // [B] episode index (keys the reset RNG)
    double* init_state;    // [B][4] optional externally supplied reset states (tests)
    signed char* board;    // [B][9]
    signed char* planes;   // [B][2][4][9]
    int* player;           // [B] side to move (1 black, 2 white)
    // record ring, slot-major
    float* r_obs;          // [ring][B][D]
    int* r_action;         // [ring][B]
    float* r_reward;       // [ring][B]
    double* r_pi;          // [ring][B][A]
    double* r_root;        // [ring][B]
    int* r_player;         // [ring][B]
    unsigned char* r_done; // [ring][B]
    unsigned long long* counters;  // env steps, simulations, finished episodes, sum of finished episode lengths
};

struct EnvLaunch {
    EnvState env;
    int B;
    unsigned long long seed;
    int use_init;
    double temperature;  // >= 0: constant; < 0: board-game schedule (1.0 for the first 6 moves, then 0.1; config.py:236-241)
    unsigned int move_counter;
    int slot, sims;
    float* obs;
    unsigned char* mask;
    int* cur;
    int* opp;
    double* temp_out;
    const int* action;
    const double* pi;
    const double* root;
};

inline void env_free(EnvState& e) {
    void* bufs[] = {e.cp_state, e.steps, e.episode, e.init_state, e.board, e.planes, e.player, e.r_obs, e.r_action,
                    e.r_reward, e.r_pi, e.r_root, e.r_player, e.r_done, e.counters};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    e = EnvState{};
}

inline hipError_t env_alloc(EnvState& e, int kind, int B, int A, int D, int ring_len) {
    env_free(e);
    e.kind = kind; e.B = B; e.A = A; e.D = D; e.ring_len = ring_len;
    hipError_t r;
#define MZ_ALLOC(ptr, bytes)                                  \
    if ((r = hipMalloc(&(ptr), (bytes))) != hipSuccess) return r; \
    if ((r = hipMemset((ptr), 0, (bytes))) != hipSuccess) return r;
    MZ_ALLOC(e.cp_state, (size_t)B * 4 * sizeof(double));
    MZ_ALLOC(e.steps, (size_t)B * sizeof(int));
    MZ_ALLOC(e.episode, (size_t)B * sizeof(unsigned int));
    MZ_ALLOC(e.init_state, (size_t)B * 4 * sizeof(double));
    MZ_ALLOC(e.board, (size_t)B * 9);
    MZ_ALLOC(e.planes, (size_t)B * 72);
    MZ_ALLOC(e.player, (size_t)B * sizeof(int));
    MZ_ALLOC(e.r_obs, (size_t)ring_len * B * D * sizeof(float));
    MZ_ALLOC(e.r_action, (size_t)ring_len * B * sizeof(int));
    MZ_ALLOC(e.r_reward, (size_t)ring_len * B * sizeof(float));
    MZ_ALLOC(e.r_pi, (size_t)ring_len * B * A * sizeof(double));
    MZ_ALLOC(e.r_root, (size_t)ring_len * B * sizeof(double));
    MZ_ALLOC(e.r_player, (size_t)ring_len * B * sizeof(int));
    MZ_ALLOC(e.r_done, (size_t)ring_len * B);
    MZ_ALLOC(e.counters, 4 * sizeof(unsigned long long));
#undef MZ_ALLOC
    return hipSuccess;
}

// ---- CartPole ----
__device__ inline void cartpole_fresh(const EnvLaunch& L, int e, double s[4]) {
    if (L.use_init && L.env.episode[e] == 0) {
        for (int i = 0; i < 4; i++) s[i] = L.env.init_state[e * 4 + i];
    } else {
        Philox g(L.seed, (unsigned)e, L.env.episode[e], 0x40000000u);  // reset: U(-0.05, 0.05)^4
        for (int i = 0; i < 4; i++) s[i] = -0.05 + 0.1 * g.uniform();
    }
}

__device__ inline void cartpole_obs_reset(const EnvLaunch& L, int e, const double s[4]) {
    float* o = L.obs + (size_t)e * 20;
    const float bias = (float)((0 + 1) / (double)2);  // gym_env.py:333-336,347
    for (int k = 0; k < 4; k++) {
        for (int i = 0; i < 4; i++) o[k * 5 + i] = (float)s[i];
        o[k * 5 + 4] = bias;
    }
}

__device__ inline bool cartpole_physics(double s[4], int action) {
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, total_mass = masspole + masscart, length = 0.5;
    const double polemass_length = masspole * length, force_mag = 10.0, tau = 0.02;
    const double theta_threshold = 12.0 * 2.0 * 3.14159265358979323846 / 360.0, x_threshold = 2.4;
    double x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    const double force = action == 1 ? force_mag : -force_mag;
    const double costheta = cos(theta), sintheta = sin(theta);
    const double temp = (force + polemass_length * theta_dot * theta_dot * sintheta) / total_mass;
    const double thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * costheta * costheta / total_mass));
    const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    s[0] = x; s[1] = x_dot; s[2] = theta; s[3] = theta_dot;
    return (x < -x_threshold) || (x > x_threshold) || (theta < -theta_threshold) || (theta > theta_threshold);
}

// ---- TicTacToe ----
__device__ inline void ttt_write_obs(const EnvLaunch& L, int e) {
    // [X_t, Y_t, X_t-1, Y_t-1, ..., C] from the side to move (games/env.py:242-271)
    float* o = L.obs + (size_t)e * 81;
    const signed char* pl = L.env.planes + (size_t)e * 72;
    const int me = L.env.player[e], opp = 3 - me;
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 9; i++) {
            o[(2 * t) * 9 + i] = (float)pl[((me - 1) * 4 + t) * 9 + i];
            o[(2 * t + 1) * 9 + i] = (float)pl[((opp - 1) * 4 + t) * 9 + i];
        }
    for (int i = 0; i < 9; i++) o[72 + i] = me == 1 ? 1.0f : 0.0f;
    L.cur[e] = me;
    L.opp[e] = opp;
}

__device__ inline void ttt_fresh(const EnvLaunch& L, int e) {
    for (int i = 0; i < 9; i++) L.env.board[(size_t)e * 9 + i] = 0;
    for (int i = 0; i < 72; i++) L.env.planes[(size_t)e * 72 + i] = 0;
    for (int a = 0; a < 10; a++) L.mask[(size_t)e * 10 + a] = 1;
    L.env.player[e] = 1;
    ttt_write_obs(L, e);
}

__device__ inline int ttt_line(const signed char* b, int r, int c, int dr, int dc, int colour) {
    int n = 0;
    r += dr; c += dc;
    while (r >= 0 && r < 3 && c >= 0 && c < 3 && b[r * 3 + c] == colour) { n++; r += dr; c += dc; }
    return n;
}

__global__ void k_env_reset(const EnvLaunch L) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0)
        for (int i = 0; i < 4; i++) L.env.counters[i] = 0;
    if (e >= L.B) return;
    L.env.steps[e] = 0;
    L.env.episode[e] = 0;
    if (L.env.kind == ENV_CARTPOLE) {
        double s[4];
        cartpole_fresh(L, e, s);
        for (int i = 0; i < 4; i++) L.env.cp_state[e * 4 + i] = s[i];
        cartpole_obs_reset(L, e, s);
        L.mask[(size_t)e * 2] = 1; L.mask[(size_t)e * 2 + 1] = 1;
        L.cur[e] = 1; L.opp[e] = 1;
    } else {
        ttt_fresh(L, e);
    }
}

// before the search of a move: temperature for this move and the "obs / player before acting" half of the record
__global__ void k_env_pre(const EnvLaunch L) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.B) return;
    double T = L.temperature;
    if (T < 0.0) T = L.env.steps[e] < 6 ? 1.0 : 0.1;
    L.temp_out[e] = T;
    const int D = L.env.D;
    float* ro = L.env.r_obs + ((size_t)L.slot * L.B + e) * D;
    const float* o = L.obs + (size_t)e * D;
    for (int i = 0; i < D; i++) ro[i] = o[i];
    L.env.r_player[(size_t)L.slot * L.B + e] = L.cur[e];
}

// after the search: env.step(action), record, auto-reset (pipeline.py:106-113)
__global__ void k_env_step(const EnvLaunch L) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.B) return;
    const int a = L.action[e], A = L.env.A;
    const size_t rec = (size_t)L.slot * L.B + e;
    float reward = 0.0f;
    bool done = false;
    if (L.env.kind == ENV_CARTPOLE) {
        double s[4];
        for (int i = 0; i < 4; i++) s[i] = L.env.cp_state[e * 4 + i];
        const bool term = cartpole_physics(s, a);
        reward = 1.0f;
        const int st = L.env.steps[e] + 1;
        done = term || st >= 500;
        if (!done) {
            L.env.steps[e] = st;
            for (int i = 0; i < 4; i++) L.env.cp_state[e * 4 + i] = s[i];
            float* o = L.obs + (size_t)e * 20;  // appendleft (gym_env.py:317-324)
            for (int k = 3; k > 0; k--)
                for (int i = 0; i < 5; i++) o[k * 5 + i] = o[(k - 1) * 5 + i];
            for (int i = 0; i < 4; i++) o[i] = (float)s[i];
            o[4] = (float)((a + 1) / (double)2);
        } else {
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)st);
            L.env.steps[e] = 0;
            L.env.episode[e] += 1;
            cartpole_fresh(L, e, s);
            for (int i = 0; i < 4; i++) L.env.cp_state[e * 4 + i] = s[i];
            cartpole_obs_reset(L, e, s);
        }
    } else {
        signed char* b = L.env.board + (size_t)e * 9;
        signed char* pl = L.env.planes + (size_t)e * 72;
        const int me = L.env.player[e], opp = 3 - me;
        const int st = L.env.steps[e];
        int winner = 0;
        L.mask[(size_t)e * 10 + a] = 0;
        if (a == 9) {  // resign (games/env.py:134-136)
            reward = -1.0f;
            winner = opp;
        } else {
            b[a] = (signed char)me;
            signed char* mine = pl + (me - 1) * 36;
            for (int k = 3; k > 0; k--)
                for (int i = 0; i < 9; i++) mine[k * 9 + i] = mine[(k - 1) * 9 + i];
            for (int i = 0; i < 9; i++) mine[i] = b[i] == me;
            if (st >= 4) {  // games/tictactoe.py:37-38 with the pre-increment step count
                const int r = a / 3, c = a % 3;
                const int dirs[4][2] = {{0, 1}, {1, 0}, {1, 1}, {-1, 1}};
                for (int d = 0; d < 4; d++)
                    if (1 + ttt_line(b, r, c, dirs[d][0], dirs[d][1], me) + ttt_line(b, r, c, -dirs[d][0], -dirs[d][1], me) >= 3) winner = me;
            }
            if (winner) reward = 1.0f;
        }
        bool full = true;
        for (int i = 0; i < 9; i++) full = full && b[i] != 0;
        done = winner != 0 || full;
        if (!done) {
            L.env.player[e] = opp;
            L.env.steps[e] = st + 1;
            ttt_write_obs(L, e);
        } else {
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)(st + 1));
            L.env.steps[e] = 0;
            L.env.episode[e] += 1;
            ttt_fresh(L, e);
        }
    }
    L.env.r_action[rec] = a;
    L.env.r_reward[rec] = reward;
    L.env.r_root[rec] = L.root[e];
    L.env.r_done[rec] = done ? 1 : 0;
    for (int i = 0; i < A; i++) L.env.r_pi[rec * A + i] = L.pi[(size_t)e * A + i];
    if (e == 0) {
        atomicAdd(&L.env.counters[0], (unsigned long long)L.B);
        atomicAdd(&L.env.counters[1], (unsigned long long)L.B * (unsigned long long)L.sims);
    }
}

}  // namespace mz

// mz_env.h -- device-resident environments and the self-play record ring (run_self_play's bookkeeping,
// pipeline.py:83-113) so that a whole lock-step move -- search, action sampling, env.step, trajectory record,
// auto-reset -- stays in HBM.  One thread per environment: these kernels move a few hundred bytes per env per move,
// they are launch-latency-sized, not bandwidth-sized.
//
//   CartPole-v1   : gym 0.23.1 equations (un-vendored dependency of the reference, requirements.txt:7),
//                   float64 state, float32 observation, StackFrameAndAction(4) rows [obs_t-k, (a_t-k + 1)/A] newest
//                   first (gym_env.py:306-353), players 1/1 and an all-true mask (gym_env.py:356-365), TimeLimit 500.
//   TicTacToe /   : BoardGameEnv semantics (games/env.py:117-154,242-310; games/tictactoe.py:33-77 == games/gomoku.py:72-116):
//   Gomoku          N x N board, per-player own-stone history planes (stack 4), resign action N*N, win test (num_to_win
//                   in a row) through the last move only, player switches only if the game is not over.
//   Synthetic     : stand-in for the Atari emulator (ale-py is an absent third-party dependency): every step draws a fresh
//                   U[0,1) observation from Philox, reward 0, episode ends every 1000 steps.
#pragma once
#include "mz_device.h"

namespace mz {

constexpr int ENV_CARTPOLE = 1;
constexpr int ENV_TICTACTOE = 2;
constexpr int ENV_GOMOKU = 3;
constexpr int ENV_SYNTHETIC = 4;

struct EnvState {
    int kind, B, A, D, ring_len;
    int bn, nn, win;       // board games: side length, points (bn * bn), stones in a row to win
    double* cp_state;      // [B][4]
    int* steps;            // [B] steps in the current episode
    unsigned int* episode; // [B] episode index (keys the reset RNG)
    double* init_state;    // [B][4] optional externally supplied reset states (tests)
    signed char* board;    // [B][nn]
    signed char* planes;   // [B][2][4][nn]
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
    long long* ep_start;   // [B] absolute move index of the first step of the env's open trajectory (device epilogue)
};

struct EnvLaunch {
    EnvState env;
    int B;
    unsigned long long seed;
    int use_init;
    double temperature;  // >= 0: constant; < 0: the board game's own schedule (1.0 for the first 6 (TicTacToe) / 30 (Gomoku) moves, then 0.1; config.py:236-249)
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
                    e.r_reward, e.r_pi, e.r_root, e.r_player, e.r_done, e.counters, e.ep_start};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    e = EnvState{};
}

inline hipError_t env_alloc(EnvState& e, int kind, int B, int A, int D, int ring_len, int board_n = 3, int num_to_win = 3) {
    env_free(e);
    e.kind = kind; e.B = B; e.A = A; e.D = D; e.ring_len = ring_len;
    e.bn = board_n; e.nn = board_n * board_n; e.win = num_to_win;
    hipError_t r;
#define MZ_ALLOC(ptr, bytes)                                  \
    if ((r = hipMalloc(&(ptr), (bytes))) != hipSuccess) return r; \
    if ((r = hipMemset((ptr), 0, (bytes))) != hipSuccess) return r;
    MZ_ALLOC(e.cp_state, (size_t)B * 4 * sizeof(double));
    MZ_ALLOC(e.steps, (size_t)B * sizeof(int));
    MZ_ALLOC(e.episode, (size_t)B * sizeof(unsigned int));
    MZ_ALLOC(e.init_state, (size_t)B * 4 * sizeof(double));
    MZ_ALLOC(e.board, (size_t)B * e.nn);
    MZ_ALLOC(e.planes, (size_t)B * 8 * e.nn);
    MZ_ALLOC(e.player, (size_t)B * sizeof(int));
    MZ_ALLOC(e.r_obs, (size_t)ring_len * B * D * sizeof(float));
    MZ_ALLOC(e.r_action, (size_t)ring_len * B * sizeof(int));
    MZ_ALLOC(e.r_reward, (size_t)ring_len * B * sizeof(float));
    MZ_ALLOC(e.r_pi, (size_t)ring_len * B * A * sizeof(double));
    MZ_ALLOC(e.r_root, (size_t)ring_len * B * sizeof(double));
    MZ_ALLOC(e.r_player, (size_t)ring_len * B * sizeof(int));
    MZ_ALLOC(e.r_done, (size_t)ring_len * B);
    MZ_ALLOC(e.counters, 4 * sizeof(unsigned long long));
    MZ_ALLOC(e.ep_start, (size_t)B * sizeof(long long));
#undef MZ_ALLOC
    // hipMemset on device memory returns before the fill has run, and it runs on the NULL stream -- which the planner's non-blocking
    // stream does not wait for: a late fill zeroed what k_env_reset had just written (player ids 0 -> the board step indexes the
    // history planes out of range).  Seen as a memory fault when two processes share a GPU (round 4); a latent race everywhere.
    return hipDeviceSynchronize();
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

// ---- board games (TicTacToe, Gomoku) ----
__device__ inline void board_write_obs(const EnvLaunch& L, int e) {
    // [X_t, Y_t, X_t-1, Y_t-1, ..., C] from the side to move (games/env.py:242-271)
    const int nn = L.env.nn;
    float* o = L.obs + (size_t)e * 9 * nn;
    const signed char* pl = L.env.planes + (size_t)e * 8 * nn;
    const int me = L.env.player[e], opp = 3 - me;
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < nn; i++) {
            o[(2 * t) * nn + i] = (float)pl[((me - 1) * 4 + t) * nn + i];
            o[(2 * t + 1) * nn + i] = (float)pl[((opp - 1) * 4 + t) * nn + i];
        }
    for (int i = 0; i < nn; i++) o[8 * nn + i] = me == 1 ? 1.0f : 0.0f;
    L.cur[e] = me;
    L.opp[e] = opp;
}

__device__ inline void board_fresh(const EnvLaunch& L, int e) {
    const int nn = L.env.nn;
    for (int i = 0; i < nn; i++) L.env.board[(size_t)e * nn + i] = 0;
    for (int i = 0; i < 8 * nn; i++) L.env.planes[(size_t)e * 8 * nn + i] = 0;
    for (int a = 0; a <= nn; a++) L.mask[(size_t)e * (nn + 1) + a] = 1;
    L.env.player[e] = 1;
    board_write_obs(L, e);
}

__device__ inline int board_line(const signed char* b, int n, int r, int c, int dr, int dc, int colour) {
    int k = 0;
    r += dr; c += dc;
    while (r >= 0 && r < n && c >= 0 && c < n && b[r * n + c] == colour) { k++; r += dr; c += dc; }
    return k;
}

// ---- synthetic frames: one thread per 4 observation values, keyed by (env, episode, step) ----
__global__ void k_env_synth_obs(const EnvLaunch L) {
    const int D4 = (L.env.D + 3) >> 2;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)L.B * D4) return;
    const int e = (int)(i / D4), q = (int)(i - (size_t)e * D4);
    Philox g(L.seed, (unsigned)e, L.env.episode[e] * 1000u + (unsigned)L.env.steps[e], 0x50000000u + (unsigned)q);
    uint32_t v[4];
    g.round4(v);
    float* o = L.obs + (size_t)e * L.env.D + 4 * q;
    for (int k = 0; k < 4 && 4 * q + k < L.env.D; k++) o[k] = (float)(v[k] >> 8) * (1.0f / 16777216.0f);
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
    } else if (L.env.kind == ENV_SYNTHETIC) {
        for (int a = 0; a < L.env.A; a++) L.mask[(size_t)e * L.env.A + a] = 1;
        L.cur[e] = 1; L.opp[e] = 1;
    } else {
        board_fresh(L, e);
    }
}

// before the search of a move: temperature for this move and the "obs / player before acting" half of the record
__device__ inline void env_pre_one(const EnvLaunch& L, int e) {
    double T = L.temperature;
    if (T < 0.0) T = L.env.steps[e] < (L.env.kind == ENV_GOMOKU ? 30 : 6) ? 1.0 : 0.1;  // config.py:236-249
    L.temp_out[e] = T;  // (the observation half of the record is one device-to-device copy enqueued by the host)
    L.env.r_player[(size_t)L.slot * L.B + e] = L.cur[e];
}

__global__ void k_env_pre(const EnvLaunch L) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < L.B) env_pre_one(L, e);
}

// one move's record (pipeline.py:106-113) -- written by the lane that ran the search's finish step, so that it reads
// back its own pi / root stores; loads are batched ahead of the stores (they may alias as far as the compiler knows)
__device__ inline void env_record(const EnvLaunch& L, int e, int a, float reward, bool done) {
    const int A = L.env.A;
    const size_t rec = (size_t)L.slot * L.B + e;
    L.env.r_action[rec] = a;
    L.env.r_reward[rec] = reward;
    L.env.r_root[rec] = L.root[e];
    L.env.r_done[rec] = done ? 1 : 0;
    for (int i0 = 0; i0 < A; i0 += 8) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = i0 + k < A ? L.pi[(size_t)e * A + i0 + k] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (i0 + k < A) L.env.r_pi[rec * A + i0 + k] = t[k];
    }
    if (e == 0) {
        atomicAdd(&L.env.counters[0], (unsigned long long)L.B);
        atomicAdd(&L.env.counters[1], (unsigned long long)L.B * (unsigned long long)L.sims);
    }
}

__device__ inline int group_or16(int v) {
    v |= __shfl_xor(v, 1, 16); v |= __shfl_xor(v, 2, 16); v |= __shfl_xor(v, 4, 16); v |= __shfl_xor(v, 8, 16);
    return v;
}

// Stones of `colour` in a row from (r, c) exclusive in direction (dr, dc), on a bitboard (bit = row * n + column)
__device__ inline int board_line_bits(unsigned int stones, int n, int r, int c, int dr, int dc) {
    int k = 0;
    r += dr; c += dc;
    while (r >= 0 && r < n && c >= 0 && c < n && ((stones >> (r * n + c)) & 1u)) { k++; r += dr; c += dc; }
    return k;
}

// BoardGameEnv.step for boards of at most 16 points (TicTacToe): lane i owns point i, the whole pre-move state is fetched in
// ONE batch of loads, the win test runs on a bitboard built by a ballot, and everything is stored afterwards -- two global
// round trips per move instead of the strided version's dozen.  Same results as board_step_group (tests: device env vs the
// oracle env).  `lane` 0..15; the group is 16 consecutive lanes of one wave, all of them active.
__device__ inline void board_step_small(const EnvLaunch& L, int e, int lane, int a, float& reward, bool& done) {
    const int n = L.env.bn, nn = L.env.nn;
    signed char* b = L.env.board + (size_t)e * nn;
    signed char* pl = L.env.planes + (size_t)e * 8 * nn;
    const int me = L.env.player[e], opp = 3 - me, st = L.env.steps[e];
    signed char* mine = pl + (me - 1) * 4 * nn;
    const signed char* theirs = pl + (opp - 1) * 4 * nn;
    const bool cell = lane < nn;
    const int i = cell ? lane : 0;
    const signed char bi = b[i];
    const signed char m0 = mine[i], m1 = mine[nn + i], m2 = mine[2 * nn + i];
    const signed char t0 = theirs[i], t1 = theirs[nn + i], t2 = theirs[2 * nn + i], t3 = theirs[3 * nn + i];
    const int base = (int)(__lane_id() & 48u);
    const unsigned int my_stones = (unsigned int)(__ballot(cell && bi == me) >> base) & 0xffffu;
    const unsigned int open = (unsigned int)(__ballot(cell && bi == 0 && i != a) >> base) & 0xffffu;  // empty points other than the one just played
    int winner = 0;
    reward = 0.0f;
    if (a == nn) {  // resign (games/env.py:134-136)
        reward = -1.0f;
        winner = opp;
    } else if (st >= (L.env.win - 1) * 2) {  // games/tictactoe.py:37-38 with the pre-increment step count
        const int r = a / n, c = a % n;
        const int dirs[4][2] = {{0, 1}, {1, 0}, {1, 1}, {-1, 1}};
        for (int d = 0; d < 4; d++)
            if (1 + board_line_bits(my_stones, n, r, c, dirs[d][0], dirs[d][1]) + board_line_bits(my_stones, n, r, c, -dirs[d][0], -dirs[d][1]) >= L.env.win) winner = me;
        if (winner) reward = 1.0f;
    }
    done = winner != 0 || open == 0;
    float* o = L.obs + (size_t)e * 9 * nn;
    if (!done) {
        const signed char nm0 = (signed char)((bi == me || i == a) ? 1 : 0);  // the mover's history after this move: [stones now, m0, m1, m2]
        if (cell) {
            mine[i] = nm0; mine[nn + i] = m0; mine[2 * nn + i] = m1; mine[3 * nn + i] = m2;
            // observation of the next side to move (games/env.py:242-271): its own history, the mover's, the colour plane
            o[0 * nn + i] = (float)t0; o[1 * nn + i] = (float)nm0;
            o[2 * nn + i] = (float)t1; o[3 * nn + i] = (float)m0;
            o[4 * nn + i] = (float)t2; o[5 * nn + i] = (float)m1;
            o[6 * nn + i] = (float)t3; o[7 * nn + i] = (float)m2;
            o[8 * nn + i] = opp == 1 ? 1.0f : 0.0f;
            if (i == a) { b[i] = (signed char)me; L.mask[(size_t)e * (nn + 1) + a] = 0; }
        }
        if (lane == 0) {
            L.env.player[e] = opp;
            L.env.steps[e] = st + 1;
            L.cur[e] = opp;
            L.opp[e] = me;
        }
    } else {  // auto-reset (pipeline.py:111-113)
        if (cell) {
            b[i] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) { pl[t * nn + i] = 0; o[t * nn + i] = 0.0f; }
            o[8 * nn + i] = 1.0f;
            L.mask[(size_t)e * (nn + 1) + i] = 1;
        }
        if (lane == 0) {
            L.mask[(size_t)e * (nn + 1) + nn] = 1;
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)(st + 1));
            L.env.steps[e] = 0;
            L.env.episode[e] += 1;
            L.env.player[e] = 1;
            L.cur[e] = 1;
            L.opp[e] = 2;
        }
    }
}

// BoardGameEnv.step (games/env.py:117-154) by the 16 lanes of an env's group (16 consecutive lanes of one wave).  Every
// lane reads the pre-move state; the bulk moves (history shift, next observation, reset) are strided over the group.
// Lanes of a wave run in lock-step and vector memory operations of a wave are served in issue order, so a location is
// only ever read by instructions issued BEFORE the one that overwrites it (the history shift walks downwards for that).
__device__ inline void board_step_group(const EnvLaunch& L, int e, int lane, int a, float& reward, bool& done) {
    const int n = L.env.bn, nn = L.env.nn;
    signed char* b = L.env.board + (size_t)e * nn;
    signed char* pl = L.env.planes + (size_t)e * 8 * nn;
    const int me = L.env.player[e], opp = 3 - me, st = L.env.steps[e];
    signed char* mine = pl + (me - 1) * 4 * nn;
    const signed char* theirs = pl + (opp - 1) * 4 * nn;
    int winner = 0;
    reward = 0.0f;
    if (a == nn) {  // resign (games/env.py:134-136)
        reward = -1.0f;
        winner = opp;
    } else if (st >= (L.env.win - 1) * 2) {  // games/tictactoe.py:37-38, games/gomoku.py:76-77 with the pre-increment step count
        // lanes 0..7 walk one ray each from the new stone (which the rays do not include)
        const int r = a / n, c = a % n, d = lane & 3, sg = (lane & 4) ? -1 : 1;
        const int dr = sg * (d == 0 ? 0 : d == 3 ? -1 : 1), dc = sg * (d == 1 ? 0 : 1);  // (0,1) (1,0) (1,1) (-1,1)
        const int k = lane < 8 ? board_line(b, n, r, c, dr, dc, me) : 0;
        const int k2 = __shfl_xor(k, 4, 16);
        if (group_or16(lane < 4 && 1 + k + k2 >= L.env.win)) { winner = me; reward = 1.0f; }
    }
    int open = 0;  // an empty point other than the one just played
    for (int i = lane; i < nn; i += 16) open |= (b[i] == 0 && i != a) ? 1 : 0;
    done = winner != 0 || !group_or16(open);
    float* o = L.obs + (size_t)e * 9 * nn;
    if (!done) {
        // observation of the next side to move (games/env.py:242-271): its own history (unchanged by this move), the
        // mover's history as it will be after the shift below, and the colour plane
        for (int idx = lane; idx < 4 * nn; idx += 16) {
            const int t = idx / nn, i = idx - t * nn;
            const signed char x = theirs[idx];
            const signed char y = t > 0 ? mine[idx - nn] : (signed char)((b[i] == me || i == a) ? 1 : 0);
            o[(2 * t) * nn + i] = (float)x;
            o[(2 * t + 1) * nn + i] = (float)y;
        }
        for (int i = lane; i < nn; i += 16) o[8 * nn + i] = opp == 1 ? 1.0f : 0.0f;
        for (int j = (4 * nn + 15) / 16 - 1; j >= 0; j--) {  // mine[t] <- mine[t - 1], mine[0] <- the mover's stones
            const int idx = j * 16 + lane;
            if (idx < 4 * nn) {
                const int t = idx / nn, i = idx - t * nn;
                const signed char v = t > 0 ? mine[idx - nn] : (signed char)((b[i] == me || i == a) ? 1 : 0);
                mine[idx] = v;
            }
        }
        if (lane == 0) {
            L.mask[(size_t)e * (nn + 1) + a] = 0;
            b[a] = (signed char)me;
            L.env.player[e] = opp;
            L.env.steps[e] = st + 1;
            L.cur[e] = opp;
            L.opp[e] = me;
        }
    } else {  // auto-reset (pipeline.py:111-113): board_fresh, strided
        for (int i = lane; i < nn; i += 16) b[i] = 0;
        for (int i = lane; i < 8 * nn; i += 16) { pl[i] = 0; o[i] = 0.0f; }
        for (int i = lane; i < nn; i += 16) o[8 * nn + i] = 1.0f;
        for (int i = lane; i <= nn; i += 16) L.mask[(size_t)e * (nn + 1) + i] = 1;
        if (lane == 0) {
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)(st + 1));
            L.env.steps[e] = 0;
            L.env.episode[e] += 1;
            L.env.player[e] = 1;
            L.cur[e] = 1;
            L.opp[e] = 2;
        }
    }
}

// after the search: env.step(action), record, auto-reset (pipeline.py:106-113)
__device__ inline void env_step_one(const EnvLaunch& L, int e) {  // CartPole / synthetic: one lane per env
    const int a = L.action[e];
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
    } else if (L.env.kind == ENV_SYNTHETIC) {
        const int st = L.env.steps[e] + 1;
        done = st >= 1000;
        if (done) {
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)st);
            L.env.episode[e] += 1;
        }
        L.env.steps[e] = done ? 0 : st;  // the next observation is drawn by k_env_synth_obs
    }
    env_record(L, e, a, reward, done);
}
// ---- CartPole inside the search kernel (fused self-play move): the step's inputs -- physics state, step count, the three newest
// observation rows -- do not depend on the action, so lane 0 of the env requests them BEFORE the play-policy phase of the search
// (one batch of independent loads whose round trip hides under that phase), and the action / policy / root value arrive in
// registers instead of being read back from the global stores that have just been issued.  env_step_one fetched everything after
// the action was known: four dependent global round trips at the tail of a kernel whose other lanes had nothing left to do.
struct CartPolePre {
    double s[4];
    float o[15];
    int steps;
};
__device__ __forceinline__ void cartpole_prefetch(const EnvLaunch& L, int e, CartPolePre& pre) {
#pragma unroll
    for (int i = 0; i < 4; i++) pre.s[i] = L.env.cp_state[e * 4 + i];
    const float* o = L.obs + (size_t)e * 20;
#pragma unroll
    for (int i = 0; i < 15; i++) pre.o[i] = o[i];
    pre.steps = L.env.steps[e];
}
// == env_step_one (CartPole branch) + env_record with everything passed in; `pi`: the env's policy (float64, A = 2)
__device__ __forceinline__ void cartpole_step_prefetched(const EnvLaunch& L, int e, CartPolePre& pre, int a, double root, const double* pi) {
    double s[4] = {pre.s[0], pre.s[1], pre.s[2], pre.s[3]};
    const bool term = cartpole_physics(s, a);
    const int st = pre.steps + 1;
    const bool done = term || st >= 500;
    if (!done) {
        L.env.steps[e] = st;
        for (int i = 0; i < 4; i++) L.env.cp_state[e * 4 + i] = s[i];
        float* o = L.obs + (size_t)e * 20;  // appendleft (gym_env.py:317-324)
#pragma unroll
        for (int i = 0; i < 15; i++) o[5 + i] = pre.o[i];
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
    const size_t rec = (size_t)L.slot * L.B + e;
    L.env.r_action[rec] = a;
    L.env.r_reward[rec] = 1.0f;
    L.env.r_root[rec] = root;
    L.env.r_done[rec] = done ? 1 : 0;
    L.env.r_pi[rec * 2] = pi[0];
    L.env.r_pi[rec * 2 + 1] = pi[1];
    if (e == 0) {
        atomicAdd(&L.env.counters[0], (unsigned long long)L.B);
        atomicAdd(&L.env.counters[1], (unsigned long long)L.B * (unsigned long long)L.sims);
    }
}

// ---- small boards (TicTacToe) inside the search kernel: as for CartPole, everything the step reads except the action is requested
// BEFORE the play-policy phase -- lane i's point of the board and of all eight history planes (both players: whose move it is comes
// back with the same batch), player, step count -- and action / policy / root value arrive in registers.  board_step_small paid
// four dependent global round trips (action, player -> planes, record read-back) at the kernel's tail.
struct BoardSmallPre {
    signed char b, pl[8];
    int me, st;
};
__device__ __forceinline__ void board_small_prefetch(const EnvLaunch& L, int e, int lane, BoardSmallPre& pre) {
    const int nn = L.env.nn, i = lane < nn ? lane : 0;
    pre.b = L.env.board[(size_t)e * nn + i];
    const signed char* pl = L.env.planes + (size_t)e * 8 * nn;
#pragma unroll
    for (int t = 0; t < 8; t++) pre.pl[t] = pl[t * nn + i];
    pre.me = L.env.player[e];
    pre.st = L.env.steps[e];
}
// == board_step_small + env_record on prefetched state; `pi`: the env's policy row (float64 [A]); all 16 lanes of the env call it
__device__ __forceinline__ void board_step_small_prefetched(const EnvLaunch& L, int e, int lane, const BoardSmallPre& pre, int a, double root, const double* pi) {
    const int n = L.env.bn, nn = L.env.nn, A = L.env.A;
    signed char* b = L.env.board + (size_t)e * nn;
    signed char* pl = L.env.planes + (size_t)e * 8 * nn;
    const int me = pre.me, opp = 3 - me, st = pre.st;
    signed char* mine = pl + (me - 1) * 4 * nn;
    const bool cell = lane < nn;
    const int i = cell ? lane : 0;
    const signed char bi = pre.b;
    const bool black = me == 1;  // planes 0-3: black's history, 4-7: white's
    const signed char m0 = black ? pre.pl[0] : pre.pl[4], m1 = black ? pre.pl[1] : pre.pl[5], m2 = black ? pre.pl[2] : pre.pl[6];
    const signed char t0 = black ? pre.pl[4] : pre.pl[0], t1 = black ? pre.pl[5] : pre.pl[1], t2 = black ? pre.pl[6] : pre.pl[2],
                      t3 = black ? pre.pl[7] : pre.pl[3];
    const int base = (int)(__lane_id() & 48u);
    const unsigned int my_stones = (unsigned int)(__ballot(cell && bi == me) >> base) & 0xffffu;
    const unsigned int open = (unsigned int)(__ballot(cell && bi == 0 && i != a) >> base) & 0xffffu;  // empty points other than the one just played
    int winner = 0;
    float reward = 0.0f;
    if (a == nn) {  // resign (games/env.py:134-136)
        reward = -1.0f;
        winner = opp;
    } else if (st >= (L.env.win - 1) * 2) {  // games/tictactoe.py:37-38 with the pre-increment step count
        const int r = a / n, c = a % n;
        const int dirs[4][2] = {{0, 1}, {1, 0}, {1, 1}, {-1, 1}};
        for (int d = 0; d < 4; d++)
            if (1 + board_line_bits(my_stones, n, r, c, dirs[d][0], dirs[d][1]) + board_line_bits(my_stones, n, r, c, -dirs[d][0], -dirs[d][1]) >= L.env.win) winner = me;
        if (winner) reward = 1.0f;
    }
    const bool done = winner != 0 || open == 0;
    float* o = L.obs + (size_t)e * 9 * nn;
    if (!done) {
        const signed char nm0 = (signed char)((bi == me || i == a) ? 1 : 0);  // the mover's history after this move: [stones now, m0, m1, m2]
        if (cell) {
            mine[i] = nm0; mine[nn + i] = m0; mine[2 * nn + i] = m1; mine[3 * nn + i] = m2;
            // observation of the next side to move (games/env.py:242-271): its own history, the mover's, the colour plane
            o[0 * nn + i] = (float)t0; o[1 * nn + i] = (float)nm0;
            o[2 * nn + i] = (float)t1; o[3 * nn + i] = (float)m0;
            o[4 * nn + i] = (float)t2; o[5 * nn + i] = (float)m1;
            o[6 * nn + i] = (float)t3; o[7 * nn + i] = (float)m2;
            o[8 * nn + i] = opp == 1 ? 1.0f : 0.0f;
            if (i == a) { b[i] = (signed char)me; L.mask[(size_t)e * (nn + 1) + a] = 0; }
        }
        if (lane == 0) {
            L.env.player[e] = opp;
            L.env.steps[e] = st + 1;
            L.cur[e] = opp;
            L.opp[e] = me;
        }
    } else {  // auto-reset (pipeline.py:111-113)
        if (cell) {
            b[i] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) { pl[t * nn + i] = 0; o[t * nn + i] = 0.0f; }
            o[8 * nn + i] = 1.0f;
            L.mask[(size_t)e * (nn + 1) + i] = 1;
        }
        if (lane == 0) {
            L.mask[(size_t)e * (nn + 1) + nn] = 1;
            atomicAdd(&L.env.counters[2], 1ULL);
            atomicAdd(&L.env.counters[3], (unsigned long long)(st + 1));
            L.env.steps[e] = 0;
            L.env.episode[e] += 1;
            L.env.player[e] = 1;
            L.cur[e] = 1;
            L.opp[e] = 2;
        }
    }
    // the move's record (env_record), from registers / the policy's LDS row: lane j stores policy entry j
    const size_t rec = (size_t)L.slot * L.B + e;
    if (lane < A) L.env.r_pi[rec * A + lane] = pi[lane];
    if (lane == 0) {
        L.env.r_action[rec] = a;
        L.env.r_reward[rec] = reward;
        L.env.r_root[rec] = root;
        L.env.r_done[rec] = done ? 1 : 0;
        if (e == 0) {
            atomicAdd(&L.env.counters[0], (unsigned long long)L.B);
            atomicAdd(&L.env.counters[1], (unsigned long long)L.B * (unsigned long long)L.sims);
        }
    }
}

// env.step of one env by its 16-lane group (`lane` 0..15; all 16 lanes call this)
__device__ inline void env_step_group(const EnvLaunch& L, int e, int lane) {
    if (L.env.kind != ENV_TICTACTOE && L.env.kind != ENV_GOMOKU) {
        if (lane == 0) env_step_one(L, e);
        return;
    }
    const int a = __shfl(lane == 0 ? L.action[e] : 0, 0, 16);  // lane 0 may have stored it a moment ago (fused search)
    float reward;
    bool done;
    if (L.env.nn <= 16) board_step_small(L, e, lane, a, reward, done);
    else board_step_group(L, e, lane, a, reward, done);
    if (lane == 0) env_record(L, e, a, reward, done);
}

__global__ void k_env_step(const EnvLaunch L) {  // 16 threads per env
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if ((g >> 4) < L.B) env_step_group(L, g >> 4, g & 15);
}

// ---------------------------------------------------------------------------------------------------------
// Device epilogue of run_self_play (pipeline.py:118-165): the record ring is each env's open trajectory; after every
// lock-step move one wave per env checks the two conditions of the reference's loop body, in its order --
//   (1) not a board game and the open trajectory holds acc_seq_length + unroll_steps + td_steps steps: its first
//       acc_seq_length steps become items, with n-step targets computed over the whole open trajectory (:118-142);
//   (2) the episode ended: every remaining step becomes an item (:144-165), Monte-Carlo returns for board games (:676-707)
// -- and writes the items (state, K-step action / reward / value / policy windows with the absorbing padding of
// make_unroll_sequence :710-767, priority |root value - target| :129,156) straight into the HBM replay ring
// (replay.py:67-75: slot = num_added % capacity).  The host only reads the counter.  Arithmetic is the reference's:
// float64 left-to-right sums with host-computed discount ** i, rounded to float32 where the reference's np.array(...,
// dtype=np.float32) does.
// ---------------------------------------------------------------------------------------------------------
struct ReplayRing {
    long long capacity;
    float* state;         // [capacity][D]
    signed char* action;  // [capacity][K] int8 -- or, when action16 is set (num_actions > 128), the same buffer as int16
    int action16;
    float* pi_prob;       // [capacity][K][A]
    float* value;         // [capacity][K]
    float* reward;        // [capacity][K]
    float* priority;      // [capacity]
    long long* num_added; // COMMITTED items: published by k_epi_publish after every k_epilogue launch, when all slots below it are filled
    long long* ctr;       // planner-owned: [0] reserved write cursor, [1] items of the move being written (k_epi_scan)
    int* off;             // planner-owned [B]: first slot of env e's items of this move, relative to ctr[0] (k_epi_scan: prefix sum in env order)
    int* origin;          // optional [capacity]: env that produced the item (tests), or null
    int acc, K, td, board;
    double pw[34];        // discount ** i, i = 0..td (host libm pow == Python float pow)
};

struct EpiLaunch {
    EnvState env;
    ReplayRing ring;
    int B;
    long long move_abs;   // absolute index (since mz_selfplay_reset) of the move that has just been recorded
};

// z[t], t in [0, T): target value of every step of the open trajectory (compute_n_step_target :632-673 /
// compute_mc_return_target :676-707); rec(i) = record index of trajectory position i
__device__ inline void epi_emit(const EpiLaunch& E, int e, long long start, int T, int n, long long base, long long keep_from, double* z, int lane) {
    const EnvState& V = E.env;
    const ReplayRing& R = E.ring;
    const int B = E.B, A = V.A, D = V.D, K = R.K;
    auto rec = [&](int i) { return (size_t)((start + i) % V.ring_len) * B + e; };
    if (R.board) {
        const double fr = (double)V.r_reward[rec(T - 1)];
        const int fp = V.r_player[rec(T - 1)];
        for (int t = lane; t < T; t += 64) z[t] = fr != 0.0 ? (V.r_player[rec(t)] == fp ? fr : -fr) : 0.0;
    } else {
        for (int t = lane; t < T; t += 64) {
            double acc = 0.0;
            for (int i = 0; i < R.td; i++) acc = acc + R.pw[i] * (t + i < T ? (double)V.r_reward[rec(t + i)] : 0.0);
            z[t] = acc + R.pw[R.td] * (t + R.td < T ? V.r_root[rec(t + R.td)] : 0.0);
        }
    }
    __syncthreads();
    // `base`: this env's slots, reserved by k_epi_scan in ENV ORDER (round 3 reserved with one atomicAdd per env: which env got which
    // slot then depended on workgroup scheduling, and with it every later replay draw -- the same seed gave different runs); the
    // count is published (R.num_added) only when the whole launch has filled its slots: see k_epi_publish
    // the n items are written with the wave's lanes spread over (item, element) pairs: a flush emits acc_seq_length items at once.
    // `keep_from`: when ONE move emits more items than the ring holds (a small ring, many envs finishing together), the items below it
    // would be overwritten by later items of the same launch -- by another workgroup, in no defined order -- so they are not written at
    // all: the ring ends up with exactly the last `capacity` items in order, run to run the same
    for (int j = lane; j < n * D; j += 64) {
        const int t = j / D, i = j - t * D;
        if (base + t < keep_from) continue;
        R.state[(size_t)((base + t) % R.capacity) * D + i] = V.r_obs[rec(t) * D + i];
    }
    for (int j = lane; j < n * K; j += 64) {
        const int t = j / K, k = j - t * K, idx = t + k;
        if (base + t < keep_from) continue;
        const size_t o = (size_t)((base + t) % R.capacity) * K + k;
        const bool real = idx < T;  // past the end: absorbing step (action 0, reward 0, value 0, uniform policy)
        if (R.action16) reinterpret_cast<short*>(R.action)[o] = real ? (short)V.r_action[rec(idx)] : (short)0;
        else R.action[o] = real ? (signed char)V.r_action[rec(idx)] : (signed char)0;
        R.reward[o] = real ? V.r_reward[rec(idx)] : 0.0f;
        R.value[o] = real ? (float)z[idx] : 0.0f;
    }
    for (int j = lane; j < n * K * A; j += 64) {
        const int t = j / (K * A), r = j - t * K * A, k = r / A, a = r - k * A, idx = t + k;
        if (base + t < keep_from) continue;
        R.pi_prob[((size_t)((base + t) % R.capacity) * K + k) * A + a] = idx < T ? (float)V.r_pi[rec(idx) * A + a] : (float)(1.0 / (double)A);
    }
    for (int t = lane; t < n; t += 64) {
        if (base + t < keep_from) continue;
        const size_t slot = (size_t)((base + t) % R.capacity);
        const double d = V.r_root[rec(t)] - z[t];
        R.priority[slot] = (float)(d < 0.0 ? -d : d);
        if (R.origin) R.origin[slot] = e;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_epilogue(const EpiLaunch E) {
    extern __shared__ double z[];  // [ring_len]
    const int e = blockIdx.x, lane = threadIdx.x;
    const EnvState& V = E.env;
    const ReplayRing& R = E.ring;
    long long start = V.ep_start[e];
    int len = (int)(E.move_abs + 1 - start);
    const bool done = V.r_done[(size_t)(E.move_abs % V.ring_len) * E.B + e] != 0;
    const bool flush = !R.board && len == R.acc + R.K + R.td;
    if (flush || done) {
        long long base = R.ctr[0] + (long long)R.off[e];
        const long long keep_from = R.ctr[0] + R.ctr[1] - (long long)R.capacity;  // (ctr[1]: this move's item count, k_epi_scan)
        if (flush) {
            epi_emit(E, e, start, len, R.acc, base, keep_from, z, lane);
            start += R.acc;
            len -= R.acc;
            base += R.acc;
        }
        if (done) {
            epi_emit(E, e, start, len, len, base, keep_from, z, lane);
            start = E.move_abs + 1;
        }
        if (lane == 0) V.ep_start[e] = start;
    }
}

// How many items every env emits after this move (the same tests as k_epilogue), as an exclusive prefix sum in env order: off[e],
// and the move's total in ctr[1].  One workgroup: thread t owns a contiguous run of envs, the thread totals meet in an LDS scan.
__global__ __launch_bounds__(1024) void k_epi_scan(const EpiLaunch E) {
    __shared__ int part[1024];
    const EnvState& V = E.env;
    const ReplayRing& R = E.ring;
    const int t = threadIdx.x, per = (E.B + 1023) / 1024, e0 = t * per, e1 = e0 + per < E.B ? e0 + per : E.B;
    int sum = 0;
    for (int e = e0; e < e1; e++) {
        int len = (int)(E.move_abs + 1 - V.ep_start[e]);
        const bool done = V.r_done[(size_t)(E.move_abs % V.ring_len) * E.B + e] != 0;
        const bool flush = !R.board && len == R.acc + R.K + R.td;
        int n = 0;
        if (flush) { n += R.acc; len -= R.acc; }
        if (done) n += len;
        R.off[e] = sum;  // (relative to this thread's run; the run's base is added below)
        sum += n;
    }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan of the thread totals
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int base = part[t] - sum;
    for (int e = e0; e < e1; e++) R.off[e] += base;
    if (t == 1023) R.ctr[1] = (long long)part[1023];
}

// Publish: the host (replay.num_added / sample on another stream) may only see a count whose slots are all filled.  Slots are
// reserved by k_epi_scan (R.off, R.ctr[1]); this one-thread kernel, next on the same stream -- i.e. after every workgroup of the
// epilogue has finished and its writes are visible -- copies the reserved cursor to the committed counter the host reads.
// (A last-workgroup-done counter inside k_epilogue did the same with 4096 same-address atomics per move: -9 % on the C2
// env-steps-into-replay rate, measured.)
__global__ void k_epi_publish(const ReplayRing R) {
    const long long c = R.ctr[0] + R.ctr[1];
    R.ctr[0] = c;
    *reinterpret_cast<volatile long long*>(R.num_added) = c;
}

}  // namespace mz

// The C ABI used from plain C++, no Python and no torch in the process: every entry point of
// include/marl_amd.h is called for all agent counts, feature sets and a few layouts (odd env
// counts included), then the argument checks are driven through their failure paths, and the
// headline batch is stepped as one handle and as two shards on two streams from one hipGraph.  Prints one
// line per stage and "capi host pass: ok" at the end.  Built by tests/native/Makefile (run by the
// GPU suite, tests/test_gpu_capi_native.py) and, with AddressSanitizer on the host code of driver
// and library, by profiles/diag/capi_host_asan.sh (ASan ends the process at the first host
// memory error; device code is not instrumented).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/marl_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define EXPECT(cond) do { if (!(cond)) { printf("line %d: %s failed (last error: %s)\n", __LINE__, #cond, marl_last_error()); exit(3); } } while (0)

static const char* CRAMPED[] = {"#######", "#CCUCC#", "#O   O#", "#C   C#", "#C=C@C#", "#######"};
static const char* FORCED[] = {"#######", "#CCCUC#", "#O C+U#", "#O+C C#", "#= C C#", "#CCC@C#", "#######"};
static const char* LARGE[] = {"#################", "#C@CC=CCCCCCCUUC#", "#C  C     C    C#", "#C  C COO C    C#",
                              "#C    CCCCC    C#", "#C             C#", "#C   CCCCCC    C#", "#C   CCOOCC C  C#",
                              "#C   C      C  C#", "#CUUCCCCCCC=CC@C#", "#################"};

template <class T>
static T* dalloc(size_t n) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    HIP_OK(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
    return static_cast<T*>(p);
}

static void run_env(const char* const* rows, int height, int A, int feature, int E) {
    marl_env_t* env = nullptr;
    EXPECT(marl_env_create(&env, rows, height, A, feature, 25, E, 77, 1000, 0) == MARL_OK);
    const int D = marl_env_obs_dim(env), N = marl_env_num_rows(env), NP = marl_env_num_pots(env), HW = marl_env_grid_cells(env);
    EXPECT(D > 0 && N == E * A && NP >= 0 && HW > 0 && marl_env_state_words(env) > 0);
    const size_t sbytes = marl_env_state_bytes(env);
    const int K = 3;  // steps per multi-step launch
    uint8_t* state = dalloc<uint8_t>(sbytes);
    float* obs = dalloc<float>((size_t)K * N * D + 4);
    float* rew = dalloc<float>((size_t)K * N);
    uint8_t* term = dalloc<uint8_t>((size_t)K * N);
    uint8_t* trunc = dalloc<uint8_t>((size_t)K * N);
    std::vector<int64_t> h_act((size_t)K * N);
    for (size_t k = 0; k < h_act.size(); ++k) h_act[k] = (int64_t)((k * 2654435761u >> 7) % 7);
    int64_t* act = dalloc<int64_t>(h_act.size());
    HIP_OK(hipMemcpy(act, h_act.data(), h_act.size() * 8, hipMemcpyHostToDevice));
    uint8_t* prio = dalloc<uint8_t>((size_t)K * N);
    int32_t* agents = dalloc<int32_t>((size_t)E * A * 4);
    int32_t* pots = dalloc<int32_t>((size_t)E * (NP ? NP : 1) * 2);
    uint8_t* items = dalloc<uint8_t>((size_t)E * HW);
    int32_t* misc = dalloc<int32_t>((size_t)E * 2);
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    EXPECT(marl_env_reset(env, state, obs, nullptr, nullptr, st) == MARL_OK);
    for (int t = 0; t < 30; ++t)  // runs past max_steps: the self-restart path
        EXPECT(marl_env_step(env, state, act + (t % K) * N, 8, obs, rew, term, trunc, t & 1 ? prio : nullptr, st) == MARL_OK);
    EXPECT(marl_env_featurize(env, state, obs, st) == MARL_OK);
    EXPECT(marl_env_export_state(env, state, agents, pots, items, misc, st) == MARL_OK);
    EXPECT(marl_env_import_state(env, state, agents, pots, items, misc, st) == MARL_OK);
    {
        // a hostile snapshot: coordinates whose product wraps around int32 into the grid's range,
        // far-off and negative positions, codes beyond their fields, items on every cell, a huge step
        // counter.  The library must stay inside its tables (the kernels that follow would fault or
        // write out of range otherwise) and every agent must come out on a floor cell.
        std::vector<int32_t> h_agents((size_t)E * A * 4), h_pots((size_t)E * (NP ? NP : 1) * 2), h_misc((size_t)E * 2);
        std::vector<uint8_t> h_items((size_t)E * HW, 0xFF);
        const int32_t bad[6][2] = {{65536, 65536}, {0x40000000, 4}, {-1, 3}, {2, -7}, {1 << 30, 1 << 30}, {3, 1 << 20}};
        for (size_t k = 0; k < (size_t)E * A; ++k) {
            h_agents[4 * k] = bad[k % 6][0]; h_agents[4 * k + 1] = bad[k % 6][1];
            h_agents[4 * k + 2] = 0x7FFFFFFF; h_agents[4 * k + 3] = -5;
        }
        for (auto& x : h_pots) x = 0x7FFFFFFF;
        for (size_t e = 0; e < (size_t)E; ++e) { h_misc[2 * e] = 0x7FFFFFF0; h_misc[2 * e + 1] = -1; }
        HIP_OK(hipMemcpy(agents, h_agents.data(), h_agents.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(pots, h_pots.data(), h_pots.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(items, h_items.data(), h_items.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(misc, h_misc.data(), h_misc.size() * 4, hipMemcpyHostToDevice));
        EXPECT(marl_env_import_state(env, state, agents, pots, items, misc, st) == MARL_OK);
        EXPECT(marl_env_featurize(env, state, obs, st) == MARL_OK);
        EXPECT(marl_env_export_state(env, state, agents, pots, items, misc, st) == MARL_OK);
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(h_agents.data(), agents, h_agents.size() * 4, hipMemcpyDeviceToHost));
        const int W = (int)strlen(rows[0]);
        for (size_t k = 0; k < (size_t)E * A; ++k) {
            const int r = h_agents[4 * k], c = h_agents[4 * k + 1];
            EXPECT(r >= 0 && r < height && c >= 0 && c < W && (rows[r][c] == ' ' || rows[r][c] == '+'));
            EXPECT(h_agents[4 * k + 2] >= 0 && h_agents[4 * k + 2] < 4 && h_agents[4 * k + 3] >= 0 && h_agents[4 * k + 3] < 4);
        }
        EXPECT(marl_env_reset(env, state, obs, nullptr, nullptr, st) == MARL_OK);
    }
    {
        // hostile reset overrides (a wall cell, a cell off the grid, directions beyond 3): agents must
        // still come out on floor cells with a direction in range
        std::vector<uint8_t> h_spawn((size_t)E * A), h_dir((size_t)E * A);
        for (size_t k = 0; k < h_spawn.size(); ++k) { h_spawn[k] = (k & 1) ? 0 : 255; h_dir[k] = (uint8_t)(4 + k % 250); }
        uint8_t* spawn = dalloc<uint8_t>(h_spawn.size());
        uint8_t* dirs = dalloc<uint8_t>(h_dir.size());
        HIP_OK(hipMemcpy(spawn, h_spawn.data(), h_spawn.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dirs, h_dir.data(), h_dir.size(), hipMemcpyHostToDevice));
        EXPECT(marl_env_reset(env, state, obs, spawn, dirs, st) == MARL_OK);
        EXPECT(marl_env_step(env, state, act, 8, obs, rew, term, trunc, nullptr, st) == MARL_OK);
        EXPECT(marl_env_export_state(env, state, agents, pots, items, misc, st) == MARL_OK);
        HIP_OK(hipStreamSynchronize(st));
        std::vector<int32_t> h_agents((size_t)E * A * 4);
        HIP_OK(hipMemcpy(h_agents.data(), agents, h_agents.size() * 4, hipMemcpyDeviceToHost));
        const int W = (int)strlen(rows[0]);
        for (size_t k = 0; k < (size_t)E * A; ++k) {
            const int r = h_agents[4 * k], c = h_agents[4 * k + 1];
            EXPECT(r >= 0 && r < height && c >= 0 && c < W && (rows[r][c] == ' ' || rows[r][c] == '+'));
            EXPECT(h_agents[4 * k + 2] >= 0 && h_agents[4 * k + 2] < 4 && h_agents[4 * k + 3] >= 0 && h_agents[4 * k + 3] < 4);
        }
        HIP_OK(hipFree(spawn)); HIP_OK(hipFree(dirs));
        EXPECT(marl_env_reset(env, state, obs, nullptr, nullptr, st) == MARL_OK);
    }
    EXPECT(marl_env_set_auto_reset(env, 0) == MARL_OK);
    EXPECT(marl_env_step(env, state, act, 8, obs, rew, term, trunc, nullptr, st) == MARL_OK);
    EXPECT(marl_env_set_auto_reset(env, 1) == MARL_OK);
    if (((size_t)N * D) % 4 == 0)
        EXPECT(marl_env_step_many(env, state, act, 8, K, obs, rew, term, trunc, nullptr, st) == MARL_OK);
    else
        EXPECT(marl_env_step_many(env, state, act, 8, K, obs, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_step_many(env, state, act, 8, 1, obs, rew, term, trunc, nullptr, st) == MARL_OK);
    HIP_OK(hipStreamSynchronize(st));

    // argument checks
    EXPECT(marl_env_step(env, state, act, 3, obs, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_step(env, state, act, 8, obs + 1, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_step(env, nullptr, act, 8, obs, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_step(env, state, nullptr, 8, obs, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_reset(env, state, nullptr, nullptr, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(marl_env_step_many(env, state, act, 8, 0, obs, rew, term, trunc, nullptr, st) == MARL_ERR_INVALID);
    EXPECT(strlen(marl_last_error()) > 0);

    marl_env_destroy(env);
    HIP_OK(hipStreamDestroy(st));
    for (void* p : {(void*)state, (void*)obs, (void*)rew, (void*)term, (void*)trunc, (void*)act, (void*)prio, (void*)agents,
                    (void*)pots, (void*)items, (void*)misc})
        HIP_OK(hipFree(p));
}

// The pipelined form from plain C++: the batch as two env handles whose env_index_base continues
// the numbering, each stepping on a stream of its own, all steps captured into one hipGraph with a
// branch per shard (fork and join through events, no join between steps).  The two shards must
// reproduce the single-handle batch bit for bit; the time per step of both forms is printed.
static void run_pipelined_shards() {
    const int E = 65536, A = 2, K = 16, half = E / 2;
    marl_env_t *whole = nullptr, *sh[2] = {nullptr, nullptr};
    EXPECT(marl_env_create(&whole, CRAMPED, 6, A, MARL_FEAT_GLOBAL_OBS, 1000, E, 9, 0, 0) == MARL_OK);
    for (int k = 0; k < 2; ++k)
        EXPECT(marl_env_create(&sh[k], CRAMPED, 6, A, MARL_FEAT_GLOBAL_OBS, 1000, half, 9, (int64_t)k * half, 0) == MARL_OK);
    const int D = marl_env_obs_dim(whole), N = E * A, Nh = half * A;
    EXPECT(((size_t)Nh * D) % 4 == 0);
    std::vector<int64_t> h_act((size_t)K * N);
    uint32_t x = 12345u;
    for (auto& a : h_act) { x = x * 1664525u + 1013904223u; a = (int64_t)((x >> 8) % 7u); }
    int64_t* act = dalloc<int64_t>(h_act.size());
    HIP_OK(hipMemcpy(act, h_act.data(), h_act.size() * 8, hipMemcpyHostToDevice));
    // every step's rows are kept (a ring of K + 1 observation tensors per form)
    float* obs[2]; float* rew[2]; uint8_t *term[2], *trunc[2], *state[2][2];
    for (int f = 0; f < 2; ++f) {
        obs[f] = dalloc<float>((size_t)(K + 1) * N * D);
        rew[f] = dalloc<float>((size_t)K * N);
        term[f] = dalloc<uint8_t>((size_t)K * N);
        trunc[f] = dalloc<uint8_t>((size_t)K * N);
    }
    state[0][0] = dalloc<uint8_t>(marl_env_state_bytes(whole)); state[0][1] = nullptr;
    for (int k = 0; k < 2; ++k) state[1][k] = dalloc<uint8_t>(marl_env_state_bytes(sh[k]));
    hipStream_t main_s, s[2];
    HIP_OK(hipStreamCreate(&main_s));
    for (int k = 0; k < 2; ++k) HIP_OK(hipStreamCreate(&s[k]));
    hipEvent_t fork_ev, join_ev[2], t0, t1;
    HIP_OK(hipEventCreate(&fork_ev)); HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
    for (int k = 0; k < 2; ++k) HIP_OK(hipEventCreate(&join_ev[k]));

    auto whole_steps = [&] {
        for (int t = 0; t < K; ++t)
            EXPECT(marl_env_step(whole, state[0][0], act + (size_t)t * N, 8, obs[0] + (size_t)(t + 1) * N * D, rew[0] + (size_t)t * N,
                                 term[0] + (size_t)t * N, trunc[0] + (size_t)t * N, nullptr, main_s) == MARL_OK);
    };
    auto shard_steps = [&] {
        HIP_OK(hipEventRecord(fork_ev, main_s));
        for (int k = 0; k < 2; ++k) HIP_OK(hipStreamWaitEvent(s[k], fork_ev, 0));
        for (int t = 0; t < K; ++t)
            for (int k = 0; k < 2; ++k) {
                const size_t r0 = (size_t)t * N + (size_t)k * Nh;  // shard k's rows of step t
                EXPECT(marl_env_step(sh[k], state[1][k], act + r0, 8, obs[1] + ((size_t)(t + 1) * N + (size_t)k * Nh) * D, rew[1] + r0,
                                     term[1] + r0, trunc[1] + r0, nullptr, s[k]) == MARL_OK);
            }
        for (int k = 0; k < 2; ++k) {
            HIP_OK(hipEventRecord(join_ev[k], s[k]));
            HIP_OK(hipStreamWaitEvent(main_s, join_ev[k], 0));
        }
    };
    EXPECT(marl_env_reset(whole, state[0][0], obs[0], nullptr, nullptr, main_s) == MARL_OK);
    for (int k = 0; k < 2; ++k)
        EXPECT(marl_env_reset(sh[k], state[1][k], obs[1] + (size_t)k * Nh * D, nullptr, nullptr, main_s) == MARL_OK);
    whole_steps();
    shard_steps();
    HIP_OK(hipStreamSynchronize(main_s));
    // same rows from both forms: every reward and flag, every observation of every step
    auto same = [&](const void* a, const void* b, size_t bytes) {
        std::vector<uint8_t> ha(bytes), hb(bytes);
        HIP_OK(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
        return memcmp(ha.data(), hb.data(), bytes) == 0;
    };
    EXPECT(same(rew[0], rew[1], (size_t)K * N * 4));
    EXPECT(same(trunc[0], trunc[1], (size_t)K * N));
    for (int t = 0; t <= K; ++t) EXPECT(same(obs[0] + (size_t)t * N * D, obs[1] + (size_t)t * N * D, (size_t)N * D * 4));

    // timing: both forms replayed from graphs of K steps
    hipGraph_t g[2]; hipGraphExec_t ge[2];
    for (int f = 0; f < 2; ++f) {
        HIP_OK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
        if (f == 0) whole_steps(); else shard_steps();
        HIP_OK(hipStreamEndCapture(main_s, &g[f]));
        HIP_OK(hipGraphInstantiate(&ge[f], g[f], nullptr, nullptr, 0));
    }
    float us[2];
    const int reps = 24;
    for (int f = 0; f < 2; ++f) {
        for (int r = 0; r < 4; ++r) HIP_OK(hipGraphLaunch(ge[f], main_s));
        HIP_OK(hipEventRecord(t0, main_s));
        for (int r = 0; r < reps; ++r) HIP_OK(hipGraphLaunch(ge[f], main_s));
        HIP_OK(hipEventRecord(t1, main_s));
        HIP_OK(hipStreamSynchronize(main_s));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, t0, t1));
        us[f] = ms * 1e3f / (reps * K);
    }
    printf("65,536 envs, rows kept, from C++: one handle %.2f us per step, two shards on two streams %.2f us per step (same rows)\n",
           us[0], us[1]);
    for (int f = 0; f < 2; ++f) { HIP_OK(hipGraphExecDestroy(ge[f])); HIP_OK(hipGraphDestroy(g[f])); }
    marl_env_destroy(whole);
    for (int k = 0; k < 2; ++k) marl_env_destroy(sh[k]);
    for (int f = 0; f < 2; ++f)
        for (void* p : {(void*)obs[f], (void*)rew[f], (void*)term[f], (void*)trunc[f]}) HIP_OK(hipFree(p));
    for (void* p : {(void*)state[0][0], (void*)state[1][0], (void*)state[1][1], (void*)act}) HIP_OK(hipFree(p));
    HIP_OK(hipEventDestroy(fork_ev)); HIP_OK(hipEventDestroy(t0)); HIP_OK(hipEventDestroy(t1));
    for (int k = 0; k < 2; ++k) { HIP_OK(hipEventDestroy(join_ev[k])); HIP_OK(hipStreamDestroy(s[k])); }
    HIP_OK(hipStreamDestroy(main_s));
}

static void run_small_passes() {
    const int T = 37, N = 1001;
    float* r = dalloc<float>((size_t)T * N);
    float* d = dalloc<float>((size_t)T * N);
    float* v = dalloc<float>((size_t)T * N);
    float* nv = dalloc<float>(N);
    float* adv = dalloc<float>((size_t)T * N);
    float* ret = dalloc<float>((size_t)T * N);
    uint8_t* ws = dalloc<uint8_t>(marl_gae_workspace_bytes(T, N));
    double* sums = dalloc<double>(T);
    EXPECT(marl_gae(r, d, v, nv, T, N, 0.99, 0.95, adv, ret, 0, nullptr, 0, nullptr) == MARL_OK);
    EXPECT(marl_gae(r, d, v, nv, T, N, 0.99, 0.95, adv, nullptr, 1, ws, 0, nullptr) == MARL_OK);
    EXPECT(marl_gae(r, d, v, nv, T, N, 0.99, 0.95, adv, ret, 1, ws, 1, nullptr) == MARL_OK);
    EXPECT(marl_gae(r, d, v, nv, T, N, 0.99, 0.95, adv, ret, 1, nullptr, 0, nullptr) == MARL_ERR_INVALID);
    EXPECT(marl_gae(r, d, v, nv, 0, N, 0.99, 0.95, adv, ret, 0, nullptr, 0, nullptr) != MARL_ERR_HIP);
    EXPECT(marl_gae(nullptr, d, v, nv, T, N, 0.99, 0.95, adv, ret, 0, nullptr, 0, nullptr) == MARL_ERR_INVALID);
    EXPECT(marl_row_sums(r, T, N, sums, ws, nullptr) == MARL_OK);

    const int E = 333, A = 3;
    uint64_t* totals = dalloc<uint64_t>(4);
    double* first = dalloc<double>(4);
    uint8_t* ws1 = dalloc<uint8_t>(marl_reward_events_workspace_bytes(E, A));
    uint8_t* wsm = dalloc<uint8_t>(marl_reward_events_many_workspace_bytes(T, E, A));
    EXPECT(marl_reward_events(r, E, A, totals, first, ws1, nullptr) == MARL_OK);
    EXPECT(marl_reward_events(r, E, A, totals, nullptr, ws1, nullptr) == MARL_OK);
    EXPECT(marl_reward_events_many(r, T, E, A, totals, first, wsm, nullptr) == MARL_OK);
    EXPECT(marl_reward_events_noalloc(r, E, A, totals, first, nullptr) == MARL_OK);
    EXPECT(marl_reward_events_noalloc(r, 1 << 23, 4, totals, first, nullptr) == MARL_ERR_INVALID);
    EXPECT(marl_reward_events(r, E, A, nullptr, first, ws1, nullptr) == MARL_ERR_INVALID);

    const int rows = 777, K = 7;
    float* logits = dalloc<float>((size_t)rows * (K + 1));
    int64_t* actions = dalloc<int64_t>(rows);
    float* lp = dalloc<float>(rows);
    float* ent = dalloc<float>(rows);
    float* nxt = dalloc<float>(rows);
    uint64_t* counter = dalloc<uint64_t>(1);
    EXPECT(marl_sample_categorical(logits, rows, K, 0, 5, counter, 0, actions, lp, ent, nullptr, nullptr) == MARL_OK);
    EXPECT(marl_sample_categorical(logits, rows, K, K + 1, 5, nullptr, 10, actions, nullptr, nullptr, nxt, nullptr) == MARL_OK);
    EXPECT(marl_sample_categorical(logits, rows, K, 0, 5, nullptr, 10, actions, nullptr, nullptr, nxt, nullptr) == MARL_ERR_INVALID);
    EXPECT(marl_sample_categorical(logits, rows, K, 0, 5, nullptr, 10, nullptr, nullptr, nullptr, nullptr, nullptr) == MARL_ERR_INVALID);

    float* actf = dalloc<float>(rows);
    float* gl = dalloc<float>((size_t)rows * K);
    float* gv = dalloc<float>(rows);
    double* stats = dalloc<double>(7);
    int64_t* index = dalloc<int64_t>(rows);
    uint8_t* wsp = dalloc<uint8_t>(marl_ppo_loss_workspace_bytes(rows));
    EXPECT(marl_ppo_loss(logits, rows, K, 0, lp, nullptr, actf, lp, ent, nxt, 0.2, 0.01, 0.5, gl, gv, stats, wsp, nullptr) == MARL_OK);
    EXPECT(marl_ppo_loss(logits, rows, K, K + 1, lp, index, actf, lp, ent, nxt, 0.2, 0.01, 0.5, gl, gv, stats, wsp, nullptr) == MARL_OK);
    EXPECT(marl_ppo_loss(logits, rows, K, 0, lp, nullptr, actf, lp, ent, nxt, 0.2, 0.01, 0.5, gl, gv, stats, nullptr, nullptr) == MARL_ERR_INVALID);
    HIP_OK(hipDeviceSynchronize());
    for (void* p : {(void*)r, (void*)d, (void*)v, (void*)nv, (void*)adv, (void*)ret, (void*)ws, (void*)sums, (void*)totals,
                    (void*)first, (void*)ws1, (void*)wsm, (void*)logits, (void*)actions, (void*)lp, (void*)ent, (void*)nxt,
                    (void*)counter, (void*)actf, (void*)gl, (void*)gv, (void*)stats, (void*)index, (void*)wsp})
        HIP_OK(hipFree(p));
}

int main() {
    int count = 0;
    for (int A = 1; A <= 4; ++A)
        for (int feature = 0; feature <= 4; ++feature) {
            run_env(CRAMPED, 6, A, feature, A == 2 ? 65 : 33);
            ++count;
        }
    printf("cramped_room: %d (agents, feature) combinations\n", count);
    run_env(FORCED, 7, 2, MARL_FEAT_GLOBAL_OBS, 1);
    run_env(FORCED, 7, 2, MARL_FEAT_LOCAL_OBS, 4096);
    run_env(LARGE, 11, 4, MARL_FEAT_GLOBAL_OBS, 129);
    run_env(LARGE, 11, 3, MARL_FEAT_MINIMAL_SPATIAL_OTHER, 50);
    printf("forced_coordination and the large layout\n");

    // creation failures
    marl_env_t* env = nullptr;
    static const char* RAGGED[] = {"#####", "#C U#", "#  #", "#####"};
    static const char* UNKNOWN[] = {"#####", "#CxU#", "#   #", "#####"};
    static const char* NOFLOOR[] = {"###", "#C#", "###"};
    EXPECT(marl_env_create(&env, RAGGED, 4, 2, 0, 10, 4, 1, 0, 0) == MARL_ERR_LAYOUT);
    EXPECT(marl_env_create(&env, UNKNOWN, 4, 2, 0, 10, 4, 1, 0, 0) == MARL_ERR_LAYOUT);
    EXPECT(marl_env_create(&env, NOFLOOR, 3, 2, 0, 10, 4, 1, 0, 0) != MARL_OK);
    EXPECT(marl_env_create(&env, FORCED, 7, 3, 0, 10, 4, 1, 0, 0) == MARL_ERR_LAYOUT);  // two '+' cells, three agents
    EXPECT(marl_env_create(&env, CRAMPED, 6, 5, 0, 10, 4, 1, 0, 0) == MARL_ERR_INVALID);
    EXPECT(marl_env_create(&env, CRAMPED, 6, 0, 0, 10, 4, 1, 0, 0) == MARL_ERR_INVALID);
    EXPECT(marl_env_create(&env, CRAMPED, 6, 2, 9, 10, 4, 1, 0, 0) == MARL_ERR_INVALID);
    EXPECT(marl_env_create(&env, CRAMPED, 6, 2, 0, 10, 0, 1, 0, 0) == MARL_ERR_INVALID);
    EXPECT(marl_env_create(&env, nullptr, 6, 2, 0, 10, 4, 1, 0, 0) != MARL_OK);
    EXPECT(marl_env_create(nullptr, CRAMPED, 6, 2, 0, 10, 4, 1, 0, 0) != MARL_OK);
    marl_env_destroy(nullptr);
    printf("creation failures reported\n");

    run_small_passes();
    printf("GAE, row sums, reward events, sampling, PPO loss\n");
    run_pipelined_shards();
    printf("capi host pass: ok\n");
    return 0;
}

// The C ABI used from plain C++, no Python and no torch in the process: every entry point of
// include/marl_amd.h is called for all agent counts, feature sets and a few layouts (odd env
// counts included), then the argument checks are driven through their failure paths.  Prints one
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
    printf("capi host pass: ok\n");
    return 0;
}

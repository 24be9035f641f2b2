// Pipelined shards from plain C++ (no torch in the process), as a function of the things in which the
// C++ driver of the test suite and the Python bench leg differ: steps per graph, ring size, number
// of shards, whether rewards / flags of every step share one buffer.
//   shards_cpp E K RING SHARDS SHARED_SMALL [REPS] [ACTION_BATCHES]
// Prints us per step of the whole batch for one handle and for the shards.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/marl_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define EXPECT(x) do { if (!(x)) { fprintf(stderr, "%s:%d failed: %s (%s)\n", __FILE__, __LINE__, #x, marl_last_error()); exit(1); } } while (0)

static const char* const CRAMPED[] = {"#######", "#CCUCC#", "#O   O#", "#C   C#", "#C=C@C#", "#######"};

template <class T> static T* dalloc(size_t n) { void* p = nullptr; HIP_OK(hipMalloc(&p, n * sizeof(T))); HIP_OK(hipMemset(p, 0, n * sizeof(T))); return (T*)p; }

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 65536;
    const int K = argc > 2 ? atoi(argv[2]) : 16;
    const int RING = argc > 3 ? atoi(argv[3]) : K + 1;
    const int S = argc > 4 ? atoi(argv[4]) : 2;
    const int shared_small = argc > 5 ? atoi(argv[5]) : 0;
    const int reps = argc > 6 ? atoi(argv[6]) : 24;
    const int ACT_R = argc > 7 ? atoi(argv[7]) : K;  // distinct action batches, cycled
    const int A = 2, per = E / S;
    marl_env_t* whole = nullptr;
    std::vector<marl_env_t*> sh(S, nullptr);
    EXPECT(marl_env_create(&whole, CRAMPED, 6, A, MARL_FEAT_GLOBAL_OBS, 1000, E, 9, 0, 0) == MARL_OK);
    for (int k = 0; k < S; ++k)
        EXPECT(marl_env_create(&sh[k], CRAMPED, 6, A, MARL_FEAT_GLOBAL_OBS, 1000, per, 9, (int64_t)k * per, 0) == MARL_OK);
    const int D = marl_env_obs_dim(whole);
    const size_t N = (size_t)E * A, Np = (size_t)per * A;
    std::vector<int64_t> h_act((size_t)ACT_R * N);
    uint32_t x = 12345u;
    for (auto& a : h_act) { x = x * 1664525u + 1013904223u; a = (int64_t)((x >> 8) % 7u); }
    int64_t* act = dalloc<int64_t>(h_act.size());
    HIP_OK(hipMemcpy(act, h_act.data(), h_act.size() * 8, hipMemcpyHostToDevice));
    float* obs = dalloc<float>((size_t)RING * N * D);
    const size_t small_steps = shared_small ? 1 : K;
    float* rew = dalloc<float>(small_steps * N);
    uint8_t* term = dalloc<uint8_t>(small_steps * N);
    uint8_t* trunc = dalloc<uint8_t>(small_steps * N);
    uint8_t* st_whole = dalloc<uint8_t>(marl_env_state_bytes(whole));
    std::vector<uint8_t*> st(S);
    for (int k = 0; k < S; ++k) st[k] = dalloc<uint8_t>(marl_env_state_bytes(sh[k]));
    hipStream_t main_s;
    std::vector<hipStream_t> s(S);
    HIP_OK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    for (int k = 0; k < S; ++k) HIP_OK(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking));
    hipEvent_t fork_ev, t0, t1;
    std::vector<hipEvent_t> join_ev(S);
    HIP_OK(hipEventCreate(&fork_ev)); HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
    for (int k = 0; k < S; ++k) HIP_OK(hipEventCreate(&join_ev[k]));
    int slot = 0;  // ring position, advanced per captured step (the graph then writes fixed slots)
    auto whole_steps = [&] {
        for (int t = 0; t < K; ++t) {
            const size_t so = shared_small ? 0 : (size_t)t * N;
            const int o = (t + 1) % RING;
            EXPECT(marl_env_step(whole, st_whole, act + (size_t)(t % ACT_R) * N, 8, obs + (size_t)o * N * D, rew + so, term + so, trunc + so, nullptr, main_s) == MARL_OK);
        }
    };
    auto shard_steps = [&] {
        HIP_OK(hipEventRecord(fork_ev, main_s));
        for (int k = 0; k < S; ++k) HIP_OK(hipStreamWaitEvent(s[k], fork_ev, 0));
        for (int t = 0; t < K; ++t)
            for (int k = 0; k < S; ++k) {
                const size_t so = (shared_small ? 0 : (size_t)t * N) + (size_t)k * Np;
                const int o = (t + 1) % RING;
                EXPECT(marl_env_step(sh[k], st[k], act + (size_t)(t % ACT_R) * N + (size_t)k * Np, 8, obs + ((size_t)o * N + (size_t)k * Np) * D, rew + so,
                                     term + so, trunc + so, nullptr, s[k]) == MARL_OK);
            }
        for (int k = 0; k < S; ++k) {
            HIP_OK(hipEventRecord(join_ev[k], s[k]));
            HIP_OK(hipStreamWaitEvent(main_s, join_ev[k], 0));
        }
    };
    (void)slot;
    EXPECT(marl_env_reset(whole, st_whole, obs, nullptr, nullptr, main_s) == MARL_OK);
    for (int k = 0; k < S; ++k) EXPECT(marl_env_reset(sh[k], st[k], obs + (size_t)k * Np * D, nullptr, nullptr, main_s) == MARL_OK);
    HIP_OK(hipStreamSynchronize(main_s));
    hipGraph_t g[2]; hipGraphExec_t ge[2];
    for (int f = 0; f < 2; ++f) {
        HIP_OK(hipStreamBeginCapture(main_s, hipStreamCaptureModeThreadLocal));
        if (f == 0) whole_steps(); else shard_steps();
        HIP_OK(hipStreamEndCapture(main_s, &g[f]));
        HIP_OK(hipGraphInstantiate(&ge[f], g[f], nullptr, nullptr, 0));
    }
    float us[2];
    for (int round = 0; round < 2; ++round)
        for (int f = 0; f < 2; ++f) {
            for (int r = 0; r < 4; ++r) HIP_OK(hipGraphLaunch(ge[f], main_s));
            HIP_OK(hipEventRecord(t0, main_s));
            for (int r = 0; r < reps; ++r) HIP_OK(hipGraphLaunch(ge[f], main_s));
            HIP_OK(hipEventRecord(t1, main_s));
            HIP_OK(hipStreamSynchronize(main_s));
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, t0, t1));
            us[f] = ms * 1e3f / (reps * K);
            if (round == 1 && f == 1)
                printf("E=%d K=%d ring=%d shards=%d shared_small=%d act_r=%d: one handle %.2f us per step, shards %.2f us per step\n", E, K, RING, S, shared_small, ACT_R, us[0], us[1]);
        }
    return 0;
}

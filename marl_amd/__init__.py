"""MI355X-native batched Overcooked simulator + MAPPO rollout path.

Public surface (mirrors the reference's modules for this path):
  marl_amd.env            make_env, make_vector_env, BatchedOvercookedVecEnv
  marl_amd.gae            compute_gae (HIP kernel)
  marl_amd.episode_stats  EpisodeStats (reward-event bookkeeping, HIP kernel)
  marl_amd.sampling       CategoricalSampler (rollout action sampling, HIP kernel)
  marl_amd.ppo_loss       ppo_loss (minibatch loss + head gradients, HIP kernel)
  marl_amd.rollout_buffer Buffer
  marl_amd.policy         Agent (actor-critic MLP, PyTorch-ROCm)
  marl_amd.agents         MAPPO, CMAPPO
  marl_amd.rollout        agent_environment_loop
  marl_amd.train          command-line entry point
  marl_amd.replay         replays a recorded episode (spawns, actions, serving order) and compares
  marl_amd.evaluate       plays a saved policy (the reference's tests/test_load.py)
The simulation, featurisation, GAE, event bookkeeping, sampling and the PPO loss run in marl_amd/libmarl_amd.so (hand-written HIP for
gfx950, C ABI in include/marl_amd.h); importing the package does not load it, using it does.
"""
__version__ = "0.1.0"

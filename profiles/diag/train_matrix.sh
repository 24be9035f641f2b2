run() { echo "== $*"; timeout -k 10 120 python -m marl_amd.train --num-steps 16 --max-steps 30 --data-path /tmp/dm --ppo-epoch 1 "$@" 2>&1 | tail -1 | cut -c1-120; }
run --num-agents 1 --num-envs 4 --layout overcooked_cramped_room_v0 --total-steps $((4*1*16*5))
run --num-agents 1 --num-envs 1 --layout overcooked_cramped_room_v0 --total-steps $((1*1*16*5)) --num-minibatches 2
run --num-agents 2 --num-envs 1 --layout overcooked_cramped_room_v0 --total-steps $((1*2*16*5))
run --num-agents 2 --num-envs 8 --layout overcooked_cramped_room_v0 --feature only_direction --total-steps $((8*2*16*5))
run --num-agents 2 --num-envs 8 --layout overcooked_cramped_room_v0 --feature Minimal_spatial --total-steps $((8*2*16*5))
run --num-agents 2 --num-envs 6 --layout overcooked_cramped_room_v0 --feature Minimal_spatial_other_agent_aware --total-steps $((6*2*16*5))
run --num-agents 4 --num-envs 8 --layout large_overcooked_layout --total-steps $((8*4*16*5))
run --num-agents 4 --num-envs 8 --layout overcooked_coordination_ring_v0 --feature local_obs --algorithm cmappo --total-steps $((8*4*16*5))
run --num-agents 3 --num-envs 7 --layout overcooked_counter_circuit_v0 --feature local_obs --algorithm cmappo --total-steps $((7*3*16*5))
run --num-agents 2 --num-envs 8 --layout overcooked_asymmetric_advantages_v0 --total-steps $((8*2*16*5)) --sampler torch --fused-loss off --adam foreach --graph-rollout off --graph-update off
run --num-agents 2 --num-envs 8 --layout overcooked_forced_coordination_v0 --total-steps $((8*2*16*5)) --graph-rollout on --graph-update on --log

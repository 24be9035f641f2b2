# training throughput, rollout replayed from hipGraphs vs launched eagerly
for envs in 16 256 4096; do for g in off on; do
  steps=$(( envs * 2 * 128 * 40 ))
  timeout -k 10 300 python -m marl_amd.train --num-agents 2 --num-envs $envs --layout overcooked_cramped_room_v0 --num-steps 128 --total-steps $steps --ppo-epoch 4 --lr 3e-4 --feature global_obs --graph-rollout $g --data-path /tmp/ts 2>&1 | tail -1 | cut -c1-70 | sed "s/^/envs=$envs graph=$g: /" || exit 1
done; done

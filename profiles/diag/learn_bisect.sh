for combo in "fused on" "fused off"; do
  set -- $combo
  timeout -k 10 300 python -m marl_amd.train --num-agents 2 --num-envs 256 --layout overcooked_cramped_room_v0 --num-steps 128 --num-minibatches 4 --total-steps 20000000 --ppo-epoch 4 --clip-param 0.2 --value-loss-coef 0.5 --entropy-coef 0.01 --lr 3e-4 --feature global_obs --seed 1 --sampler $1 --graph-update $2 --data-path /tmp/lb 2>&1 | tail -1 | sed "s/^/sampler=$1 graph-update=$2: /" | cut -c1-60,150-400
done

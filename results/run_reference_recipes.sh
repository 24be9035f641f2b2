#!/bin/bash
# The reference's two published training recipes for cramped_room, three seeds each, on one MI355X
# (run from the repo root on the GPU box; CSVs land under gpurun_out/results/):
#   makefile: Makefile:5-9   (E=16, T=256, 4 minibatches, 5 epochs, clip .05, vf .1, ent .01, lr 3e-4, 20 M steps)
#   readme:   README.md:58-61 (same sizes, --centralised, clip .2, vf .5)
#   forced:   Makefile:10-14 (the makefile recipe on overcooked_forced_coordination_v0)
# --shuffle numpy: the reference's np.random.shuffle of one index array (agents/MAPPO.py:156-160).
set -e
OUT=gpurun_out/results
mkdir -p $OUT
BASE="--num-agents 2 --num-envs 16 --num-steps 256 --num-minibatches 4 --total-steps ${STEPS:-20000000} --ppo-epoch 5 --entropy-coef 0.01 --gamma 0.99 --lam 0.95 --max-grad-norm 0.5 --lr 3e-4 --feature global_obs --shuffle numpy"
COMMON="$BASE --layout overcooked_cramped_room_v0"
for seed in ${SEEDS:-1 2 3}; do
  timeout -k 10 900 python -u -m marl_amd.train $COMMON --clip-param 0.05 --value-loss-coef 0.1 --seed $seed --data-path $OUT/r02_makefile_cramped > $OUT/makefile_seed_$seed.log 2>&1
  tail -1 $OUT/makefile_seed_$seed.log | cut -c1-200
  timeout -k 10 900 python -u -m marl_amd.train $COMMON --centralised --clip-param 0.2 --value-loss-coef 0.5 --seed $seed --data-path $OUT/r02_readme_cramped > $OUT/readme_seed_$seed.log 2>&1
  tail -1 $OUT/readme_seed_$seed.log | cut -c1-200
  timeout -k 10 900 python -u -m marl_amd.train $BASE --layout overcooked_forced_coordination_v0 --clip-param 0.05 --value-loss-coef 0.1 --seed $seed --data-path $OUT/r02_makefile_forced > $OUT/forced_seed_$seed.log 2>&1
  tail -1 $OUT/forced_seed_$seed.log | cut -c1-200
done

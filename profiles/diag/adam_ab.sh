# same-box comparison of the Adam variants (update time per 65,536-sample update at 256 envs)
for v in foreach fused foreach fused; do
  PYTHONPATH=$PWD timeout -k 10 300 python profiles/diag/phase_split.py --num-agents 2 --num-envs 256 --layout overcooked_cramped_room_v0 --num-steps 128 --total-steps 13107200 --ppo-epoch 4 --lr 3e-4 --feature global_obs --data-path /tmp/ts --adam $v 2>&1 | tail -1 | sed "s/^/$v: /" || exit 1
done

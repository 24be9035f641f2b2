# kernel-time breakdown of a short MAPPO training run (rollout + GAE + PPO update)
export TMPDIR=/tmp
OUT=gpurun_out/prof_train
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 -m marl_amd.train --num-agents 2 --num-envs ${ENVS:-4096} --layout overcooked_cramped_room_v0 --num-steps 128 --total-steps $(( ${ENVS:-4096} * 2 * 128 * 3 )) --ppo-epoch 4 --lr 3e-4 --feature global_obs ${EXTRA} > $OUT/train.log 2>&1 || { tail -5 $OUT/train.log; exit 1; }
tail -3 $OUT/train.log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_train/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:28]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY

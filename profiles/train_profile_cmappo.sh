# BASELINE config 3: centralised MAPPO on forced_coordination, 64 envs, kernel-time breakdown of 40 updates
export TMPDIR=/tmp
OUT=gpurun_out/prof_train_cmappo
mkdir -p $OUT
ARGS="--num-agents 2 --num-envs 64 --layout overcooked_forced_coordination_v0 --num-steps 128 --num-minibatches 4 --ppo-epoch 4 --clip-param 0.2 --value-loss-coef 0.5 --entropy-coef 0.01 --lr 3e-4 --feature global_obs --seed 1 --algorithm cmappo --centralised --data-path /tmp/dc"
timeout -k 10 200 python3 -m marl_amd.train $ARGS --total-steps $(( 64 * 2 * 128 * 300 )) 2>&1 | tail -1 | cut -c1-100
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 -m marl_amd.train $ARGS --total-steps $(( 64 * 2 * 128 * 40 )) > $OUT/train.log 2>&1 || { tail -5 $OUT/train.log; exit 1; }
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_train_cmappo/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:26]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
rm -rf $OUT

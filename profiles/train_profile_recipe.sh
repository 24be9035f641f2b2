# kernel-time breakdown of the reference's Makefile recipe (E=16, T=256, 5 epochs, 4 minibatches), 60 updates
export TMPDIR=/tmp
OUT=gpurun_out/prof_train_recipe
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 -m marl_amd.train --num-agents 2 --num-envs 16 --layout overcooked_cramped_room_v0 --num-steps 256 --num-minibatches 4 --total-steps $(( 16 * 2 * 256 * 60 )) --ppo-epoch 5 --clip-param 0.05 --value-loss-coef 0.1 --lr 3e-4 --feature global_obs --shuffle numpy ${EXTRA} > $OUT/train.log 2>&1 || { tail -5 $OUT/train.log; exit 1; }
tail -2 $OUT/train.log | cut -c1-160
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_train_recipe/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:32]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY

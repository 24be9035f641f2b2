# waves per workgroup (MARL_AMD_WAVES) on the secondary configs, two alternating observation tensors
# like profiles/bench_secondary.py: us per launch
run() {  # layout agents feature envs
  for w in 1 2 4; do
    MARL_AMD_WAVES=$w timeout -k 10 120 python bench.py --layout $1 --agents $2 --feature $3 --envs $4 --obs-ring 2 --steps 512 --warmup 64 --no-cpu-baseline --no-extra-legs \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 A=$2 $3 E=$4 waves=$w us_per_launch=%.2f GB/s=%.0f' % (r['us_per_launch'], r['achieved']))" || exit 1
  done
}
run overcooked_coordination_ring_v0 4 global_obs 32768 &&
run large_overcooked_layout 4 global_obs 32768 &&
run overcooked_coordination_ring_v0 4 local_obs 65536 &&
run overcooked_counter_circuit_v0 2 global_obs 131072

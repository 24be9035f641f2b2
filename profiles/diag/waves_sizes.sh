# launch time of the env kernel beyond the last-level cache, by waves per workgroup
set -e
for envs in 131072 262144; do
  for w in 1 2 4; do
    MARL_AMD_WAVES=$w timeout -k 10 120 python bench.py --envs $envs --steps 512 --warmup 64 --no-cpu-baseline --no-rows-kept \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('envs',$envs,'waves',$w,'us',round(d['roofline']['us_per_launch'],2),'GB/s',round(d['roofline']['achieved']))"
  done
done

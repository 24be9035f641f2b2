# non-temporal hint on/off over env counts: where does it start to pay?
for e in 65536 98304 131072 163840 196608 262144 1048576; do for nt in 0 1; do
  MARL_AMD_STREAM_NT=$nt timeout -k 10 120 python bench.py --no-cpu-baseline --no-rows-kept --steps 512 --envs $e | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('envs=$e nt=$nt us_per_launch=%.2f GB/s=%.0f' % (r['us_per_launch'], r['achieved']))" || exit 1
done; done

for rep in 1 2; do for w in 1 2 4; do
  MARL_AMD_WAVES=$w timeout -k 10 120 python bench.py --no-cpu-baseline --no-rows-kept --steps 4096 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves/block=$w us_per_launch=%.2f frac=%.3f' % (d['roofline']['us_per_launch'], d['roofline']['frac']))" || exit 1
done; done

# A/B diagnostic variants of the library in separate bench runs on the same box: "name[:waves]" ...
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; w=${spec#*:}; [ "$w" = "$spec" ] && w=4
  MARL_AMD_WAVES=$w MARL_AMD_LIB=$PWD/profiles/diag/libvar_$name.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-rows-kept --steps 4096 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name waves=$w us_per_launch=%.2f frac=%.3f' % (d['roofline']['us_per_launch'], d['roofline']['frac']))" || exit 1
done; done

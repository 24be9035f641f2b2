# A/B the cache-policy variants of the obs store stream in separate bench runs (same box)
for rep in 1 2; do
for aux in 0 2 16 17 18; do
  MARL_AMD_LIB=$PWD/profiles/diag/libvar_$aux.so timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4096 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aux=$aux us_per_launch=%.2f frac=%.3f' % (d['roofline']['us_per_launch'], d['roofline']['frac']))" || exit 1
done; done

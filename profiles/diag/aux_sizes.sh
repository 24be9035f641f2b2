# store cache-policy variants (libvar_aux<N>.so) at several env counts
for e in 65536 262144; do for aux in 0 1 2 3 16 17 18 19; do
  MARL_AMD_LIB=$PWD/profiles/diag/libvar_aux$aux.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-rows-kept --steps 1024 --envs $e | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('envs=$e aux=$aux us_per_launch=%.2f GB/s=%.0f' % (r['us_per_launch'], r['achieved']))" || exit 1
done; done

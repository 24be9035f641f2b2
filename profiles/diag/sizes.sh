# full kernel vs stream-only (MARL_AMD_ABLATE=2) time per launch over env counts
# needs the diagnostic library: sh profiles/diag/build_variants.sh ablate:-DOC_ABLATE
export MARL_AMD_LIB=$PWD/profiles/diag/libvar_ablate.so
for e in 65536 131072 262144 524288; do for a in 0 2; do
  MARL_AMD_ABLATE=$a timeout -k 10 120 python bench.py --no-cpu-baseline --no-rows-kept --steps 1024 --envs $e | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('envs=$e ablate=$a us_per_launch=%.2f us_per_65536=%.2f GB/s=%.0f' % (r['us_per_launch'], r['us_per_launch']*65536/$e, r['achieved']))" || exit 1
done; done

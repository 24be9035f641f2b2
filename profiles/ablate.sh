# Phase ablations of the fused kernel (profiling aid).  Needs the diagnostic library that reads
# MARL_AMD_ABLATE: sh profiles/diag/build_variants.sh ablate:-DOC_ABLATE  (the product library ignores it).
export MARL_AMD_LIB=$PWD/profiles/diag/libvar_ablate.so
export TMPDIR=/tmp
for a in ${ABLATE_SET:-0 1 2}; do
  MARL_AMD_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$a -o t -- python3 bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-rows-kept > gpurun_out/abl_$a.json 2>/dev/null || exit 1
  grep env_kernel gpurun_out/abl_$a/t_kernel_stats.csv | awk -F'",' '{print "ablate='$a'", $2}' | cut -c1-120
done

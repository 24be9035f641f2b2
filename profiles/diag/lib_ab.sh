#!/bin/bash
# A/B of the product library against a diagnostic variant (profiles/diag/libvar_<name>.so) on the three
# bench configurations: bash profiles/diag/lib_ab.sh <name> [rounds]
name=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for lib in product $name; do
    for cfg in 2 4 5; do
      if [ $lib = product ]; then unset MARL_AMD_LIB; else export MARL_AMD_LIB=$PWD/profiles/diag/libvar_$lib.so; fi
      timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=lambda k,f: round(d[k][f],2) if k in d and f in d[k] else None
print('$lib cfg$cfg kept', round(d['roofline']['us_per_launch'],2), 'frac', round(d['roofline']['frac'],3), 'llc', g('llc_resident','us_per_launch'), 'many', g('step_many','us_per_step'), 'shards', g('two_shards','us_per_step'), 'policy_ms', round(d['policy_driven']['ms_per_step'],4))
" || exit 1
    done
  done
done

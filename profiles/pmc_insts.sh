# Instruction-mix counters of the fused kernel (one pass, 8 SQ counters)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_insts
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_insts -o insts -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-rows-kept > gpurun_out/pmc_insts/bench.json 2> gpurun_out/pmc_insts/err.txt
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/pmc_insts/insts_counter_collection.csv')))
v = collections.defaultdict(list)
for r in rows:
    if 'env_kernel' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
w = sum(v['SQ_WAVES'][-20:]) / 20
for k in sorted(v):
    m = sum(v[k][-20:]) / 20
    print(f"{k:20s} {m:14.0f}  per wave {m / w:10.1f}")
PY

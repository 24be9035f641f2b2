cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/gpurun_out/bin/fill_counters
$B
p=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_64B_sum" "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_WRITE_sum TCC_EA0_WRREQ_DRAM_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"; do
  p=$((p+1))
  timeout -s KILL 60 rocprofv3 --pmc $set --output-format csv -d /tmp/fc_$p -o c -- $B > /dev/null 2>/tmp/fc_$p.err || { echo "pass $p failed"; tail -3 /tmp/fc_$p.err; continue; }
  python3 - /tmp/fc_$p <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:14s} {c:40s} mean {sum(v)/len(v):16.1f} over {len(v)} launches")
PY
done

# Needs a library built with profiles/diag/xcd_split.patch (git apply, then make in marl_amd/csrc): the product does not read these variables.
b() { python bench.py --steps 2048 --warmup 64 --no-cpu-baseline --no-extra-legs --action-batches 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['us_per_launch'],2))"; }
for wv in 1 2; do
echo "waves $wv equal: kept $(MARL_AMD_WAVES=$wv b)  llc $(MARL_AMD_WAVES=$wv b --obs-ring 2)"
for d in 0.03 0.05 0.07 0.09; do
W=$(python -c "e=1+$d; o=1-$d; print(','.join(str(e if i%2==0 else o) for i in range(8)))")
echo "waves $wv even/odd +-$d: kept $(MARL_AMD_WAVES=$wv MARL_AMD_XCD_WEIGHTS=$W b)  llc $(MARL_AMD_WAVES=$wv MARL_AMD_XCD_WEIGHTS=$W b --obs-ring 2)"
done; done
echo "waves 4 equal: kept $(b)  llc $(b --obs-ring 2)"

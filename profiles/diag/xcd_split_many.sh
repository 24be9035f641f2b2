# Needs a library built with profiles/diag/xcd_split.patch (git apply, then make in marl_amd/csrc): the product does not read these variables.
# multi-step launch with an uneven split over the XCDs: us per step for several even/odd weights
for wv in 2 1; do
for d in 0 0.04 0.06 0.08 0.10 0.12; do
echo "waves $wv delta $d: $(MARL_AMD_XCD_SPLIT=$d MARL_AMD_XCD_SPLIT_WAVES=$wv SM_K=32,128 python profiles/diag/step_many_stagger.py 2>/dev/null | sed 's/.*E=65536: //')"
done; done
export MARL_AMD_LIB=$PWD/profiles/diag/libvar_stamps.so
MARL_AMD_XCD_SPLIT=0.08 python profiles/diag/stamps_many.py 2>/dev/null | grep "^xcc\|launch of\|per CU"

# small stores before (product library for two agents) or after (libvar_deferall.so) the row stream, two-agent configs
for lib in "" $PWD/profiles/diag/libvar_deferall.so; do
  for spec in "overcooked_cramped_room_v0 65536 2" "overcooked_cramped_room_v0 65536 32" "overcooked_counter_circuit_v0 131072 2" "overcooked_cramped_room_v0 262144 2" "overcooked_cramped_room_v0 1048576 2"; do
    set -- $spec
    MARL_AMD_LIB=$lib timeout -k 10 120 python bench.py --layout $1 --envs $2 --obs-ring $3 --steps 512 --warmup 64 --no-cpu-baseline --no-extra-legs \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib=${lib##*/} $1 E=$2 ring=$3 us_per_launch=%.2f GB/s=%.0f' % (r['us_per_launch'], r['achieved']))" || exit 1
  done
done

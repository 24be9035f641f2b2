# builds diagnostic variants of the library: name:extra-flags
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/var
build() {  # name flags...
  name=$1; shift
  d=build/var/$name; mkdir -p $d
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $*"
  for a in 1 2 3 4; do /opt/rocm/bin/hipcc $F -DOC_AGENTS=$a -c marl_amd/csrc/env_kernels.hip -o $d/env_a$a.o & done
  /opt/rocm/bin/hipcc $F -c marl_amd/csrc/gae_kernels.hip -o $d/gae.o &
  /opt/rocm/bin/hipcc $F -c marl_amd/csrc/stats_kernels.hip -o $d/stats.o &
  /opt/rocm/bin/hipcc $F -c marl_amd/csrc/sample_kernels.hip -o $d/sample.o &
  /opt/rocm/bin/hipcc $F -c marl_amd/csrc/ppo_kernels.hip -o $d/ppo.o &
  /opt/rocm/bin/hipcc $F -c marl_amd/csrc/capi.hip -o $d/capi.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/diag/libvar_$name.so $d/*.o
}
for spec in "$@"; do build ${spec%%:*} ${spec#*:}; done

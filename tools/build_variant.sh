#!/bin/bash
# Builds gpurun_out-independent VARIANTS of libmw_cdna4.so with extra hipcc flags for A/B timing (MW_LIB_PATH selects one at run time):
#   bash tools/build_variant.sh <name> <extra flags...>   ->  miniweatherml_amd/ab/libmw_<name>.so
# (miniweatherml_amd/ab/ is listed in .gpurunignore so that stale A/B builds never travel with a round-end push: take the line out for
#  the gpurun call that times them, and `rm -rf miniweatherml_amd/ab` afterwards)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=${MW_SRC_DIR:-$R/miniweatherml_amd/csrc}     # MW_SRC_DIR: another checkout's csrc (e.g. `git archive <rev>` for a same-box baseline)
mkdir -p $R/miniweatherml_amd/ab /tmp/mwvar_$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-unused-variable -ffp-contract=on -I/opt/rocm/include"
objs=""
for s in mw_host.cpp mw_dycore.hip mw_kessler.hip mw_mlp.hip mw_column.hip mw_output.hip mw_netcdf.cpp mw_rccl.cpp mw_h5.cpp; do
  o=/tmp/mwvar_$name/${s%.*}.o
  x=""; case $s in *.hip) x="-x hip";; esac
  extra=""; case $s in *.hip) extra="$*";; esac      # (the extra flags reach every HIP source: -DMW_KES_SWEEP=1, -DMW_ZERO_SKIP=0, ...)
  /opt/rocm/bin/hipcc $FLAGS $extra $x -c $C/$s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/miniweatherml_amd/ab/libmw_$name.so $objs -ldl
echo built $R/miniweatherml_amd/ab/libmw_$name.so

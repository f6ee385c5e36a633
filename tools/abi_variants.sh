#!/bin/bash
# Builds variant copies of the library that differ in the -D flags of veto_abi.hip only (build/libveto_abi_<name>.so).
# usage: tools/abi_variants.sh name1:"-DFLAG=.." ...
set -e
R=$PWD; C=$R/veto_amd/csrc; O=$R/build/obj; mkdir -p $O
SRC="gemm_split_ps rowops ffn_fused attention postprocess roialign sgg_eval losses backward train"
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -Wno-unused-value"
for s in $SRC; do
  if [ ! -f $O/$s.o ] || [ $C/$s.hip -nt $O/$s.o ] || [ $C/kernels.h -nt $O/$s.o ] || [ $C/common.h -nt $O/$s.o ]; then
    (cd $C && hipcc $FL -c $s.hip -o $O/$s.o) &
  fi
done
wait
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (cd $C && hipcc $FL $flags -c veto_abi.hip -o $O/abi_$name.o && hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/libveto_abi_$name.so $O/abi_$name.o $(for s in $SRC; do echo $O/$s.o; done) && echo built $name) &
done
wait

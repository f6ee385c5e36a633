#!/bin/bash
# Builds variant copies of the library that differ in the -D flags of ONE source: build/libveto_<stem>_<name>.so (A/B arms for
# VETO_AMD_LIB=...; build/ is neither tracked nor kept: prune it after the run that used it).
# usage: tools/variants.sh <source stem, e.g. qkv_attn_fused> name1:"-DFLAG=.." name2:"..." ...   (run from the repo root)
set -e
R=$PWD; C=$R/veto_amd/csrc; O=$R/build/obj
STEM=$1; shift
python3 -c "from veto_amd.build import build_native; build_native()"    # fills build/obj with the other sources' objects
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -Wno-unused-value"
OTHERS=$(python3 -c "from veto_amd.build import SOURCES; print(' '.join('$O/' + s + '.o' for s in SOURCES if s != '$STEM.hip'))")
for o in $OTHERS; do [ -f $o ] || (cd $C && hipcc $FL -c $(basename ${o%.o}) -o $o); done
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (cd $C && hipcc $FL $flags -c $STEM.hip -o $O/${STEM}_$name.o && hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/libveto_${STEM}_$name.so $O/${STEM}_$name.o $OTHERS && echo built build/libveto_${STEM}_$name.so) &
done
wait

#!/bin/bash
# GPU box: times the layer-tail launch of every build/libveto_ffn_<name>.so given (or all), two rounds (boxes warm up).
# usage: tools/tail_variants.sh [names...]
names="$@"; [ -z "$names" ] && names=$(ls build/libveto_ffn_*.so | sed 's/.*libveto_ffn_\(.*\)\.so/\1/')
for round in 1 2; do
  for n in $names; do
    echo "== $n"; VETO_AMD_LIB=build/libveto_ffn_$n.so python tools/layer_tail_bench.py 287280 2>&1 | grep -v amdgpu.ids | grep "one launch"
  done
done

#!/bin/bash
# GPU box: times the fused FeedForward kernel of every build/libveto_ffn_<name>.so given (or all). usage: tools/ffn_run_variants.sh [names...]
names="$@"; [ -z "$names" ] && names=$(ls build/libveto_ffn_*.so | sed 's/.*libveto_ffn_\(.*\)\.so/\1/')
for n in $names; do
  VETO_AMD_LIB=build/libveto_ffn_$n.so FFN_FAST=1 python tools/ffn_bench.py 287280 2>&1 | grep -v amdgpu.ids
done

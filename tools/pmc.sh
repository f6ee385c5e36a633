#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> "<counters>" [bench args...]
# One rocprofv3 --pmc pass of a short bench run; CSVs land in gpurun_out/pmc_<tag>/.
tag=$1; ctrs=$2; shift 2
R=$PWD
export TMPDIR=/tmp
cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag rc=$?"

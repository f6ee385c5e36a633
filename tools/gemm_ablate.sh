#!/bin/bash
# GPU box: per-variant device time of the split GEMM kernel on the QKV shape (rocprofv3 kernel stats).
R=$PWD; export TMPDIR=/tmp
for bm in ${BMS:-256 128}; do for abl in ${ABLS:-0 1 2 3 4 7}; do
  out=$R/gpurun_out/abl_${bm}_${abl}
  (cd /tmp && VETO_GEMM_BM=$bm VETO_GEMM_ABLATE=$abl rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/gemm_bench.py ${NSHAPES:-1} > $out.log 2>&1)
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  echo "BM=$bm ABL=$abl $(grep gemm_split $f | awk -F, '{print "calls",$2,"avg_us",$4/1000}')"
done; done

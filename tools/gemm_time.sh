#!/bin/bash
# GPU box: kernel-trace timing of the split GEMM on the four transformer shapes. usage: tools/gemm_time.sh <tag> [env...]
tag=$1; shift
R=$PWD; export TMPDIR=/tmp
out=$R/gpurun_out/gt_$tag; rm -rf $out
(cd /tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/gemm_bench.py 4 > $out.log 2>&1)
python3 - <<P
import csv,glob
fs=glob.glob('$out/*/*kernel_trace.csv')
rows=[r for r in csv.DictReader(open(fs[0])) if 'gemm_split' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('$tag', ' | '.join('min %.0f med %.0f'%(min(d[i:i+7]), sorted(d[i:i+7])[3]) for i in range(0,len(d),7)))
P

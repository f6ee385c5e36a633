#!/bin/bash
# GPU box: the per-round measurement set. usage: tools/profile_round.sh <tag> [bench args...]
# kernel stats (rocprofv3 --kernel-trace --stats), three PMC passes (never combined with other tracing), bench JSON.
tag=$1; shift; R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra "$@" > $O/stats.log 2>&1)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 tools/trace_launches.py $(ls $O/stats/*/*kernel_trace.csv | head -1) > $O/gemm_by_launch.txt
tools/pmc.sh ${tag}_fetch FETCH_SIZE --no-extra "$@"; python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_fetch > $O/pmc_fetch.txt
tools/pmc.sh ${tag}_write "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --no-extra "$@"; python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_write > $O/pmc_write.txt
tools/pmc.sh ${tag}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY TCC_HIT_sum TCC_MISS_sum" --no-extra "$@"; python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_mfma > $O/pmc_mfma.txt
# (another workload than the headline's: TRAFFIC_WORKLOAD="12 36 6 6 mixed" TRAFFIC_OUT=traffic_l6h6.json tools/profile_round.sh <tag> --layers 6 --heads 6)
python3 tools/pmc_traffic.py $tag gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/pmc_${tag}_mfma ${TRAFFIC_WORKLOAD:+--workload $TRAFFIC_WORKLOAD} --out ${TRAFFIC_OUT:-traffic.json} > $O/traffic.txt
cp profiles/${TRAFFIC_OUT:-traffic.json} $O/traffic.json
python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json

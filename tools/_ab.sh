python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 40 --no-extra 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],3), d.get('logit_max_abs_err'), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if k in ('gemm_qkv0_lc','assemble_tokens','attention')})
"; done

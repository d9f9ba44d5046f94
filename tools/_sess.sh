mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_equivariance.py -q -m gpu -x -k "lm_head or equivar or rc" 2>&1 | tail -2
timeout 180 python tools/step_families.py 2>&1 | tail -1 > gpurun_out/lm.log
python -c "
import json
for l in open('gpurun_out/lm.log'):
    d=json.loads(l); print(d['step_ms'], 'lm_head', d['lm_head_ms'], d['lm_head_n'], 'embed', d['embed_ms'])
"

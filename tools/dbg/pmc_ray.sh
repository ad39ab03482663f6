#!/bin/bash
# PMC passes over tools/run_hot.py for the k_ray kernels: is the attention LDS- or VALU-bound?  (one counter set per run)
R=$PWD; T=${1:-ray}; mkdir -p $R/gpurun_out/$T; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/$T/lds_counters.txt
for SET in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  N=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/$T/pmc/$N -o p -- python $R/tools/run_hot.py --iters 1 > $R/gpurun_out/$T/pmc_$N.log 2>&1
done
cd $R; python tools/pmc_summary.py gpurun_out/$T/pmc gpurun_out/$T/pmc_counters.json > gpurun_out/$T/pmc_summary.log 2>&1; rm -rf gpurun_out/$T/pmc
python - <<PY
import json
d=json.load(open('gpurun_out/$T/pmc_counters.json'))
for k,v in d['kernels'].items():
    if 'k_ray' in k: print(k, json.dumps({a: round(b) for a, b in v.items()}))
PY

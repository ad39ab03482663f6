#!/bin/bash
# a few PMC passes over tools/run_hot.py (one counter set per run), summarised by tools/pmc_summary.py
R=$PWD; T=${1:-q}; mkdir -p $R/gpurun_out/$T; cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  N=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/$T/pmc/$N -o p -- python $R/tools/run_hot.py --iters 1 > $R/gpurun_out/$T/pmc_$N.log 2>&1
done
cd $R; python tools/pmc_summary.py gpurun_out/$T/pmc gpurun_out/$T/pmc_counters.json > gpurun_out/$T/pmc_summary.log 2>&1; rm -rf gpurun_out/$T/pmc
python - <<PY
import json
d=json.load(open('gpurun_out/$T/pmc_counters.json'))
for k,v in d['kernels'].items():
    if 'k_chain' in k: print(k, json.dumps({a: round(b) for a, b in v.items()}))
PY

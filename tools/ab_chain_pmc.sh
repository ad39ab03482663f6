#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_MFMA of k_chain per build variant (one rocprofv3 --pmc pass each, nothing else enabled):
#   tools/ab_chain_pmc.sh OUT.json libgnr.so libgnr_x.so ...      (the per-tile figures: / 128 000 tiles of the B = 32 volume launch)
OUT=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  rm -rf /tmp/abpmc_$LIB
  GNR_LIB=$LIB rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/abpmc_$LIB -o p -- python $R/tools/run_hot.py --iters 1 > /tmp/abpmc_$LIB.log 2>&1
done
cd $R
python - "$OUT" "$@" <<'PY'
import sys, json, os
sys.path.insert(0, 'tools')
import pmc_summary
out = {}
for lib in sys.argv[2:]:
    res = pmc_summary.aggregate('/tmp/abpmc_' + lib)
    k = res.get('k_chain<6, false, false, false, true>', {})
    r = res.get('k_chain<6, true, false, false, true>', {})
    out[lib] = {'volume_launch': {'SQ_INSTS_VALU': k.get('SQ_INSTS_VALU'), 'SQ_INSTS_MFMA': k.get('SQ_INSTS_MFMA'),
                                  'vector_insts_per_tile_excl_mfma': None if not k else round((k['SQ_INSTS_VALU'] - k['SQ_INSTS_MFMA']) / 128000, 1),
                                  'mfma_per_tile': None if not k else round(k['SQ_INSTS_MFMA'] / 128000, 1)},
                'render_launch': {'SQ_INSTS_VALU': r.get('SQ_INSTS_VALU'), 'SQ_INSTS_MFMA': r.get('SQ_INSTS_MFMA')}}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(out, indent=1))
PY

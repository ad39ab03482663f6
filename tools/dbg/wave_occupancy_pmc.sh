#!/bin/bash
# Mean wavefront life as a fraction of the launch for every kernel of a command, from one rocprofv3 --pmc pass:
#   SQ_WAVE_CYCLES (quad-cycles summed over wavefronts) * 4 / (GRBM_GUI_ACTIVE / 8 XCDs * SQ_WAVES)      -> a launch whose wave slots stand
#   empty towards its end (static shares + wavefronts of unequal speed) shows well below 1.
#   tools/dbg/wave_occupancy_pmc.sh OUT.json -- python tools/time_volume_bwd.py --scenes 8
OUT=$1; shift; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wocc
rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/wocc -o p -- "$@" > /tmp/wocc.log 2>&1
cd $R
python - "$OUT" <<'PY'
import sys, json
sys.path.insert(0, 'tools')
import pmc_summary
res = pmc_summary.aggregate('/tmp/wocc')
out = {}
for k, v in res.items():
    if not all(c in v for c in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAVES')) or v['SQ_WAVES'] < 64 or v['GRBM_GUI_ACTIVE'] < 4e5:
        continue
    out[k] = {'mean_wave_life_frac': round(4.0 * v['SQ_WAVE_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8.0 * v['SQ_WAVES']), 3), 'waves': v['SQ_WAVES'],
              'launch_us_at_2.4GHz': round(v['GRBM_GUI_ACTIVE'] / 8.0 / 2400.0, 1)}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['launch_us_at_2.4GHz']):
    print(f"{v['mean_wave_life_frac']:6.3f}  {v['waves']:9.0f} waves  {v['launch_us_at_2.4GHz']:9.1f} us  {k[:110]}")
PY

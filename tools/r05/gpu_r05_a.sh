#!/bin/bash
# round 5, GPU call A: phase-1 prototype of the 8-point tile (timing + counters), fp64 arbiter at the benched shape, scene groups
R=$PWD; T=r05_a; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocm-smi --showclocks > $O/gpu.txt 2>&1
GNR_LIB=libgnr_p1.so timeout 300 python $R/tools/ab_chain_p1.py --out $O/p1.json > $O/p1.log 2>&1
( cd $R && timeout 600 python -m pytest tests/test_range_guard.py -m gpu -k benched -x -q > $O/arbiter.log 2>&1; cp gpurun_out/parity_errors.json $O/arbiter_parity_errors.json )
timeout 300 python $R/tools/ab_scene_groups.py --out $O/scene_groups.json > $O/scene_groups.log 2>&1
for SET in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $SET | tr ' ' '_')
  GNR_LIB=libgnr_p1.so timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc/$N -o p -- python $R/tools/ab_chain_p1.py --repeat 1 --iters 1 > $O/pmc_$N.log 2>&1
done
for G in 0 8; do
  for SET in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $SET | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/pmc_g$G/$N -o p -- python $R/tools/ab_scene_groups.py --once $G > $O/pmc_g${G}_$N.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py $O/pmc $O/p1_pmc.json > $O/p1_pmc_summary.log 2>&1
python tools/pmc_summary.py $O/pmc_g0 $O/groups_pmc_all32.json > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_g8 $O/groups_pmc_by8.json > /dev/null 2>&1
rm -rf $O/pmc $O/pmc_g0 $O/pmc_g8
tail -40 $O/p1.log; tail -5 $O/arbiter.log; tail -30 $O/scene_groups.log; cat $O/p1_pmc_summary.log | head -60

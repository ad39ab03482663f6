#!/bin/bash
# Per-kernel times of steady-state train steps for a list of library builds (rocprofv3 --kernel-trace, one run each):
#   tools/ab_train_kernels.sh OUTDIR libgnr.so libgnr_x.so ...   -> OUTDIR/<lib>_train_kernels.txt (tools/prof_summary.py, last 330 ms)
OUT=$1; shift
R=$PWD
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  rm -rf /tmp/abtr_$LIB
  GNR_LIB=$LIB rocprofv3 --kernel-trace --stats -d /tmp/abtr_$LIB -o t -- python $R/tools/train_step_bench.py --steps 3 --warmup 3 > $R/$OUT/${LIB}_train.json 2> /tmp/abtr_$LIB.log
  DB=$(find /tmp/abtr_$LIB -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB "$LIB: train_step_bench --steps 3 --warmup 3, last 330 ms" --last-ms 330 > $R/$OUT/${LIB}_train_kernels.txt
  echo "== $LIB"; grep -E "gnr::" $R/$OUT/${LIB}_train_kernels.txt | head -24 | cut -c1-140
  tail -1 $R/$OUT/${LIB}_train.json | cut -c1-300
done

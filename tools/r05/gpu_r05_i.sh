#!/bin/bash
# round 5, GPU call I: what the partner's spills cost: product against a measurement build whose partner drops the decoder .0 weight gradients (no spills)
R=$PWD; T=r05_i; O=$R/gpurun_out/$T; mkdir -p $O
for L in libgnr.so libgnr_pwx.so libgnr.so libgnr_pwx.so; do
  echo "== $L" >> $O/ab.txt
  GNR_LIB=$L timeout 300 python tools/time_volume_bwd.py --scenes 8 2>/dev/null | grep "view loop 1" >> $O/ab.txt
done
cat $O/ab.txt

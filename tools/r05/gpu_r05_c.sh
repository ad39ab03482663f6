#!/bin/bash
# round 5, GPU call C: backward twins on the f16 matrix cores -- tests of everything that trains, then A/B against the fp32-MFMA build
R=$PWD; T=r05_c; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_train_step.py tests/test_bwd_twins.py tests/test_determinism.py tests/test_pack_device.py tests/test_torch_ops.py tests/test_model_mirror.py tests/test_ray_tail.py -m gpu -x -q > $O/tests.log 2>&1
tail -5 $O/tests.log
for L in libgnr.so libgnr_bwdf32.so libgnr.so libgnr_bwdf32.so; do
  echo "== $L" >> $O/volume_bwd_ab.txt
  GNR_LIB=$L timeout 300 python tools/time_volume_bwd.py --scenes 8 >> $O/volume_bwd_ab.txt 2>&1
done
cat $O/volume_bwd_ab.txt
timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 > $O/train_pairs.json 2> $O/train_pairs.err
GNR_LIB=libgnr_bwdf32.so timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 > $O/train_f32.json 2> $O/train_f32.err
cut -c1-400 $O/train_pairs.json; echo; cut -c1-400 $O/train_f32.json

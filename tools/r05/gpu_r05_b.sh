#!/bin/bash
# round 5, GPU call B: whole GPU suite on the device packer / hardened operators, then the train step at the box's budget and at 2 CPUs
R=$PWD; T=r05_b; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; cp gpurun_out/parity_errors.json $O/parity_errors.json
timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 > $O/train_default.json 2> $O/train_default.err
timeout 400 python tools/train_step_bench.py --steps 16 --warmup 24 --cpus 2 --flat-exchange-steps 8 > $O/train_2cpu.json 2> $O/train_2cpu.err
tail -5 $O/suite.log; cut -c1-1500 $O/train_default.json; echo; cut -c1-1800 $O/train_2cpu.json; tail -3 $O/train_2cpu.err

R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e -o e -- python $R/tools/run_hot.py --batch 32 --iters 6 > $R/gpurun_out/prof_e.log 2>&1
cd $R
DB=$(find gpurun_out/prof_e -name "*.db" | head -1)
python tools/prof_summary.py $DB "run_hot B=32 x6" > gpurun_out/prof_e_summary.txt
head -20 gpurun_out/prof_e_summary.txt

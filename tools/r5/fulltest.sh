R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5full; mkdir -p $A; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $A/pytest_gpu_tail.txt
python tools/cpu_baseline_threads.py 16 32 64 128 256 > $A/cpu_baseline_threads.txt 2>&1
tail -8 $A/pytest_gpu_tail.txt; cat $A/cpu_baseline_threads.txt

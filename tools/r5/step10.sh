#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > /dev/null 2>&1
python $R/tools/gap_report.py /tmp/prof_c2 0 > $O/one_iter.txt 2>&1

# one-iteration kernel traces of C2 / C3 / C5 (rocprofv3 --kernel-trace) -> gpurun_out/prof_r3/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/r03_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $O/r03_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $O/r03_bench_c2_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/r03_c3_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c3 0 > $O/r03_c3_voxelflow_one_iteration.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --workload c5_cain_l2f_720p_b1_s1 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/r03_c5_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c5 0 > $O/r03_c5_cain_l2f_720p_one_iteration.txt 2>&1
ls -la $O

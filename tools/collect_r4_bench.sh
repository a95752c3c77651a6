# Round-4 bench artefacts: bench line, one-iteration kernel table under rocprofv3, layer table -> gpurun_out/art4/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art4; rm -rf $A; mkdir -p $A
python bench.py --steps 5 --warmup 2 > $A/r04_bench_line.json 2> $A/r04_bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $A/r04_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r04_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r04_bench_c2_kernel_stats.csv
cd $R
python tools/layer_table.py --workload c2_sepconv_256x448_b4_s5 --top 60 > $A/r04_layer_table_c2.txt 2>/dev/null
tail -3 $A/r04_bench_line.err; cat $A/r04_bench_line.json

# ws kernel: section trace + ablation variants (built by tools/build_variant.sh) -> gpurun_out/art4/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art4; mkdir -p $A
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > $A/r04_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r04_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r04_bench_c2_kernel_stats.csv
cd $R
python tools/ws_time.py 8 > $A/r04_ws_variants.txt 2>&1
for v in NOMFMA NOSTAGE NOAREAD NOMFMAONLY; do SAVFI_HIP_LIB=tools/scratch/variants/libsavfi_$v.so python tools/ws_time.py 8 >> $A/r04_ws_variants.txt 2>&1; done
SAVFI_HIP_LIB=tools/scratch/variants/libsavfi_trace.so python tools/ws_trace.py 8 > $A/r04_ws_section_trace.txt 2>&1
cat $A/r04_ws_variants.txt $A/r04_ws_section_trace.txt

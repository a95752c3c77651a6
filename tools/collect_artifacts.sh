set -x
rm -rf gpurun_out/art; mkdir -p gpurun_out/art
R=$GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 2 > gpurun_out/art/r02_bench_line.json 2> gpurun_out/art/r02_bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/art/r02_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $R/gpurun_out/art/r02_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $R/gpurun_out/art/r02_bench_c2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > gpurun_out/art/r02_hbm_traffic_sepconv.json 2> gpurun_out/art/r02_hbm_traffic.err
python tools/kernel_bench.py --batches 1,2,4,8 > gpurun_out/art/r02_kernel_bench.jsonl 2>/dev/null
python tools/tasks_bench.py > gpurun_out/art/r02_tasks_bench.jsonl 2>/dev/null
python tools/parity_report.py > gpurun_out/art/r02_parity_report.jsonl 2>/dev/null
for w in c3_voxelflow_metasgd_256x256_b8_s5 c4_sepconv_msl_256x448_b4_s5 c5_cain_l2f_720p_b1_s1 c1_cain_64x64_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5; do python bench.py --workload $w --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null >> gpurun_out/art/r02_other_configs.jsonl; done
python bench.py --workload c1_cain_64x64_b1_s1 --steps 5 --warmup 2 --no-cpu-baseline --graph-inner-loop 0 2>/dev/null >> gpurun_out/art/r02_other_configs.jsonl
python bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --graph-inner-loop 1 --task-streams 4 2>/dev/null >> gpurun_out/art/r02_other_configs.jsonl
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --graph-inner-loop 0 > /dev/null 2>&1; python $R/tools/gap_report.py /tmp/prof_c3 0 > $R/gpurun_out/art/r02_c3_voxelflow_one_iteration.txt 2>&1; cd $R
for cfg in "0 1 0" "0 1 4" "1 1 4" "0 2 2" "1 2 2" "1 4 0" "1 2 0"; do set -- $cfg; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --graph-inner-loop $1 --task-streams $2 --task-batch $3 2>/dev/null >> gpurun_out/art/r02_modes.jsonl; done
cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/wp_$c -- python $R/tools/wino_traffic.py > /dev/null 2>&1; done; cd $R
python tools/wino_traffic.py parse /tmp/wp_FETCH_SIZE /tmp/wp_WRITE_SIZE > gpurun_out/art/r02_wino_traffic.json 2>/dev/null
[ -f $R/tools/scratch/libsavfi_hip_trace.so ] || python tools/wino_trace.py --build > /dev/null 2>&1
SAVFI_HIP_LIB=$R/tools/scratch/libsavfi_hip_trace.so python tools/wino_trace.py 2>/dev/null | grep -v "^/opt" > gpurun_out/art/r02_wino_workgroup_phases.txt
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_valu tools/mfma_valu_overlap.hip 2>/dev/null && /tmp/mfma_valu > gpurun_out/art/r02_mfma_valu_overlap.txt 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fast-path 2>/dev/null >> gpurun_out/art/r02_modes.jsonl
python tools/wgrad_forms_bench.py > gpurun_out/art/r02_wgrad_forms.txt 2>/dev/null
ls -la gpurun_out/art

# Round-4 artefacts: everything profiles/r04_* of the final state is made from.  Run on the GPU box from the repo root:
#   bash tools/collect_r4_artifacts.sh        -> gpurun_out/art4/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art4; mkdir -p $A
cd $R
T0=$(date +%s); python bench.py > $A/r04_bench_line.json 2> $A/r04_bench_line.err; T1=$(date +%s); echo "python bench.py (default flags): wall $((T1 - T0)) s" > $A/r04_bench_default_run_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > $A/r04_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r04_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r04_bench_c2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r04_hbm_traffic_sepconv.json 2> $A/r04_hbm_traffic.err
bash tools/collect_r4_pmc_sepconv.sh > /dev/null 2>&1; cp gpurun_out/prof_r4/r04_pmc_sepconv_ws.txt $A/ 2>/dev/null
cd $R
for w in c3_voxelflow_metasgd_256x256_b8_s5 c4_sepconv_msl_256x448_b4_s5 c5_cain_l2f_720p_b1_s1 c1_cain_64x64_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5; do python bench.py --workload $w --steps 3 --warmup 2 2>/dev/null >> $A/r04_other_configs.jsonl; done
for cfg in "0 1 0" "1 4 0" "0 1 4" "1 1 4"; do set -- $cfg; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 --no-kernel-timer --graph-inner-loop $1 --task-streams $2 --task-batch $3 2>/dev/null >> $A/r04_modes.jsonl; done
python tools/layer_table.py --workload c2_sepconv_256x448_b4_s5 --top 60 > $A/r04_layer_table_c2.txt 2>/dev/null
python tools/kernel_bench.py --batches 1,2,4,8 > $A/r04_kernel_bench.jsonl 2>/dev/null
python tools/parity_report.py > $A/r04_parity_report.jsonl 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -9 > $A/r04_pytest_gpu_tail.txt

SAVFI_SEPCONV_SUBNETS_ONE_BY_ONE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 --no-kernel-timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['subnets']='one by one (SAVFI_SEPCONV_SUBNETS_ONE_BY_ONE=1)'; print(json.dumps(d))" >> $A/r04_modes.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > $A/r04_smoke.txt 2>&1; tail -3 $A/r04_smoke.txt

# Round-4 artefacts of the final state (SepConv op on frames of 8-bit images).  On the GPU box from the repo root:
#   bash tools/collect_r4_frames8.sh        -> gpurun_out/art4f/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art4f; rm -rf $A; mkdir -p $A
cd $R
T0=$(date +%s); python bench.py > $A/r04_bench_line.json 2> $A/r04_bench_line.err; T1=$(date +%s); echo "python bench.py (default flags): wall $((T1 - T0)) s" > $A/r04_bench_default_run_time.txt
# A/B on this box: the six-product kernels for every frame tensor
SAVFI_SEPCONV_NO_FRAMES8=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['sepconv']='six-product kernels (SAVFI_SEPCONV_NO_FRAMES8=1)'; print(json.dumps(d))" > $A/r04_frames8_ab.jsonl
SAVFI_SEPCONV_TAPS_PLANAR=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['sepconv']='frames8, taps and gradients planar (SAVFI_SEPCONV_TAPS_PLANAR=1)'; print(json.dumps(d))" >> $A/r04_frames8_ab.jsonl
SAVFI_SEPCONV_GRADS_PLANAR=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['sepconv']='frames8, taps unit-major, gradients planar (SAVFI_SEPCONV_GRADS_PLANAR=1)'; print(json.dumps(d))" >> $A/r04_frames8_ab.jsonl
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['sepconv']='frames8, taps and inner-loop gradients unit-major (default)'; print(json.dumps(d))" >> $A/r04_frames8_ab.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > $A/r04_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r04_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r04_bench_c2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r04_hbm_traffic_sepconv_frames8.json 2> $A/r04_hbm_traffic_frames8.err
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/sepconv_x6_pmc.py f8 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 sepconv > $A/r04_pmc_sepconv_ws_frames8.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/sepconv_x6_pmc.py f8 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 sepconv >> $A/r04_pmc_sepconv_ws_frames8.txt 2>&1
cd $R
python tools/frames8_time.py 8 > $A/r04_frames8_time.txt 2>&1
python tools/layer_table.py --workload c2_sepconv_256x448_b4_s5 --top 60 > $A/r04_layer_table_c2_final.txt 2>/dev/null
for w in c4_sepconv_msl_256x448_b4_s5 c3_voxelflow_metasgd_256x256_b8_s5 c5_cain_l2f_720p_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5 c1_cain_64x64_b1_s1; do python bench.py --workload $w --steps 3 --warmup 2 2>/dev/null >> $A/r04_other_configs_frames8.jsonl; done
python -m pytest tests -m gpu -q 2>&1 | tail -9 > $A/r04_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $A/r04_smoke.txt 2>&1; tail -3 $A/r04_smoke.txt
cat $A/r04_bench_line.json | cut -c1-1500; cat $A/r04_bench_default_run_time.txt; tail -3 $A/r04_pytest_gpu_tail.txt

# Round-3 artefacts: everything profiles/r03_* is made from.  Run on the GPU box from the repo root:
#   bash tools/collect_artifacts.sh        -> gpurun_out/art/
set -x
rm -rf gpurun_out/art; mkdir -p gpurun_out/art
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art
python bench.py --steps 5 --warmup 2 > $A/r03_bench_line.json 2> $A/r03_bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $A/r03_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r03_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r03_bench_c2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r03_hbm_traffic_sepconv.json 2> $A/r03_hbm_traffic.err
python tools/convk_bench.py --check > $A/r03_convk_check.jsonl 2>/dev/null
python tools/convk_bench.py --time --iters 10 > $A/r03_convk_bench.jsonl 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/probe tools/bf16_split_probe.hip 2>/dev/null && /tmp/probe > $A/r03_bf16_split_probe.txt 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/convk_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 > $A/r03_pmc_conv_kernels_pass1.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/convk_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 > $A/r03_pmc_conv_kernels_pass2.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc3 | head -24 > $A/r03_pmc_per_kernel_c2_iterations.txt 2>&1
cd $R
for w in c3_voxelflow_metasgd_256x256_b8_s5 c4_sepconv_msl_256x448_b4_s5 c5_cain_l2f_720p_b1_s1 c1_cain_64x64_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5; do python bench.py --workload $w --steps 3 --warmup 2 2>/dev/null >> $A/r03_other_configs.jsonl; done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $R/bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1; python $R/tools/gap_report.py /tmp/prof_c3 0 > $A/r03_c3_voxelflow_one_iteration.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --workload c5_cain_l2f_720p_b1_s1 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --graph-inner-loop 0 > /dev/null 2>&1; python $R/tools/gap_report.py /tmp/prof_c5 0 > $A/r03_c5_cain_l2f_720p_one_iteration.txt 2>&1
cd $R
for cfg in "0 1 0" "1 4 0" "0 1 4" "1 1 4" "0 2 2" "1 2 2"; do set -- $cfg; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --graph-inner-loop $1 --task-streams $2 --task-batch $3 2>/dev/null >> $A/r03_modes.jsonl; done
python bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --task-batch 4 2>/dev/null >> $A/r03_modes.jsonl
python bench.py --workload c3_voxelflow_metasgd_256x256_b8_s5 --steps 3 --warmup 2 --no-cpu-baseline --task-batch 0 2>/dev/null >> $A/r03_modes.jsonl
python tools/parity_report.py > $A/r03_parity_report.jsonl 2>/dev/null
python tools/kernel_bench.py --batches 1,2,4,8 > $A/r03_kernel_bench.jsonl 2>/dev/null
SAVFI_SEPCONV_F32_MFMA=1 python tools/kernel_bench.py --only sepconv --batches 2,8 --hw 256x448 > $A/r03_kernel_bench_sepconv_f32_mfma.jsonl 2>/dev/null
python tools/sepconv_x6_accuracy.py > $A/r03_sepconv_x6_accuracy.txt 2>/dev/null
SAVFI_SEPCONV_F32_MFMA=1 python tools/sepconv_x6_accuracy.py >> $A/r03_sepconv_x6_accuracy.txt 2>/dev/null
for w in c2_sepconv_256x448_b4_s5 c3_voxelflow_metasgd_256x256_b8_s5 c5_cain_l2f_720p_b1_s1; do python tools/layer_table.py --workload $w --top 60 >> $A/r03_layer_tables.txt 2>/dev/null; done
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcx1 -- python $R/tools/sepconv_x6_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcx1 | head -3 > $A/r03_pmc_sepconv_x6.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmcx2 -- python $R/tools/sepconv_x6_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcx2 | head -3 >> $A/r03_pmc_sepconv_x6.txt 2>&1
cd $R
ls -la $A

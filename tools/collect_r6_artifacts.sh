# Round-6 artefacts of the final tree.  On the GPU box from the repo root, in three calls (a lost call loses one part only):
#   bash tools/collect_r6_artifacts.sh a|b|c   -> gpurun_out/art6/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art6; mkdir -p $A
cd $R
part=${1:-a}
if [ "$part" = a ]; then
T0=$(date +%s); timeout 600 python bench.py > $A/r06_bench_line.json 2> $A/r06_bench_line.err; T1=$(date +%s); echo "python bench.py (default flags): wall $((T1 - T0)) s" > $A/r06_bench_default_run_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > $A/r06_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r06_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r06_bench_c2_kernel_stats.csv
# HBM traffic of the roofline kernel (separate --pmc passes, the guide's corrections: tools/hbm_traffic.py)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r06_hbm_traffic_sepconv.json 2> $A/r06_hbm_traffic.err
# the F(4x4) kernel: matrix-pipe occupancy, VALU, LDS conflicts
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc41 -- python $R/tools/r6/wino4_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc41 wino4 > $A/r06_pmc_wino4.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc42 -- python $R/tools/r6/wino4_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc42 wino4 >> $A/r06_pmc_wino4.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/sepconv_pair_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 sepconv > $A/r06_pmc_sepconv_ws.txt 2>&1

fi
if [ "$part" = b ]; then
cd $R
python tools/r6/wino4_check.py --time > $A/r06_wino4_check.txt 2>&1
(python tools/r6/wino4_time.py deep convk; python tools/r6/wino4_time.py small convk) > $A/r06_wino4_vs_convk.txt 2>&1
python tools/upsample_bench.py 2>&1 | grep lib > $A/r06_upsample_bench.txt
python tools/layer_table.py --workload c2_sepconv_256x448_b4_s5 --top 200 > $A/r06_layer_table_c2.txt 2>/dev/null
python tools/layer_table.py --workload c5_cain_l2f_720p_b1_s1 --top 60 > $A/r06_layer_table_c5.txt 2>/dev/null

fi
if [ "$part" = c ]; then
for w in c2script_sepconv_metasgd_adamax_256x448_b3_s3 c4_sepconv_msl_256x448_b4_s5 c3_voxelflow_metasgd_256x256_b8_s5 c5_cain_l2f_720p_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5 c1_cain_64x64_b1_s1; do timeout 400 python bench.py --workload $w --steps 3 --warmup 2 2>/dev/null >> $A/r06_other_configs.jsonl; done
# two ranks on ONE GPU over gloo: the multi-rank GPU path of the final tree (the driver's 8-GPU run is the only RCCL run there is)
SAVFI_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $A/r06_world2_one_gpu_gloo.json 2> $A/r06_world2_one_gpu_gloo.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -9 > $A/r06_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $A/r06_smoke.txt 2>&1; tail -3 $A/r06_smoke.txt
cat $A/r06_bench_line.json | cut -c1-1200; cat $A/r06_bench_default_run_time.txt; tail -3 $A/r06_pytest_gpu_tail.txt

fi

# Round-5 artefacts of the final tree.  On the GPU box from the repo root:   bash tools/collect_r5_artifacts.sh   -> gpurun_out/art5/
set -x
R=$GRAFT_REPO_ROOT
A=$R/gpurun_out/art5; rm -rf $A; mkdir -p $A
cd $R
V=$R/tools/scratch/variants
T0=$(date +%s); python bench.py > $A/r05_bench_line.json 2> $A/r05_bench_line.err; T1=$(date +%s); echo "python bench.py (default flags): wall $((T1 - T0)) s" > $A/r05_bench_default_run_time.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4"
tag() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['mode']['sepconv']=sys.argv[1]; print(json.dumps(d))" "$1"; }
# the ladder on this box
SAVFI_SEPCONV_PAIR_TWO_LAUNCHES=1 $B 2>/dev/null | tag "two launches per pass (SAVFI_SEPCONV_PAIR_TWO_LAUNCHES=1)" > $A/r05_pair_ab.jsonl
$B 2>/dev/null | tag "one launch per pass (default)" >> $A/r05_pair_ab.jsonl
SAVFI_UPSAMPLE_BWD_FORM=1 $B 2>/dev/null | tag "default, bilinear x2 backward tiled (SAVFI_UPSAMPLE_BWD_FORM=1)" >> $A/r05_pair_ab.jsonl
SAVFI_WGRAD_NO_BIAS=1 $B 2>/dev/null | tag "default, bias sums as their own pass (SAVFI_WGRAD_NO_BIAS=1)" >> $A/r05_pair_ab.jsonl
SAVFI_NO_CONV_CHAIN=1 $B 2>/dev/null | tag "default, every conv + ReLU with its own derivative pass (SAVFI_NO_CONV_CHAIN=1: no deferral to the next convolution / bilinear x2 / pool-and-skip adjoint)" >> $A/r05_pair_ab.jsonl
SAVFI_WGRAD3_NO_RING=1 $B 2>/dev/null | tag "default, 3x3 weight gradients routed as before the all-taps kernel (SAVFI_WGRAD3_NO_RING=1)" >> $A/r05_pair_ab.jsonl
$B 2>/dev/null | tag "one launch per pass (default), again" >> $A/r05_pair_ab.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong-c4 > $A/r05_bench_line_profiled.json 2>/dev/null
python $R/tools/gap_report.py /tmp/prof_c2 0 > $A/r05_bench_c2_one_iteration.txt 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); head -40 "$f" > $A/r05_bench_c2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r05_hbm_traffic_sepconv.json 2> $A/r05_hbm_traffic.err
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/sepconv_pair_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 sepconv > $A/r05_pmc_sepconv_ws_final.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/sepconv_pair_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 sepconv >> $A/r05_pmc_sepconv_ws_final.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/tools/sepconv_pair_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc3 sepconv >> $A/r05_pmc_sepconv_ws_final.txt 2>&1
cd $R
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace.py 8 f8 3 > $A/r05_ws_trace_dma.txt 2>&1
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace.py 4 f8 3 > $A/r05_ws_trace_dma_b4.txt 2>&1
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace_fwd.py 8 1 > $A/r05_ws_trace_fwd.txt 2>&1
python tools/frames8_time.py 8 > $A/r05_frames8_time.txt 2>&1
python tools/frames8_time.py 4 256 448 bwd_frames8_unit16,fwd_frames8_unit16 >> $A/r05_frames8_time.txt 2>&1
tools/scratch/membench > $A/r05_membench.txt 2>&1
python tools/glue_bench.py 2>&1 | grep map > $A/r05_glue_bench.txt
python tools/upsample_bench.py 2>&1 | grep lib > $A/r05_upsample_bench.txt
SAVFI_UPSAMPLE_BWD_FORM=1 python tools/upsample_bench.py 2>&1 | grep lib | sed 's/"lib": "default"/"lib": "tiled form (SAVFI_UPSAMPLE_BWD_FORM=1)"/' >> $A/r05_upsample_bench.txt
(python tools/wgrad3_forms_time.py c2; python tools/wgrad3_forms_time.py c5) 2>&1 | grep "^{" > $A/r05_wgrad3_forms.txt
(SAVFI_WGRAD3_FORM=0 python tools/wgrad3_forms_time.py c2; SAVFI_WGRAD3_FORM=0 python tools/wgrad3_forms_time.py c5) 2>&1 | grep "^{" | sed 's/^{/{"form": "tap-split kernel (SAVFI_WGRAD3_FORM=0)", /' >> $A/r05_wgrad3_forms.txt
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcw1 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcw1 wgrad > $A/r05_pmc_wgrad3.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmcw2 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmcw2 wgrad >> $A/r05_pmc_wgrad3.txt 2>&1
cd $R
python tools/layer_table.py --workload c2_sepconv_256x448_b4_s5 --top 60 > $A/r05_layer_table_c2.txt 2>/dev/null
SAVFI_WGRAD3_FORM=0 python bench.py --workload c5_cain_l2f_720p_b1_s1 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['config']['note']='3x3 weight gradients on the tap-split kernel (SAVFI_WGRAD3_FORM=0)'; print(json.dumps(d))" > $A/r05_c5_wgrad_ab.jsonl
for w in c4_sepconv_msl_256x448_b4_s5 c3_voxelflow_metasgd_256x256_b8_s5 c5_cain_l2f_720p_b1_s1 rrin_256x448_b4_s5 superslomo_256x448_b4_s5 c1_cain_64x64_b1_s1; do python bench.py --workload $w --steps 3 --warmup 2 2>/dev/null >> $A/r05_other_configs.jsonl; done
python -m pytest tests -m gpu -q 2>&1 | tail -9 > $A/r05_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $A/r05_smoke.txt 2>&1; tail -3 $A/r05_smoke.txt
cat $A/r05_bench_line.json | cut -c1-1500; cat $A/r05_bench_default_run_time.txt; tail -3 $A/r05_pytest_gpu_tail.txt

R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5s7; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace_fwd.py 8 1 > $A/fwd_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for n in default noaread nostage nomfma; do
  if [ $n == default ]; then L=$R/meta-interpolation_amd/lib/libsavfi_hip.so; else L=$V/libsavfi_$n.so; fi
  rm -rf /tmp/pmcx
  SAVFI_HIP_LIB=$L rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmcx -- python $R/tools/sepconv_pair_pmc.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pmcx sepconv 2>&1 | grep "<true" | sed "s/^/$n /" >> $A/lds_conflicts.txt
done
cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
python tools/hbm_traffic.py parse /tmp/pmc_f /tmp/pmc_w > $A/r05_hbm_traffic_sepconv.json 2> $A/r05_hbm_traffic.err
tail -16 $A/fwd_trace.txt | cut -c1-420; cat $A/lds_conflicts.txt | cut -c1-330; python -c "
import json; d=json.load(open('$A/r05_hbm_traffic_sepconv.json')); print(d['calibration']); print({k: round(v['traffic_over_algorithmic'],3) for k,v in d['kernels'].items()})"

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w3; mkdir -p $O; cd $R
V=$R/tools/scratch/variants
for n in default ringabl1 ringabl2 ringabl3; do
  if [ $n == default ]; then L=$R/meta-interpolation_amd/lib/libsavfi_hip.so; else L=$V/libsavfi_$n.so; fi
  SAVFI_HIP_LIB=$L timeout 200 python tools/wgrad3_forms_time.py c2s 2>&1 | grep "^{" | sed "s/^/$n /" >> $O/abl.txt
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 wgrad > $O/pmc_ring.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 wgrad >> $O/pmc_ring.txt 2>&1

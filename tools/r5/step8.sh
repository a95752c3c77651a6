#!/bin/bash
# PMC of the all-taps weight gradient
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 wgrad > $O/pmc.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc2 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 wgrad >> $O/pmc.txt 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/tools/wgrad3_pmc.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pmc3 wgrad >> $O/pmc.txt 2>&1

R=$GRAFT_REPO_ROOT; cd $R
V=$R/tools/scratch/variants
for n in wpv16 wpv24; do SAVFI_HIP_LIB=$V/libsavfi_$n.so python tools/r5/dbg2.py $V/libsavfi_$n.so 2>&1 | tail -3; done
VARIANTS="wpv16 wpv24" bash tools/r5/step4.sh

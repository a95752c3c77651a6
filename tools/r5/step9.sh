R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_sepconv_frames8_gpu.py tests/test_hip_ops_gpu.py -x -q -k "sepconv or frames8 or pair" 2>&1 | tail -3
VARIANTS="prev" bash tools/r5/step4.sh

# HBM traffic of the sepconv kernels: separate FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/prof_r4/r04_hbm_traffic_sepconv.json
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tf -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/tw -- python $R/tools/hbm_traffic.py run > /dev/null 2>&1
cd $R && python tools/hbm_traffic.py parse /tmp/tf /tmp/tw > $O/r04_hbm_traffic_sepconv.json 2> $O/r04_hbm_traffic.err
python - <<PY
import json
d=json.load(open("$O/r04_hbm_traffic_sepconv.json"))
for k,v in d["kernels"].items(): print(k, round(v["traffic_over_algorithmic"],3))
PY

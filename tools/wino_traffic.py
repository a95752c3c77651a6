"""HBM-side traffic of wino_conv3x3 on three SepConv layer shapes (N = 8): run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE and
parse with `python tools/wino_traffic.py parse <dir_fetch> <dir_write>` (FETCH_SIZE x 2: the gfx950 correction of tools/hbm_traffic.py).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "parse":
    import csv, glob, json
    shapes = [(32,32,384,512),(128,128,96,128),(51,51,258,450)]
    out = {}
    for d, name, corr in ((sys.argv[2], "FETCH_SIZE", 2.0), (sys.argv[3], "WRITE_SIZE", 1.0)):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "wino_conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == name]
        for k, (ci, co, h, w) in enumerate(shapes):
            v = vals[3 * k: 3 * k + 3]
            e = out.setdefault("%d->%d@%dx%d N=8" % (ci, co, h, w), {"input_bytes": 8 * ci * h * w * 4, "output_bytes": 8 * co * h * w * 4})
            e[name + "_bytes"] = round(sum(v) / len(v) * 1024 * corr)
    for e in out.values():
        e["fetch_over_input"] = round(e["FETCH_SIZE_bytes"] / e["input_bytes"], 3)
        e["write_over_output"] = round(e["WRITE_SIZE_bytes"] / e["output_bytes"], 3)
    print(json.dumps(out, indent=1))
    sys.exit(0)
import torch
from meta_interpolation_amd import hip_ops
for ci,co,h,w in [(32,32,384,512),(128,128,96,128),(51,51,258,450)]:
    x = torch.randn(8,ci,h,w,device='cuda'); wt = torch.randn(co,ci,3,3,device='cuda')/(3*ci**.5); b = torch.randn(co,device='cuda')
    for _ in range(3): y = hip_ops.conv3x3(x,wt,b,0,0.0,1)
    torch.cuda.synchronize()

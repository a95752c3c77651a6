R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5s2; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
for n in nomfma; do SAVFI_HIP_LIB=$V/libsavfi_$n.so python tools/frames8_time.py 8 256 448 bwd_frames8_unit16 2>&1 | grep op | sed "s/^/$n /"; done
python tools/frames8_time.py 8 256 448 bwd_frames8_unit16 2>&1 | grep op
python - <<'P'
import torch, time
x = torch.empty(400*1024*1024//4, device='cuda'); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): y.copy_(x)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("copy 400 MB: %.1f us, %.0f GB/s read+write" % (1e3 * ms, 2 * x.numel() * 4 / ms / 1e6))
P

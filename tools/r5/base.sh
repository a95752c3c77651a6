# round-5 baseline of the round-4 tree on this box
R=$GRAFT_REPO_ROOT; A=$R/gpurun_out/r5base; mkdir -p $A; cd $R
V=$R/tools/scratch/variants
python tools/frames8_time.py 8 256 448 bwd_frames8,bwd_frames8_unit16,fwd_frames8,fwd_frames8_unit16 > $A/time_b8.txt 2>&1
python tools/frames8_time.py 4 256 448 bwd_frames8,bwd_frames8_unit16,fwd_frames8_unit16 > $A/time_b4.txt 2>&1
for n in nostage nomfma loadhit; do SAVFI_HIP_LIB=$V/libsavfi_$n.so python tools/frames8_time.py 8 256 448 bwd_frames8_unit16 2>&1 | sed "s/^/$n /" >> $A/time_variants.txt; done
SAVFI_HIP_LIB=$V/libsavfi_trace.so python tools/ws_trace.py 8 f8 3 > $A/trace_u16.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strong-c4 > $A/bench.json 2>$A/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c2 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-strong-c4 > /dev/null 2>&1
f=$(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1)
python - "$f" > $A/sepconv_launches.txt <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'sepconv' in n:
        d[n[:60]+' grid='+r.get('Grid_Size_X','?')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    v2=sorted(v); print(k, len(v), 'mean %.1f min %.1f med %.1f max %.1f'%(sum(v)/len(v), v2[0], v2[len(v2)//2], v2[-1]))
P
cd $R; cat $A/time_b8.txt $A/time_b4.txt $A/time_variants.txt; tail -14 $A/trace_u16.txt | cut -c1-600; cat $A/sepconv_launches.txt; cut -c1-300 $A/bench.json

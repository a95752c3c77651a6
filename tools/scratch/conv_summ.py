import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
g=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'wino_conv3x3' in n:
        key=(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
        g.setdefault(key, []).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in g.items():
    # fwd and bwd of the layers sharing a grid: print distinct clusters
    v2=sorted(v)
    print(k, len(v), 'min', round(v2[0],1), 'median', round(v2[len(v2)//2],1), 'max', round(v2[-1],1))

cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ops_pmc
mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  FUSED=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o pmc -- python $GRAFT_REPO_ROOT/tools/ops_bench.py > $O/$C.log 2>&1
done
python - <<'PY'
import csv, os, glob, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ops_pmc'
tot=collections.defaultdict(lambda: collections.defaultdict(list))
for C in ('FETCH_SIZE','WRITE_SIZE'):
    for r in csv.DictReader(open(glob.glob(O+'/'+C+'/*counter_collection.csv')[0])):
        if r['Counter_Name']==C and ('query_ball' in r['Kernel_Name'] or 'group_point' in r['Kernel_Name']):
            tot[r['Kernel_Name'][:40]+' grid'+r['Grid_Size']][C].append(float(r['Counter_Value']))
s=0
for k,v in tot.items():
    fe=sum(v['FETCH_SIZE'])/len(v['FETCH_SIZE']); wr=sum(v['WRITE_SIZE'])/len(v['WRITE_SIZE'])
    b=(2*fe+wr)*1024; s+=b
    print(k, 'fetchKB %.0f writeKB %.0f bytes %.0f'%(fe,wr,b))
print('TOTAL bytes per batch', round(s))
PY

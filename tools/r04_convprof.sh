export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r04
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04/cprof -o cp -- python $R/tools/conv_bench.py --min-cin 512 ${CONVB_ARGS} > $R/gpurun_out/r04/convprof_out.txt 2>&1
cd $R
DB=$(find gpurun_out/r04/cprof -name "*.db" | head -1)
python - <<PY
import sqlite3
cur=sqlite3.connect("$DB").cursor()
rows=cur.execute("select name, grid_x, (end-start)/1000.0 from kernels order by start").fetchall()
seq=[]
for n,g,t in rows:
    if not (n.startswith("void conv1x1") or n.startswith("conv_gn") or n.startswith("gn_")): continue
    k=(n.split("(")[0][:44],g)
    if seq and seq[-1][0]==k: seq[-1][1].append(t)
    else: seq.append([k,[t]])
# collapse repeating groups: print run-length encoded
for k,v in seq:
    print("%-46s grid %9d  x%2d  avg %9.1f us  min %9.1f" % (k[0],k[1],len(v),sum(v)/len(v),min(v)))
PY
rm -rf gpurun_out/r04/cprof
cat gpurun_out/r04/convprof_out.txt

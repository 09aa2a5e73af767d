cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq3 && timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VSKIPPED -d /tmp/sq3 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --lean --steps 6 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob('/tmp/sq3/**/*counter_collection.csv', recursive=True): rows += list(csv.DictReader(open(f)))
acc=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    if 'pt_render_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(acc): print(k, acc[k]/n[k], n[k])
PY

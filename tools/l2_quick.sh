# L2 hits and misses of the render kernel for one build of the library: bash tools/l2_quick.sh <tag> <lib path relative to the repo or ""> <one_dispatch args...>
set -e
tag=$1; lib=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=""
if [ -n "$lib" ]; then L="--lib $R/$lib"; fi
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/l2_${tag} -o g --output-format csv -- python3 $R/tools/one_dispatch.py "$@" $L > $R/gpurun_out/l2_${tag}.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$R/gpurun_out/l2_${tag}/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
k = [r for r in rows if "pt_render" in r["Kernel_Name"]]
last = max(int(r["Dispatch_Id"]) for r in k)
c = {}
for r in k:
    if int(r["Dispatch_Id"]) == last:
        c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
print("${tag}", {k_: v for k_, v in c.items()}, "l2_hit_rate", round(hit / max(1.0, hit + miss), 4), "fetch_GB(x2)", round(c.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, 3))
PY
tail -1 $R/gpurun_out/l2_${tag}.log

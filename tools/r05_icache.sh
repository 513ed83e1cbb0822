# instruction-cache requests / misses per kernel of a human-scale step (the block sorter is 50 k instructions).  Usage: bash tools/r05_icache.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/ic
RH_SUB_BATCHES=1 timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d /tmp/ic -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>$O/r05_icache.err
f=$(find /tmp/ic -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("$f")):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQC_ICACHE_REQ": n[k]+=1
rows=sorted(acc.items(), key=lambda kv:-kv[1].get("SQC_ICACHE_REQ",0))[:25]
out=open("$O/r05_icache.txt","w")
for k,v in rows:
    req=v.get("SQC_ICACHE_REQ",0); mis=v.get("SQC_ICACHE_MISSES",0)
    line="%-70s launches %4d  icache req %14.0f  misses %12.0f  (%.2f %%)"%(k[:70], n[k], req, mis, 100*mis/req if req else 0)
    print(line); out.write(line+"\n")
PY

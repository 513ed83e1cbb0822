# what the largest kernels of a human-scale step spend their issue slots on: instruction mix and LDS conflict counters, one pass per group.  Usage: bash tools/r05_pmc_mix.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
pass() {  # tag counters...
  tag=$1; shift
  rm -rf /tmp/pm_$tag
  RH_SUB_BATCHES=1 timeout -k 10 500 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pm_$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-h2d > /dev/null 2>$O/r05_mix_$tag.err
  f=$(find /tmp/pm_$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$tag: no counters collected"; tail -3 $O/r05_mix_$tag.err; return; }
  python - "$f" "$O/r05_mix_$tag.txt" "$@" <<'PY'
import csv, collections, sys
f, out, names = sys.argv[1], sys.argv[2], sys.argv[3:]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void ","")][r["Counter_Name"]]+=float(r["Counter_Value"])
rows=sorted(acc.items(), key=lambda kv:-kv[1].get(names[0],0))[:14]
with open(out,"w") as o:
    hdr="%-56s"%"kernel"+"".join("%22s"%n for n in names)
    print(hdr); o.write(hdr+"\n")
    for k,v in rows:
        line="%-56s"%k[:56]+"".join("%22.0f"%v.get(n,0) for n in names)
        print(line); o.write(line+"\n")
PY
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass lds SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU

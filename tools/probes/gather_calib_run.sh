# FETCH_SIZE per access class (tools/probes/gather_calib.hip) + the request-size counters that say what a request really fetched.  Usage (GPU box): bash tools/probes/gather_calib_run.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/probes/gather_calib.hip -o /tmp/gather_calib || exit 1
/tmp/gather_calib > $O/r06_calib_plain.txt 2>&1
rm -f $O/r06_calib_counters.txt
for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1); rm -rf /tmp/cal_$tag
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/cal_$tag -o p -- /tmp/gather_calib > /dev/null 2>$O/r06_calib_$tag.err
  f=$(find /tmp/cal_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' >> $O/r06_calib_counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if k.startswith("cal_"): print(k, {c: x[-1] for c, x in v.items()})
PY
done
cat $O/r06_calib_plain.txt; cat $O/r06_calib_counters.txt

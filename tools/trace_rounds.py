# dev tool: per chunk-round kernel durations from a rocprofv3 --kernel-trace csv under /tmp/prof (single-stream run)
import csv,glob,sys
f=glob.glob("/tmp/prof/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rnd=-1
per={}
pat=sys.argv[1] if len(sys.argv)>1 else ""
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if n.startswith("k_events_norm"): rnd+=1
    if n.startswith("__amd"): continue
    per.setdefault(rnd,[]).append((n,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
for rd,l in sorted(per.items()):
    print("round",rd,"total",round(sum(x[1] for x in l),2))
    print("   "+" ".join(f"{n.replace('k_','')[:16]}={t:.2f}" for n,t in l if t>=0.05 and pat in n))

import csv,glob,collections,sys
f=glob.glob("/tmp/prof/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tot=collections.OrderedDict(); first={}
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    t=tot.setdefault(n,[0,0.0,0.0]); t[0]+=1; t[1]+=d
    if n not in first: first[n]=d
print("kernel calls total_ms first_launch_ms")
for n,(c,d,_) in sorted(tot.items(), key=lambda kv:-kv[1][1]): print(f"{n[:44]:44s} {c:4d} {d:9.2f} {first[n]:8.2f}")
print("sum", sum(v[1] for v in tot.values()))

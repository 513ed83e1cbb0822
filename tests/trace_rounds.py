import csv,glob,sys
f=glob.glob("/tmp/prof/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rnd=-1
want=set(int(x) for x in sys.argv[1:]) if len(sys.argv)>1 else {0,5,9}
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void ","")
    if n.startswith("k_events_norm"): rnd+=1
    if rnd in want and not n.startswith("__amd"):
        print(rnd, f"{n[:28]:28s}", round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,3), "ms", "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "lds", r.get("LDS_Block_Size"))

import csv, glob, sys, collections
for d in sys.argv[1:]:
    rows=[]
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id") or r.get("Stream_Id")))
    rows.sort()
    ks=[r for r in rows if "k_step" in r[2]]
    print(d, len(rows), "kernels", len(ks), "k_step launches")
    t0=ks[len(ks)//2][0]
    for s,e,n,q in [r for r in rows if r[0]>=t0][:40]:
        print("  %9.1f -> %9.1f  (%7.1f us)  q=%s  %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,n))

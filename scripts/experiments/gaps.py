import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "serhip" in r["Kernel_Name"]][-16:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-28:]
    print(f"{name:30s} dur {(e-s)/1e3:8.1f} us  gap before {((s-prev)/1e3 if prev else 0):7.1f} us")
    prev = e

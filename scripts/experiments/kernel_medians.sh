cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tw && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --cpu-clips 0 > /tmp/tw.log 2>&1; python3 - <<EOF
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("/tmp/tw/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "serhip" in n: d[n[:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    v.sort(); print(k, len(v), "median us", v[len(v)//2])
EOF

cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --mode gray --steps 5 --warmup 1 --cpu-baseline 0 --host-io 0 --traffic off > /dev/null 2>&1; python - <<PY
import csv,glob
for f in glob.glob("/tmp/kt/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "ray_" in r["Name"] or "gray_decode" in r["Name"] or "DeviceScan" in r["Name"] or "fill" in r["Name"].lower(): print(r["Name"].split("(")[0][:60], r["Calls"], r["AverageNs"], r["MinNs"])
PY

cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt2; SLR_WHAT=mf rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt2 -o kt -- python $GRAFT_REPO_ROOT/profiles/prof_driver.py > /dev/null 2>&1; python - <<PY
import csv,glob
for f in glob.glob("/tmp/kt2/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "mf_rect_decode_dma" in r["Name"]: print(r["Name"].split("(")[0][:70], r["Calls"], r["AverageNs"], r["MinNs"])
PY

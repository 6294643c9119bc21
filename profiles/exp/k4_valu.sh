#!/bin/bash
# NOTE (round 2): the stop hook is compiled only into -DSLR_DEBUG_HOOKS builds: make -C structure-light-reconstructor_amd/csrc clean all CXXFLAGS+=-DSLR_DEBUG_HOOKS
# VALU / LDS instruction counts of the binned K4 per phase (SLR_DEBUG_K4_STOP ablation), per pixel at 4096x3000
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
cat > /tmp/k4drv.py <<PY
import importlib, os, sys, torch
sys.path.insert(0, "$REPO")
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H); ctx.set_calibration(calib)
st = synth.render_mf_stack(W, H, seed=1234, device=dev); torch.cuda.synchronize()
dec = [ctx.mf_decode(st[c], 40) for c in range(2)]
for _ in range(3): ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], want_match=False)
ctx.synchronize(); ctx.close()
PY
for stop in 1 2 4 5 0; do
  rm -rf /tmp/k4p; SLR_DEBUG_K4_STOP=$stop rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -f csv -d /tmp/k4p -o k4 -- python /tmp/k4drv.py > /tmp/k4p.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/k4p/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'mf_match_binned' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
px=4096*3000.0
print("stop=$stop", "  ".join("%s %.1f/px" % (k.replace('SQ_INSTS_',''), sum(v)/len(v)*64/px) if 'INSTS' in k else "%s %.3g" % (k, sum(v)/len(v)) for k,v in sorted(acc.items())))
PY
done

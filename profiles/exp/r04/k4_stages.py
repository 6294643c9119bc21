"""K4 stage times (debug-hooks build: the kernel returns after stage N without writing) per shape: python profiles/exp/r04/k4_stages.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.getcwd())
slr = importlib.import_module("structure-light-reconstructor_amd"); synth = importlib.import_module("structure-light-reconstructor_amd.synth")
capi = slr.capi
capi._lib = None; capi.LIB_PATH = os.path.join(os.getcwd(), "profiles/exp/ab/so/k4_%s.so" % os.environ.get("K4_VAR", "dbg")); capi.load_library()
W, H = 4096, 3000; dev = torch.device("cuda", 0); ctx = slr.Context(0)
calib, _ = synth.make_calibration(W, H); ctx.set_calibration(calib)
st = synth.render_mf_stack(W, H, seed=1234, device=dev); torch.cuda.synchronize()
dec = [ctx.mf_decode(st[c], 40) for c in range(2)]
xyz = torch.empty((H, W, 3), dtype=torch.float32, device=dev); has = torch.empty((H, W), dtype=torch.uint8, device=dev)
def run(): ctx.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], want_match=False, xyz=xyz, has=has)
for algo in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4,5,6").split(",")]:
    ctx.set_option(capi.OPT_MF_MATCH_ALGO, algo)
    line = []
    for stop in (1, 2, 4, 5, 0):
        ctx.set_option(capi.OPT_DEBUG_K4_STOP, stop)
        for _ in range(3): run()
        ctx.synchronize(); ctx.timer_begin()
        for _ in range(20): run()
        line.append("stop%d %.1f" % (stop, ctx.timer_end() / 20 * 1e3))
    print("algo %d: %s us" % (algo, "  ".join(line)), flush=True)

"""Round 5, VERDICT r4 item 3 (decode -> match in one persistent launch): what is there to win by running the HBM-side fused
decode and the VALU/LDS-side K4 BESIDE each other instead of behind each other?  The cheapest honest probe: the timed region's own
batch (8 frames, verged 0.2 rad rig, grouped launches) on ONE stream, against two such batches on TWO streams whose launches the
hardware is free to overlap (decode of batch B beside the K4 of batch A), with the decode's resident workgroups limited to 2 / 1 per
CU so that K4's workgroups find LDS and wave slots beside them (SLR_OPT_DEBUG_RECT_RESIDENT), and K4 alone beside a decode alone.
Prints ms per frame; same process, clocks settled."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.pop("SLR_POISON_OUTPUTS", None); os.environ.pop("SLR_POISON_SCRATCH", None)
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H, F = 4096, 3000, 8
dev = torch.device("cuda", 0)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
cA, cB = slr.Context(0, stream=sA), slr.Context(0, stream=sB)
rig = synth.make_verged_rig(W, H, 0.2, -0.15)
for c in (cA, cB):
    c.set_calibration(rig["calib"])
    synth.install_verged_maps(c, rig, W, H)
stack = torch.stack([synth.render_mf_stack(W, H, seed=1234 + f, noise=2, device=dev) for f in range(F)])
torch.cuda.synchronize()
out = [(torch.empty((F, H, W, 3), dtype=torch.float32, device=dev), torch.empty((F, H, W), dtype=torch.uint8, device=dev)) for _ in range(2)]
ph = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
def batch(c, k): c.reconstruct_mf_batch(stack, 40, True, W=W, xyz=out[k][0], has=out[k][1])
def sync(): cA.synchronize(); cB.synchronize(); torch.cuda.synchronize()
def timed(fn, n=12, frames=F):
    for _ in range(6): fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n / frames * 1e3
print("one stream, one batch of 8 after the other          %.4f ms per frame" % timed(lambda: batch(cA, 0)))
def both(): batch(cA, 0); batch(cB, 1)
print("two streams, two batches of 8 free to overlap       %.4f ms per frame" % timed(both, frames=2 * F))
for res in (512, 256):
    for c in (cA, cB): c.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, res)
    print("  decode limited to %d resident workgroups: one stream %.4f, two streams %.4f ms per frame" % (
        res, timed(lambda: batch(cA, 0)), timed(both, frames=2 * F)))
for c in (cA, cB): c.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, 0)
# K4 alone (stream B, phases of frame 0 decoded once) beside the pair decode alone (stream A)
p2, _ = cA.mf_rectify_decode_pair(stack[0, 0], stack[0, 1], 40, W=W, want_valid=False, phase=ph)
sync()
ones = torch.ones((H, W), dtype=torch.uint8, device=dev)
xk, hk = torch.empty((H, W, 3), dtype=torch.float32, device=dev), torch.empty((H, W), dtype=torch.uint8, device=dev)
def dec(): cA.mf_rectify_decode_pair(stack[1, 0], stack[1, 1], 40, W=W, want_valid=False, phase=[out[0][0][0, :, :, 0], out[0][0][0, :, :, 1]] if False else ph2)
ph2 = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
def k4(): cB.mf_triangulate(ph[0], ones, ph[1], ones, want_match=False, xyz=xk, has=hk)
td, tk = timed(dec, 40, 1), timed(k4, 40, 1)
def dk(): dec(); k4()
print("single-frame launches: decode alone %.4f ms, K4 alone %.4f ms, both on two streams %.4f ms (sum %.4f)" % (td, tk, timed(dk, 40, 1), td + tk))
for res in (512, 256):
    cA.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, res)
    td = timed(dec, 40, 1)
    print("  decode on %d resident workgroups: alone %.4f ms, beside K4 %.4f ms (sum %.4f)" % (res, td, timed(dk, 40, 1), td + tk))

"""Round 6 experiment: the fused Gray decode compiled for 8 waves per SIMD (SLR_OPT_DEBUG_FLAGS bit 9: four workgroups per CU) against the
shipped form (the experiment's kernel is NOT in the tree any more -- LABNOTES section 12 has its description and numbers; this
script is the harness it was measured with), same process, 4096x3000 GRAY_EPI stack on the verged rig: bit-equality of the whole-path outputs and per-kernel times."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
slr = importlib.import_module("structure-light-reconstructor_amd")
synth = importlib.import_module("structure-light-reconstructor_amd.synth")
W, H, F = 4096, 3000, 4
dev = torch.device("cuda", 0)
ctx = slr.Context(0)
rig = synth.make_verged_rig(W, H, 0.2, -0.15)
ctx.set_calibration(rig["calib"])
synth.install_verged_maps(ctx, rig, W, H)
ncol = synth.gray_num_bits(W)
stack = torch.stack([synth.render_gray_stack(W, H, W, seed=50 + f, noise=2, device=dev) for f in range(F)])
torch.cuda.synchronize()
res = {}
for rep in range(2):
    for flags, resident in ((0, 0), (512, 0), (512, 768), (512, 512), (0, 512)):
        ctx.set_option(slr.capi.OPT_DEBUG_FLAGS, flags)
        ctx.set_option(slr.capi.OPT_DEBUG_RECT_RESIDENT, resident)
        for _ in range(6):
            out = ctx.reconstruct_batch(slr.capi.MODE_GE, stack, 40, 0, n_col_bits=ncol, scan_w=W, rectify=True, W=W)
        ctx.synchronize()
        ctx.set_option(slr.capi.OPT_PROFILE_STRIDE, 1)
        ctx.profile_enable(True); ctx.profile_reset()
        ctx.timer_begin()
        for _ in range(10):
            out = ctx.reconstruct_batch(slr.capi.MODE_GE, stack, 40, 0, n_col_bits=ncol, scan_w=W, rectify=True, W=W)
        ms = ctx.timer_end() / (10 * F)
        pr = ctx.profile(); ctx.profile_enable(False)
        res[flags] = (out[0].clone(), out[1].clone())
        print("flags %3d resident %4d: %.4f ms per frame  " % (flags, resident, ms) + "  ".join("%s %.1f us" % (k, v[0] / v[1] * 1e3) for k, v in sorted(pr.items())))
print("bit-equal:", bool(torch.equal(res[0][0].view(torch.int32), res[512][0].view(torch.int32)) and torch.equal(res[0][1], res[512][1])),
      "matched fraction %.3f" % res[0][1].float().mean().item())

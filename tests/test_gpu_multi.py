"""Row 8(e) on the GPU box: the one-process multi-GPU entry of the C ABI (slr_reconstruct_mf_multi: frame f -> ctx f mod n, peer
copies assemble the cloud), its helpers (ordered prefix index, valid-point compaction) and the torchrun + collective leg of
bench.py with two ranks.  The box has ONE GPU: several contexts / ranks share device 0 -- the code path is the same, the peer
copies and the all-gather just do not leave the device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import bits_equal, calib_parts, np_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLACK = 40


def test_prefix_index_row_and_column_major(ctx):
    rng = np.random.default_rng(5)
    for h, w, p in [(1, 1, 1.0), (3, 5, 0.5), (37, 129, 0.3), (480, 640, 0.7), (1024, 1280, 0.05), (2, 4099, 0.9)]:
        flags = (rng.random((h, w)) < p).astype(np.uint8) * rng.integers(1, 255, (h, w), dtype=np.uint8)
        for cm in (False, True):
            f = flags.T if cm else flags                                   # enumeration order as a flat array
            rank = np.cumsum(f.reshape(-1) != 0) - (f.reshape(-1) != 0)
            exp = np.where(f.reshape(-1) != 0, rank + 7, 0xFFFFFFFF).astype(np.uint32).reshape(f.shape)
            exp = exp.T if cm else exp
            idx, tot = ctx.prefix_index(flags, column_major=cm, first=7)
            assert tot == int((flags != 0).sum()) and np.array_equal(idx, exp), (h, w, cm)
            didx, dtot = ctx.prefix_index(torch.from_numpy(flags).cuda(), column_major=cm, first=7)
            assert dtot == tot and np.array_equal(didx.cpu().numpy().view(np.uint32), exp)


def test_compact_points(ctx):
    rng = np.random.default_rng(6)
    for n, p in [(1, 1.0), (1000, 0.5), (300000, 0.2), (4097, 0.0)]:
        xyz = rng.standard_normal((n, 3)).astype(np.float32)
        has = (rng.random(n) < p).astype(np.uint8)
        pts, src = ctx.compact_points(xyz, has)
        sel = np.flatnonzero(has)
        assert np.array_equal(src.view(np.uint32), sel.astype(np.uint32)) and bits_equal(pts, xyz[sel])
        dp, ds = ctx.compact_points(torch.from_numpy(xyz).cuda(), torch.from_numpy(has).cuda())
        assert bits_equal(np_of(dp), xyz[sel]) and np.array_equal(np_of(ds).view(np.uint32), sel.astype(np.uint32))


@pytest.mark.parametrize("n_ctx,n_frames", [(2, 5), (3, 3), (1, 2), (3, 2)])
def test_multi_ctx_entry_against_the_oracle(slr, oracle, synth, n_ctx, n_frames):
    W, H = 320, 64
    calib, _ = synth.make_calibration(W, H, with_T=True)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    ctxs = [slr.Context(0) for _ in range(n_ctx)]
    try:
        for c in ctxs:
            c.set_calibration(calib)
            for cam in range(2):
                c.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
        frames = [synth.render_mf_stack(W, H, seed=100 + f, noise=2) for f in range(n_frames)]
        stacks = []
        for k in range(n_ctx):
            mine = [frames[f] for f in range(k, n_frames, n_ctx)]
            stacks.append(torch.stack(mine).cuda().contiguous() if mine else torch.empty((0, 2, 14, H, W), dtype=torch.uint8, device="cuda"))
        xa, ha = slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=n_ctx - 1)
        camL, camR, Q, T = calib_parts(oracle, calib)
        for f in range(n_frames):
            dec = []
            for cam in range(2):
                raw = frames[f][cam].numpy()
                rect = np.stack([oracle.remap_u8(raw[p], maps[cam][0].numpy(), maps[cam][1].numpy()) for p in range(14)])
                dec.append(oracle.mf_decode(rect, BLACK))
            exyz, ehas, _ = oracle.mf_triangulate(dec[0][0], dec[0][1], dec[1][0], dec[1][1], camL, camR, Q, T)
            assert bits_equal(np_of(ha[f]), ehas) and bits_equal(np_of(xa[f]), exyz), f
        # without assembly: the shards stay on their contexts
        xs, hs = slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=-1)
        for k in range(n_ctx):
            for j, f in enumerate(range(k, n_frames, n_ctx)):
                assert torch.equal(xs[k][j], xa[f]) and torch.equal(hs[k][j], ha[f])
    finally:
        for c in ctxs:
            c.close()


def _small_job(slr, synth, n_ctx, n_frames, devices=None):
    W, H = 320, 64
    calib, _ = synth.make_calibration(W, H, with_T=True)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    devices = devices or [0] * n_ctx
    ctxs = [slr.Context(d) for d in devices]
    for c in ctxs:
        c.set_calibration(calib)
        for cam in range(2):
            c.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    frames = [synth.render_mf_stack(W, H, seed=700 + f, noise=2) for f in range(n_frames)]
    stacks = []
    for k in range(n_ctx):
        mine = [frames[f] for f in range(k, n_frames, n_ctx)]
        dev = torch.device("cuda", devices[k])
        stacks.append(torch.stack(mine).to(dev).contiguous() if mine else torch.empty((0, 2, 14, H, W), dtype=torch.uint8, device=dev))
    return ctxs, stacks, frames


@pytest.mark.parametrize("n_ctx,n_frames", [(2, 5), (3, 4), (1, 2), (3, 2)])
def test_allgather_entry_every_context_gets_the_whole_cloud(slr, synth, n_ctx, n_frames):
    """slr_reconstruct_mf_allgather (north_star's exchange: every device ends with the assembled cloud): each context's copy ==
    the gather-to-one entry's result (itself compared with the oracle above), frame for frame, bit for bit"""
    ctxs, stacks, _ = _small_job(slr, synth, n_ctx, n_frames)
    try:
        xa, ha = slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=0)
        xs, hs, direct = slr.capi.reconstruct_mf_allgather(ctxs, stacks, BLACK, True, require_peer=True)
        assert direct == 1                                     # one device: every destination is directly addressable
        for k in range(n_ctx):
            assert torch.equal(hs[k], ha) and torch.equal(xs[k], xa), k
        assert ha.float().mean().item() > 0.2
        # the exchange proves itself on the devices: every context's checksums of its assembled cloud agree (slr_verify_assembled),
        # and the proof can fail -- one word changed in one context's copy
        assert slr.capi.verify_assembled(ctxs, xs, hs) == 0
        words = ctxs[0].cloud_checksums(xs[0], hs[0])
        assert len(set(int(w) for w in words)) == n_frames              # distinct frames, distinct words
        if n_ctx > 1:
            xs[n_ctx - 1][n_frames - 1, 5, 7, 1] += 1.0
            torch.cuda.synchronize()
            assert slr.capi.verify_assembled(ctxs, xs, hs) == 1
    finally:
        for c in ctxs:
            c.close()


def test_multi_entries_validate_every_context_before_any_work(slr, synth):
    """a context that must refuse the job (here: no rectification maps) makes the whole call fail BEFORE any other
    context is given work -- the caller's output buffers are untouched and may be freed at once"""
    ctxs, stacks, _ = _small_job(slr, synth, 3, 6)
    bare = slr.Context(0)                                      # calibration but no maps
    try:
        calib, _ = synth.make_calibration(320, 64, with_T=True)
        bare.set_calibration(calib)
        bad = [ctxs[0], ctxs[1], bare]
        n, H, W = 3, 64, 320
        xyz = [torch.full((2, H, W, 3), 7.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        has = [torch.full((2, H, W), 9, dtype=torch.uint8, device="cuda") for _ in range(n)]
        xall = torch.full((6, H, W, 3), 7.0, dtype=torch.float32, device="cuda")
        hall = torch.full((6, H, W), 9, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        import ctypes as C
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        arr_c = (C.c_void_p * n)(*[c.h.value for c in bad])
        lib = ctxs[0].lib
        st = lib.slr_reconstruct_mf_multi(arr_c, n, 6, arr(stacks), W, W, H, BLACK, 1, arr(xyz), arr(has), 0,
                                          C.c_void_p(xall.data_ptr()), C.c_void_p(hall.data_ptr()))
        assert st == slr.capi.ERR_NOT_CONFIGURED
        assert b"rectify maps" in lib.slr_last_error(ctxs[0].h)
        xa = [torch.full((6, H, W, 3), 7.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        ha = [torch.full((6, H, W), 9, dtype=torch.uint8, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        st = lib.slr_reconstruct_mf_allgather(arr_c, n, 6, arr(stacks), W, W, H, BLACK, 1, arr(xa), arr(ha), 0, None)
        assert st == slr.capi.ERR_NOT_CONFIGURED
        for c in ctxs:
            c.synchronize()
        for t in xyz + xa + [xall]:
            assert bool((t == 7.0).all())
        for t in has + ha + [hall]:
            assert bool((t == 9).all())
    finally:
        bare.close()
        for c in ctxs:
            c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two physical GPUs (the round's test boxes have one)")
def test_multi_entries_on_two_physical_devices(slr, synth):
    """the real multi-device path: one context per GPU, peer copies over xGMI; gather-to-one and all-gather == one context"""
    ctxs, stacks, frames = _small_job(slr, synth, 2, 5, devices=[0, 1])
    one, one_stacks, _ = _small_job(slr, synth, 1, 5)
    try:
        ex, eh = slr.capi.reconstruct_mf_multi(one, one_stacks, BLACK, True, gather_ctx=0)
        xa, ha = slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=1)
        assert xa.device.index == 1 and torch.equal(ha.cpu(), eh.cpu()) and torch.equal(xa.cpu(), ex.cpu())
        xs, hs, direct = slr.capi.reconstruct_mf_allgather(ctxs, stacks, BLACK, True)
        print("peer access between the two GPUs:", direct)
        for k in range(2):
            assert xs[k].device.index == k and torch.equal(hs[k].cpu(), eh.cpu()) and torch.equal(xs[k].cpu(), ex.cpu())
        assert slr.capi.verify_assembled(ctxs, xs, hs) == 0             # each GPU checksums its own copy, the words agree
    finally:
        for c in ctxs + one:
            c.close()


def test_multi_ctx_entry_fullsize_equals_the_batch_entry(slr, synth):
    """config 4's shape on what the box has: 4 frames of 4096x3000 over two contexts == slr_reconstruct_mf_batch on one"""
    W, H = 4096, 3000
    dev = torch.device("cuda", 0)
    calib, _ = synth.make_calibration(W, H)
    maps = [synth.make_rectify_maps(W, H, cam, device=dev) for cam in range(2)]
    frames = torch.stack([synth.render_mf_stack(W, H, seed=40 + f, noise=2, device=dev) for f in range(4)])
    torch.cuda.synchronize()
    ctxs = [slr.Context(0) for _ in range(2)]
    try:
        for c in ctxs:
            c.set_calibration(calib)
            for cam in range(2):
                c.set_rectify_maps(cam, maps[cam][0], maps[cam][1])
        ex, eh = ctxs[0].reconstruct_mf_batch(frames, BLACK, True)
        ctxs[0].synchronize()
        stacks = [frames[0::2].contiguous(), frames[1::2].contiguous()]
        xa, ha = slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=0)
        assert torch.equal(ha, eh) and torch.equal(xa, ex)
        assert 0.2 < eh.float().mean().item() < 1.0
        # sparse assembly: only the valid points of a frame
        pts, src = ctxs[1].compact_points(xa[3], ha[3])
        sel = torch.nonzero(ha[3].reshape(-1)).reshape(-1)
        assert torch.equal(src.to(torch.int64), sel) and torch.equal(pts, xa[3].reshape(-1, 3)[sel])
    finally:
        for c in ctxs:
            c.close()


def test_bench_two_ranks_with_a_collective_on_this_box():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per "GPU", frames sharded, one all-gather of
    the final cloud) with both ranks on device 0.  RCCL first; the ONLY accepted reason to run the same code path over gloo
    instead is RCCL refusing two ranks on one device (on a box with two GPUs that cannot happen).  The bench line records which
    backend ran and how many ranks it had; both are asserted, and which one ran is printed."""
    env = dict(os.environ, SLR_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    refused = None
    for backend, port in (("nccl", "29541"), ("gloo", "29542")):
        env["SLR_BENCH_BACKEND"] = backend
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2",
               "--width", "1024", "--height", "512", "--gather", "final", "--traffic", "off", "--cpu-baseline", "0"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
            assert d["config"]["distinct_frames"] == 2 and d["config"]["frames_per_gpu_per_step"] == 2 * d["config"]["passes_per_step"] and "all-gather" in d["config"]["parallelism"]
            assert d["collective_backend"] == backend and d["collective_ranks"] == 2, d.get("collective")
            assert len(d["collective"]["devices"]) == 2 and d["collective"]["distinct_devices"] == 1   # both ranks on the one GPU
            # the exchange proves itself: every rank compared all 4 frames of its assembled cloud with their owners' checksums
            assert d["gather_proof"]["all_ranks_ok"] and d["gather_proof"]["frames_verified_on_every_rank"] == 4, d["gather_proof"]
            assert d["gather_inclusive_value"] is not None
            print("collective leg ran over %s (RCCL refused: %r)" % (backend, refused))
            return
        if backend == "nccl":
            # RCCL may only be given up for its "two ranks on one device" refusal
            text = r.stderr + r.stdout
            dup = any(k in text for k in ("Duplicate GPU", "duplicate GPU", "ncclInvalidUsage", "invalid usage"))
            assert dup, "RCCL failed for a reason other than two ranks on one device:\n" + text[-1500:]
            refused = [ln for ln in text.splitlines() if "uplicate GPU" in ln or "nvalid" in ln][:2]
            continue
        pytest.fail("bench.py --gpus 2 failed over gloo: rc %d\n%s" % (r.returncode, r.stderr[-1500:]))


def test_bench_config5_two_row_bands_with_a_collective_on_this_box():
    """bench.py --mode mfn as the driver launches it for N = 2 (BASELINE config 5's multi-GPU leg: ONE frame split into row bands,
    every rank holding only the source rows its band's maps point into, one all-gather of the bands), both ranks on device 0, at a
    small size.  RCCL first; gloo only if RCCL refuses two ranks on one device.  The ranks' assembled clouds must agree."""
    env = dict(os.environ, SLR_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for backend, port in (("nccl", "29543"), ("gloo", "29544")):
        env["SLR_BENCH_BACKEND"] = backend
        # (round 6: started BARE -- bench.py spawns its own two ranks under torch.distributed.run, --mode mfn included)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "mfn", "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--frames", "1", "--width", "2048", "--height", "500"]
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong" and d["config"]["mode"] == "mfn"
            assert d["config"]["rows_of_this_rank"] == [0, 250] and "row bands" in d["config"]["parallelism"]
            assert d["gather_proof"] == {"frames": 1, "all_ranks_agree": True} and d["final_allgather_ms"] > 0
            print("config 5 row-band leg ran over %s" % backend)
            return
        if backend == "nccl":
            text = r.stderr + r.stdout
            assert any(k in text for k in ("Duplicate GPU", "duplicate GPU", "ncclInvalidUsage", "invalid usage")), \
                "RCCL failed for a reason other than two ranks on one device:\n" + text[-1500:]
            continue
        pytest.fail("bench.py --mode mfn --gpus 2 failed over gloo: rc %d\n%s" % (r.returncode, r.stderr[-1500:]))


# ---- round 6: the blocked assignment, the exchange on its own, bench.py as its own launcher, --impl one-process -----------------
def _blocked_job(slr, synth, n_ctx, n_frames, W=320, H=64):
    calib, _ = synth.make_calibration(W, H, with_T=True)
    maps = [synth.make_rectify_maps(W, H, cam) for cam in range(2)]
    ctxs = [slr.Context(0) for _ in range(n_ctx)]
    for c in ctxs:
        c.set_calibration(calib)
        for cam in range(2):
            c.set_rectify_maps(cam, maps[cam][0].numpy(), maps[cam][1].numpy())
    frames = torch.stack([synth.render_mf_stack(W, H, seed=900 + f, noise=2) for f in range(n_frames)]).cuda()
    S = (n_frames + n_ctx - 1) // n_ctx
    stacks = [frames[min(n_frames, k * S):min(n_frames, (k + 1) * S)].contiguous() for k in range(n_ctx)]
    return ctxs, stacks, frames


@pytest.mark.parametrize("n_ctx,n_frames,group", [(3, 7, 0), (2, 5, 2), (3, 2, 0), (1, 3, 0), (4, 21, 3)])
def test_allgather_ex_blocked_equals_the_batch_entry(slr, synth, n_ctx, n_frames, group):
    """slr_reconstruct_mf_allgather_ex(SLR_ASSIGN_BLOCKED): context k owns frames [k S, (k + 1) S) (ragged and empty shards included),
    computed in groups in place and pushed per group on per-destination streams == slr_reconstruct_mf_batch of all frames on one
    context; the cyclic assignment through the same entry == the old entry; the exchange alone (slr_allgather_clouds) after
    slr_reconstruct_mf_multi wrote the shards in place gives the same arrays"""
    ctxs, stacks, frames = _blocked_job(slr, synth, n_ctx, n_frames)
    try:
        if group:
            for c in ctxs:
                c.set_option(slr.capi.OPT_MF_BATCH_GROUP, group)
        ex, eh = ctxs[0].reconstruct_mf_batch(frames, BLACK, True)
        ctxs[0].synchronize()
        xs, hs, direct = slr.capi.reconstruct_mf_allgather(ctxs, stacks, BLACK, True, require_peer=True, assignment=slr.capi.ASSIGN_BLOCKED)
        assert direct == 1
        for k in range(n_ctx):
            assert torch.equal(hs[k], eh) and torch.equal(xs[k], ex), k
        assert slr.capi.verify_assembled(ctxs, xs, hs) == 0
        # the exchange on its own: poisoned assembled arrays, every context's shard written in place, then pushed
        S = (n_frames + n_ctx - 1) // n_ctx
        if n_frames == n_ctx * S:                           # (slr_reconstruct_mf_multi's shares are cyclic counts: equal shards only)
            xa = [torch.full((n_frames, 64, 320, 3), 7.0, dtype=torch.float32, device="cuda") for _ in range(n_ctx)]
            ha = [torch.full((n_frames, 64, 320), 9, dtype=torch.uint8, device="cuda") for _ in range(n_ctx)]
            slr.capi.reconstruct_mf_multi(ctxs, stacks, BLACK, True, gather_ctx=-1, xyz=[xa[k][k * S:(k + 1) * S] for k in range(n_ctx)],
                                          has=[ha[k][k * S:(k + 1) * S] for k in range(n_ctx)])
            for k in range(n_ctx):                           # nothing but the own shard is written before the exchange
                others = [f for f in range(n_frames) if not k * S <= f < (k + 1) * S]
                assert all(bool((ha[k][f] == 9).all()) for f in others)
            assert slr.capi.allgather_clouds(ctxs, xa, ha, assignment=slr.capi.ASSIGN_BLOCKED, require_peer=True) == 1
            for k in range(n_ctx):
                assert torch.equal(ha[k], eh) and torch.equal(xa[k], ex), k
        # cyclic through the _ex entry == the round-2 entry
        cyc = [frames[k::n_ctx].contiguous() for k in range(n_ctx)]
        xc, hc, _ = slr.capi.reconstruct_mf_allgather(ctxs, cyc, BLACK, True, assignment=slr.capi.ASSIGN_CYCLIC)
        xo, ho, _ = slr.capi.reconstruct_mf_allgather(ctxs, cyc, BLACK, True)
        for k in range(n_ctx):
            assert torch.equal(hc[k], eh) and torch.equal(xc[k], ex) and torch.equal(ho[k], eh) and torch.equal(xo[k], ex)
    finally:
        for c in ctxs:
            c.close()


def test_allgather_ex_refuses_a_bad_assignment(slr, synth):
    ctxs, stacks, _ = _blocked_job(slr, synth, 2, 4)
    try:
        with pytest.raises(slr.capi.SlrError) as e:
            slr.capi.reconstruct_mf_allgather(ctxs, stacks, BLACK, True, assignment=2)
        assert e.value.status == slr.capi.ERR_INVALID_ARG and "assignment" in str(e.value)
    finally:
        for c in ctxs:
            c.close()


def _bench_line(cmd, env, timeout=900):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    return r, (json.loads(lines[-1]) if r.returncode == 0 and lines else None)


def test_bench_started_bare_spawns_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher (the way the driver starts --gpus 1): the script re-executes itself under
    torch.distributed.run, one rank per "GPU" (both on device 0 here), and prints the one line.  RCCL first; gloo only on RCCL's
    two-ranks-on-one-device refusal."""
    env = dict(os.environ, SLR_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for backend in ("nccl", "gloo"):
        env["SLR_BENCH_BACKEND"] = backend
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2", "--width", "1024",
               "--height", "512", "--traffic", "off", "--cpu-baseline", "0"]
        r, d = _bench_line(cmd, env)
        if d is not None:
            assert d["n_gpus"] == 2 and d["collective_backend"] == backend and d["collective_ranks"] == 2
            assert d["gather_proof"]["all_ranks_ok"] and d["gather_proof"]["frames_verified_on_every_rank"] == 4
            assert d["final_allgather_ms"] > 0 and d["hbm_footprint_per_gpu"]["sum_of_parts"] > 0
            assert d["realistic_maps"] == [] and d["cpu_baseline"] is None and d["host_buffers_pcie_inclusive"] is None   # N > 1: no extras
            print("bare --gpus 2 ran over %s" % backend)
            return
        if backend == "nccl":
            text = r.stderr + r.stdout
            assert any(k in text for k in ("Duplicate GPU", "duplicate GPU", "ncclInvalidUsage", "invalid usage")), \
                "RCCL failed for a reason other than two ranks on one device:\n" + text[-1500:]
            continue
        pytest.fail("bare bench.py --gpus 2 failed over gloo: rc %d\n%s" % (r.returncode, r.stderr[-1500:]))


def test_bench_bare_refuses_more_ranks_than_gpus_without_the_dry_run_switch():
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU box runs this for real")
    env = {k: v for k, v in os.environ.items() if k not in ("SLR_BENCH_ONE_DEVICE", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "this box has" in r.stderr and '{"metric"' not in r.stdout


@pytest.mark.parametrize("gather", ["after", "final"])
def test_bench_one_process_three_contexts(gather):
    """--impl one-process: what a C++ host calls -- N contexts from one process, slr_reconstruct_mf_multi in place + ONE
    slr_allgather_clouds (or slr_reconstruct_mf_allgather_ex inside the timed region), the same line with collective_backend "peer" """
    env = dict(os.environ, SLR_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--impl", "one-process", "--gpus", "3", "--steps", "2", "--warmup", "1", "--frames", "3",
           "--width", "1024", "--height", "512", "--gather", gather, "--passes", "2"]
    r, d = _bench_line(cmd, env)
    assert d is not None, "rc %d\n%s" % (r.returncode, r.stderr[-2000:])
    assert d["n_gpus"] == 3 and d["collective_backend"] == "peer" and d["collective_ranks"] == 3 and d["scaling"] == "weak"
    assert d["config"]["impl"] == "one-process" and d["config"]["distinct_frames"] == 3 and d["value"] > 0
    assert d["gather_proof"]["all_ranks_ok"] and d["gather_proof"]["frames_verified_on_every_rank"] == 9
    assert d["gather_proof"]["slr_verify_assembled_mismatches"] == 0
    assert ((d["final_allgather_ms"] or 0) > 0) == (gather == "after") and d["gather_inclusive_value"] is not None
    assert d["roofline"]["kernel"] in ("slr_mf_rectify_decode_pair", "slr_mf_match_triangulate")
    assert d["hbm_footprint_per_gpu"]["assembled_xyz_and_mask"] == 9 * 1024 * 512 * 13

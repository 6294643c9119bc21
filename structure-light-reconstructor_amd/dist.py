"""Multi-GPU driver: stereo frames shard across the GPUs of one node, one process per GPU.

The reference has nothing distributed; frames are independent, so the only exchange step is the
assembly of the final point cloud: one RCCL all-gather (torch.distributed backend "nccl" IS RCCL on ROCm) of the
per-frame XYZ + mask.  No collective happens inside the decode / match kernels.  On CPU test rigs the same code
runs over gloo (tests/test_dist_gloo.py, world_size 2), with the reconstruction step injected.
"""
import inspect

import torch
import torch.distributed as dist


def frames_per_rank(n_frames, world):
    """max shard length; shards are padded to it so the all-gather has equal counts on every rank"""
    return (n_frames + world - 1) // world


def shard_frames(n_frames, rank, world, assignment="cyclic"):
    """This rank's global frame indices, ascending.  "cyclic": frame f -> rank f % world; "blocked": rank r owns
    frames [r*S, (r+1)*S), S = frames_per_rank -- its shard is then ONE contiguous piece of the assembled cloud, so a batch entry
    point can write all of it in one call and a single in-place all-gather assembles everything (what bench.py does)."""
    if assignment == "blocked":
        S = frames_per_rank(n_frames, world)
        return list(range(min(n_frames, rank * S), min(n_frames, (rank + 1) * S)))
    assert assignment == "cyclic"
    return list(range(rank, n_frames, world))


def local_slots(assembled, rank, world, assignment="cyclic"):
    """This rank's shard slots as a VIEW of an assembled array [S*world][...]: writing a frame there puts it where the gather
    wants it (slot s = frame s*world + rank, or frame rank*S + s)."""
    S = assembled.shape[0] // world
    return assembled[rank::world] if assignment == "cyclic" else assembled[rank * S:(rank + 1) * S]


def _alias(out, src, rank):
    """How `src` lies relative to `out` for an all-gather whose chunk `rank` is src-sized: "inplace" -- src IS chunk `rank` of out
    (RCCL / NCCL's in-place rule: sendbuff == recvbuff + rank * count), "disjoint" -- no byte in common, "overlap" -- anything else
    (never valid as an aliased input: the caller must hand a copy)."""
    sz = src.numel() * src.element_size()
    o0, s0 = out.data_ptr(), src.data_ptr()
    if s0 + sz <= o0 or s0 >= o0 + out.numel() * out.element_size():
        return "disjoint"
    return "inplace" if (src.is_contiguous() and out.is_contiguous() and s0 == o0 + rank * sz) else "overlap"


def _all_gather(out, src, group):
    """dist.all_gather_into_tensor, in place on RCCL when `src` is this rank's own chunk of `out` (checked: an aliased input that
    is NOT at recvbuff + rank * count would corrupt the result or hang, and only on hardware with two or more GPUs) -- any other
    aliasing gets a private copy; a gloo group (the CPU tests, the dry run of the N > 1 path on a one-GPU box) gets host tensors
    and a private copy of an aliased input."""
    kind = _alias(out, src, dist.get_rank(group))
    if kind == "overlap":
        src = src.clone()
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, src, group=group)
    elif out.is_cuda:
        torch.cuda.current_stream(out.device).synchronize()
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, src.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, src.clone() if kind == "inplace" else src, group=group)


def gather_point_clouds(xyz_local, has_local, n_frames, group=None, out=None, assignment="cyclic"):
    """All-gather the per-rank shards; returns the clouds in global frame order on every rank.

    xyz_local [S][H][W][3] f32, has_local [S][H][W] u8, with S == frames_per_rank(n_frames, world); shard slot s holds global
    frame s*world + rank ("cyclic") or rank*S + s ("blocked"); slots past the end of a short shard are padding and are dropped.
    "cyclic": one all-gather per shard slot and array -- slot s of every rank lands in frames [s*world, (s+1)*world) of the
    output, which IS global frame order; "blocked": one all-gather per array (rank-major order is frame order).  Either way no
    transpose or re-ordering copy of the assembled cloud (10 GB at config 4) happens, and a collective moves >= world x 160 MB at
    4096x3000.  `out` = (xyz [S*world][H][W][3], has [S*world][H][W]) to gather into caller-owned buffers; shard slots that
    already live in their place there (local_slots(out[0], ...), see reconstruct_sharded) are gathered in place.
    Returns (xyz [n_frames][H][W][3], has [n_frames][H][W]): views of the assembled arrays.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = frames_per_rank(n_frames, world)
    assert xyz_local.shape[0] == S and has_local.shape[0] == S
    if world == 1:
        return xyz_local[:n_frames], has_local[:n_frames]
    if out is None:
        out = (torch.empty((world * S,) + tuple(xyz_local.shape[1:]), dtype=xyz_local.dtype, device=xyz_local.device),
               torch.empty((world * S,) + tuple(has_local.shape[1:]), dtype=has_local.dtype, device=has_local.device))
    g_xyz, g_has = out
    assert g_xyz.shape[0] == world * S and g_has.shape[0] == world * S and g_xyz.is_contiguous() and g_has.is_contiguous()
    for g, loc in ((g_xyz, xyz_local), (g_has, has_local)):
        if assignment == "blocked":
            _all_gather(g, loc if loc.is_contiguous() else loc.contiguous(), group)
        else:
            for s in range(S):
                _all_gather(g[s * world:(s + 1) * world], loc[s:s + 1], group)      # ([s:s+1] of a strided view is contiguous)
    return g_xyz[:n_frames], g_has[:n_frames]


def frame_checksums(xyz, has):
    """One 64-bit word per frame of xyz [n][H][W][3] f32 / has [n][H][W] u8: sums of the raw bits weighted per ELEMENT -- by row
    and by position inside the row (wrapping int64 arithmetic) -- so that a shifted or permuted row, swapped channels or pixels
    and a misaligned chunk all change the word, not only a changed row sum.  Computed on the arrays' device.  Used to prove an
    assembled cloud: every rank checksums its LOCAL results, the words travel by a second (tiny) all-gather, and each gathered
    frame must reproduce its owner's word.  (The library's own proof, slr_cloud_checksums, weights per pixel and channel too; the
    two are different functions and are never compared with each other.)"""
    n, H = xyz.shape[0], xyz.shape[1]
    w = torch.arange(1, H + 1, dtype=torch.int64, device=xyz.device)
    out = torch.empty(n, dtype=torch.int64, device=xyz.device)
    wx = wh = None
    R = 256                                                  # rows per block: the int64 temporaries stay at 8 x a block, not 8 x a frame
    for f in range(n):
        bx = xyz[f].contiguous().view(torch.int32).view(H, -1)
        bh = has[f].contiguous().view(H, -1)
        if wx is None:
            wx = torch.arange(bx.shape[1], dtype=torch.int64, device=xyz.device) % 65521 + 1
            wh = torch.arange(bh.shape[1], dtype=torch.int64, device=xyz.device) % 65521 + 1
        sx = torch.zeros((), dtype=torch.int64, device=xyz.device)
        sh = torch.zeros((), dtype=torch.int64, device=xyz.device)
        for r0 in range(0, H, R):
            sx += ((bx[r0:r0 + R].to(torch.int64) * wx).sum(dim=1) * w[r0:r0 + R]).sum()
            sh += ((bh[r0:r0 + R].to(torch.int64) * wh).sum(dim=1) * w[r0:r0 + R]).sum()
        out[f] = sx * 1000003 + sh
    return out


def verify_gathered(xyz_all, has_all, mine, n_frames, group=None, assignment="cyclic"):
    """mine = frame_checksums of this rank's LOCAL shard slots, taken BEFORE the gather ([S] int64).  Every rank all-gathers
    these words and compares each frame of the assembled cloud with its owner's.  Raises RuntimeError on the first mismatch (on
    every rank that sees one); returns the number of frames checked."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = frames_per_rank(n_frames, world)
    assert mine.shape == (S,) and mine.dtype == torch.int64
    if world > 1:
        allw = torch.empty(world * S, dtype=torch.int64, device=mine.device)
        _all_gather(allw, mine, group)                                                   # [rank][slot]
        owner = (allw.view(world, S).t().reshape(-1) if assignment == "cyclic" else allw)[:n_frames]
    else:
        owner = mine[:n_frames]
    got = frame_checksums(xyz_all[:n_frames], has_all[:n_frames])
    bad = torch.nonzero(got != owner).reshape(-1)
    if bad.numel():
        f = int(bad[0])
        own = (f % world, f // world) if assignment == "cyclic" else (f // S, f % S)
        raise RuntimeError("assembled point cloud: frame %d (owner rank %d, slot %d) does not match its owner's checksum "
                           "(%d of %d frames differ)" % (f, own[0], own[1], bad.numel(), n_frames))
    return int(n_frames)


def reconstruct_sharded(n_frames, H, W, load_frame, reconstruct, device, group=None, verify=False, assignment="cyclic"):
    """Whole multi-GPU job: every rank reconstructs its shard, then the all-gathers assemble the result.

    load_frame(f) -> whatever `reconstruct` consumes for global frame f (e.g. a [2][14][H][pitch] u8 stack in HBM)
    reconstruct(frame, xyz_out [H][W][3] f32, has_out [H][W] u8) writes global frame f's cloud into the two views it is
    handed (the HIP path: Context.reconstruct_mf*(..., xyz=xyz_out, has=has_out)) -- they are the frame's own place in the
    assembled arrays, so nothing is copied between the kernels and the collective.
    verify=True: prove the assembled cloud with per-frame checksums (verify_gathered).
    """
    try:                                                     # the callback's arity, checked up front (round 4 changed the signature)
        inspect.signature(reconstruct).bind(None, None, None)
    except TypeError as e:
        raise TypeError("reconstruct_sharded: `reconstruct` must take (frame, xyz_out, has_out) and write the frame's cloud into "
                        "the two views (not the old reconstruct(frame) -> (xyz, has))") from e
    except ValueError:                                       # a callable without an introspectable signature: let the call speak
        pass
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = frames_per_rank(n_frames, world)
    g_xyz = torch.zeros((S * world, H, W, 3), dtype=torch.float32, device=device)
    g_has = torch.zeros((S * world, H, W), dtype=torch.uint8, device=device)
    loc_xyz, loc_has = local_slots(g_xyz, rank, world, assignment), local_slots(g_has, rank, world, assignment)
    for s, f in enumerate(shard_frames(n_frames, rank, world, assignment)):
        reconstruct(load_frame(f), loc_xyz[s], loc_has[s])
    mine = frame_checksums(loc_xyz, loc_has) if verify else None              # before the (in-place) gather touches anything
    xyz, has = gather_point_clouds(loc_xyz, loc_has, n_frames, group, out=(g_xyz, g_has), assignment=assignment)
    if verify:
        verify_gathered(xyz, has, mine, n_frames, group, assignment)
    return xyz, has


# ---- one huge frame (BASELINE config 5): shard by ROW BANDS ------------------------------------------------
def shard_rows(H, rank, world):
    """rows [r0, r1) of rank `rank`: `world` bands of ceil(H / world) rows (the last ones may be short or empty).
    Every kernel of the unrectified path is row-local (decode is per pixel, the match runs along a row), so a band
    needs no halo and no collective."""
    band = (H + world - 1) // world
    r0 = min(H, rank * band)
    return r0, min(H, r0 + band)


def reconstruct_row_sharded(H, W, reconstruct_rows, device, group=None):
    """reconstruct_rows(r0, r1) -> (xyz [r1-r0][W][3] f32, has [r1-r0][W] u8) for this rank's band, e.g. the HIP path on
    row views of the planes.  One all-gather of the (padded, equal-sized) bands assembles the frame on every rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    band = (H + world - 1) // world
    r0, r1 = shard_rows(H, rank, world)
    xyz = torch.zeros((band, W, 3), dtype=torch.float32, device=device)
    has = torch.zeros((band, W), dtype=torch.uint8, device=device)
    if r1 > r0:
        x, h = reconstruct_rows(r0, r1)
        xyz[:r1 - r0].copy_(x)
        has[:r1 - r0].copy_(h)
    if world == 1:
        return xyz[:H], has[:H]
    g_xyz = torch.empty((world * band, W, 3), dtype=torch.float32, device=device)
    g_has = torch.empty((world * band, W), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(g_xyz, xyz, group=group)
    dist.all_gather_into_tensor(g_has, has, group=group)
    return g_xyz[:H].contiguous(), g_has[:H].contiguous()          # bands are consecutive: padding only at the end

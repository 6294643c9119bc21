"""Multi-GPU driver: stereo frames shard across the GPUs of one node, one process per GPU.

The reference has nothing distributed; frames are independent, so the only exchange step is the
assembly of the final point cloud: one RCCL all-gather (torch.distributed backend "nccl" IS RCCL on ROCm) of the
per-frame XYZ + mask.  No collective happens inside the decode / match kernels.  On CPU test rigs the same code
runs over gloo (tests/test_dist_gloo.py, world_size 2), with the reconstruction step injected.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """frame f -> rank f % world.  Returns this rank's global frame indices, ascending."""
    return list(range(rank, n_frames, world))


def frames_per_rank(n_frames, world):
    """max shard length; shards are padded to it so the all-gather has equal counts on every rank"""
    return (n_frames + world - 1) // world


def gather_point_clouds(xyz_local, has_local, n_frames, group=None):
    """All-gather the per-rank shards and return the clouds in global frame order.

    xyz_local [S][H][W][3] f32, has_local [S][H][W] u8, with S == frames_per_rank(n_frames, world); shard slot s
    holds global frame rank + s*world (slots past the end of a short shard are padding and are dropped).
    Returns (xyz [n_frames][H][W][3], has [n_frames][H][W]) on every rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = frames_per_rank(n_frames, world)
    assert xyz_local.shape[0] == S and has_local.shape[0] == S
    if world == 1:
        return xyz_local[:n_frames], has_local[:n_frames]
    # output = concatenation of the shards along dim 0 (the layout both RCCL and gloo accept)
    g_xyz = torch.empty((world * S,) + tuple(xyz_local.shape[1:]), dtype=xyz_local.dtype, device=xyz_local.device)
    g_has = torch.empty((world * S,) + tuple(has_local.shape[1:]), dtype=has_local.dtype, device=has_local.device)
    dist.all_gather_into_tensor(g_xyz, xyz_local.contiguous(), group=group)
    dist.all_gather_into_tensor(g_has, has_local.contiguous(), group=group)
    # [rank][slot] -> global frame slot*world + rank
    g_xyz = g_xyz.view((world, S) + tuple(xyz_local.shape[1:]))
    g_has = g_has.view((world, S) + tuple(has_local.shape[1:]))
    xyz = g_xyz.transpose(0, 1).reshape((S * world,) + tuple(xyz_local.shape[1:]))[:n_frames]
    has = g_has.transpose(0, 1).reshape((S * world,) + tuple(has_local.shape[1:]))[:n_frames]
    return xyz.contiguous(), has.contiguous()


def reconstruct_sharded(n_frames, H, W, load_frame, reconstruct, device, group=None):
    """Whole multi-GPU job: every rank reconstructs its shard, then one all-gather assembles the result.

    load_frame(f) -> whatever `reconstruct` consumes for global frame f (e.g. a [2][14][H][pitch] u8 stack in HBM)
    reconstruct(frame) -> (xyz [H][W][3] f32, has [H][W] u8) on `device` (the HIP path: Context.reconstruct_mf*)
    """
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    S = frames_per_rank(n_frames, world)
    xyz = torch.zeros((S, H, W, 3), dtype=torch.float32, device=device)
    has = torch.zeros((S, H, W), dtype=torch.uint8, device=device)
    for s, f in enumerate(shard_frames(n_frames, rank, world)):
        x, h = reconstruct(load_frame(f))
        xyz[s].copy_(x)
        has[s].copy_(h)
    return gather_point_clouds(xyz, has, n_frames, group)


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

"""Instance sharding across the GPUs of one node.  One process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  Instances are independent through encode / SDF decode and
registration pairs are independent after matching, so the data path has NO collective: the only exchanges are
  * a one-off broadcast of the packed weights (29.6 MB fp32) from rank 0, and
  * a gather / all-gather of per-instance codes (z_so3 768 + z_inv 256 + s 1 + t 3 = 1028 floats = 4.1 KB) to the rank
    that runs a scene's 32x32 matcher, and of the (R, t) results (12 floats per pair).
All messages are KB..MB sized, i.e. latency-bound: flat broadcast / direct gather, never a ring of small buckets."""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _host_staged():
    """gloo moves all_gather / gather payloads through host memory only (the CPU tests, and the single-GPU dry run of the N > 1
    path): stage device tensors on the host for the collective and bring the result back.  RCCL ("nccl") takes device tensors."""
    return dist.get_backend() == "gloo"


def shard_range(n_items, rank=None, world_size=None):
    """Contiguous block partition of n_items: rank r gets [lo, hi).  Sizes differ by at most one."""
    if rank is None or world_size is None:
        rank, world_size = world()
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def balanced_assignment(costs, world_size=None):
    """Cost-balanced partition of items with (integer) costs over the ranks: -> list of index lists, one per rank, each ascending.
    Equal costs keep the contiguous block partition of shard_range (nothing to balance: identical to the unbalanced path).  Otherwise longest-
    processing-time-first: items by decreasing cost (ties: by index) onto the currently lightest rank (ties: lowest rank) -- a deterministic
    function of `costs`, so every rank computes the same assignment without talking; max load <= 4/3 of the optimum, and within a few per cent of
    the mean once a rank holds several items.  Motivation (SURVEY 8e, configs[3]): a 3RScan instance has 1 k - 60 k raw points and the ragged FPS
    costs ~P per cloud, so equal instance COUNTS leave ranks with very unequal work."""
    if world_size is None:
        world_size = world()[1]
    n = len(costs)
    costs = [int(c) for c in costs]
    if n == 0 or min(costs) == max(costs):
        return [list(range(*shard_range(n, r, world_size))) for r in range(world_size)]
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in sorted(range(n), key=lambda i: (-costs[i], i)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        load[r] += costs[i]
        out[r].append(i)
    return [sorted(o) for o in out]


def assignment_restore(assign, device=None):
    """Index tensor that puts rows concatenated in rank order (rank 0's items, rank 1's, ...) back into the original item order."""
    flat = [i for a in assign for i in a]
    inv = torch.empty(len(flat), dtype=torch.long)
    inv[torch.tensor(flat, dtype=torch.long)] = torch.arange(len(flat))
    return inv.to(device) if device is not None else inv


ENCODE_COST_POINTS = 1024   # what one instance costs beyond its FPS pass, in points: the encoder works on 1024 sampled points whatever the cloud's size


def broadcast_weights(module, src=0):
    """Broadcast every parameter of `module` from rank `src` as ONE flat fp32 buffer (one collective, not 78)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    params = [p for p in module.parameters()]
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


CODE_KEYS = ("z_so3", "z_inv", "s", "t")


def pack_codes(emb):
    """{z_so3 [B,C,3], z_inv [B,C], s [B], t [B,1,3]} -> [B, 4C+4] rows (one message per shard)."""
    B, C = emb["z_inv"].shape      # explicit widths: reshape(B, -1) is ambiguous for an empty shard (B = 0)
    return torch.cat([emb["z_so3"].reshape(B, 3 * C), emb["z_inv"].reshape(B, C), emb["s"].reshape(B, 1), emb["t"].reshape(B, 3)], 1).contiguous()


def empty_codes(c_dim, device, dtype=torch.float32):
    """The codes of an EMPTY shard (a rank that owns no instance: fewer instances than ranks, common for a small 3RScan
    scene on an 8-GPU node): zero-row tensors of the right widths, so the rank still takes part in the collectives."""
    return unpack_codes(torch.zeros(0, 4 * c_dim + 4, device=device, dtype=dtype))


def unpack_codes(rows):
    B, W = rows.shape
    C = (W - 4) // 4
    return {"z_so3": rows[:, :3 * C].reshape(B, C, 3), "z_inv": rows[:, 3 * C:4 * C], "s": rows[:, 4 * C], "t": rows[:, 4 * C + 1:].reshape(B, 1, 3)}


def all_gather_codes(emb, counts=None):
    """All ranks receive the codes of all shards, concatenated in rank order.  `counts` = rows per rank (defaults
    to equal shards); ragged shards are padded to the largest."""
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return emb
    rows = pack_codes(emb)
    if counts is None:
        counts = [rows.shape[0]] * ws
    mx = max(max(counts), 1)     # at least one (zero) row per message: an all-empty exchange stays a valid collective
    pad = rows.new_zeros(mx, rows.shape[1])
    pad[: rows.shape[0]] = rows
    dev = pad.device
    if _host_staged():
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return unpack_codes(torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(dev))


def all_gather_rows(rows, counts):
    """All ranks receive the float rows [n_r, W] of all ranks, concatenated in rank order (counts[r] = n_r; ragged and EMPTY shards are
    padded to the largest, so every rank always takes part).  The generic form of all_gather_codes: also carries the (R | t) rows of the
    registration results (12 floats = 48 B per pair)."""
    rank, ws = world()
    if ws == 1:
        return rows
    mx = max(max(counts), 1)
    pad = rows.new_zeros(mx, rows.shape[1])
    pad[: rows.shape[0]] = rows
    dev = pad.device
    if _host_staged():
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(dev)


def shard_counts(n_items, world_size=None):
    if world_size is None:
        world_size = world()[1]
    return [shard_range(n_items, r, world_size)[1] - shard_range(n_items, r, world_size)[0] for r in range(world_size)]


def gather_codes(emb, dst=0, counts=None):
    """Rank `dst` receives all codes (rank order); other ranks get None."""
    rank, ws = world()
    if ws == 1:
        return emb
    rows = pack_codes(emb)
    if counts is None:
        counts = [rows.shape[0]] * ws
    mx = max(max(counts), 1)
    pad = rows.new_zeros(mx, rows.shape[1])
    pad[: rows.shape[0]] = rows
    dev = pad.device
    if _host_staged():
        pad = pad.cpu()
    bufs = [torch.empty_like(pad) for _ in range(ws)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return unpack_codes(torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(dev))


def sharded_encode(model, x_all):
    """Encode a list/batch of instances x_all [n,3,N] (same tensor on every rank) with each rank encoding its block,
    then all-gather the codes so that every rank can run the (deterministic, tiny) matcher."""
    rank, ws = world()
    n = x_all.shape[0]
    lo, hi = shard_range(n, rank, ws)
    if hi > lo:
        emb = model.encode(x_all[lo:hi].contiguous())
    else:   # this rank owns nothing (n < world size): ls_encode rejects B = 0, and skipping the all-gather would deadlock the others
        emb = empty_codes(model.encoder.c_dim, x_all.device)
    counts = [shard_range(n, r, ws)[1] - shard_range(n, r, ws)[0] for r in range(ws)]
    return all_gather_codes(emb, counts)


def sharded_encode_fps(model, clouds, balance=True):
    """Shape_Prior.encode_fps over a LIST of raw clouds [Ni,3] (the flat (scene, instance) list of SURVEY 8e): every rank samples and
    encodes its share (one ragged FPS launch + one encoder batch per rank), the codes are all-gathered and returned in list order.  Same
    list on every rank.  balance=True: the share is cost-balanced (balanced_assignment on P_i + ENCODE_COST_POINTS: the ragged FPS costs ~P per
    cloud) instead of equal counts; equal-size clouds keep the block partition."""
    rank, ws = world()
    n = len(clouds)
    dev = clouds[0].device
    if balance:
        assign = balanced_assignment([c.shape[0] + ENCODE_COST_POINTS for c in clouds], ws)
    else:
        assign = [list(range(*shard_range(n, r, ws))) for r in range(ws)]
    mine = [clouds[i] for i in assign[rank]]
    if mine:
        mx = max(c.shape[0] for c in mine)
        buf = torch.zeros(len(mine), 3, mx, device=dev)
        mask = torch.zeros(len(mine), 1, mx, dtype=torch.bool, device=dev)
        for i, c in enumerate(mine):
            buf[i, :, : c.shape[0]] = c.T
            mask[i, :, : c.shape[0]] = True
        emb = model.encode_fps(buf, mask)
    else:
        emb = empty_codes(model.encoder.c_dim, dev)
    allc = all_gather_codes(emb, [len(a) for a in assign])
    if all(a == list(range(*shard_range(n, r, ws))) for r, a in enumerate(assign)):
        return allc
    inv = assignment_restore(assign, allc["z_inv"].device)
    return {k: v[inv] for k, v in allc.items()}


def sharded_pairs(n_pairs, register_block):
    """Block-partition a list of n_pairs registration problems: `register_block(lo, hi)` -> (R [hi-lo,3,3], t [hi-lo,3,1]) for this
    rank's block (never called on an empty block); the (R | t) rows are all-gathered: -> (R [n,3,3], t [n,3,1]) on every rank."""
    rank, ws = world()
    lo, hi = shard_range(n_pairs, rank, ws)
    if hi > lo:
        R, t = register_block(lo, hi)
        rows = torch.cat([R.reshape(hi - lo, 9), t.reshape(hi - lo, 3)], 1).contiguous()
    else:
        rows = None
    if rows is None:   # an empty block still needs a device / dtype for its zero-row message
        rows = torch.zeros(0, 12, device=_some_device(), dtype=torch.float32)
    allr = all_gather_rows(rows, shard_counts(n_pairs, ws))
    return allr[:, :9].reshape(-1, 3, 3), allr[:, 9:].reshape(-1, 3, 1)


def _some_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def sharded_sdf_grid(model, codes, query, gather=False):
    """configs[4] (dense SDF reconstruction, instances sharded over the node): codes = dict with n rows, query [n,M,3] (or [1,M,3]
    shared by all instances).  Every rank decodes ITS block of instances -> (lo, hi, sdf [hi-lo, M]); the grids stay on the rank that
    made them (8 MB per 128^3 instance: the mesh step runs where the grid is) unless gather=True (then every rank gets [n, M])."""
    rank, ws = world()
    n = codes["z_inv"].shape[0]
    lo, hi = shard_range(n, rank, ws)
    M = query.shape[1]
    dev = codes["z_inv"].device
    if hi > lo:
        q = query.expand(n, -1, -1)[lo:hi].contiguous() if query.shape[0] == 1 else query[lo:hi].contiguous()
        part = {k: v[lo:hi].contiguous() for k, v in codes.items()}
        sdf = model.decoder(q, None, part, return_sdf=True)
    else:
        sdf = torch.zeros(0, M, device=dev)
    if not gather:
        return lo, hi, sdf
    return lo, hi, all_gather_rows(sdf, shard_counts(n, ws))

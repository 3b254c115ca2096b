"""Chromosome sharding across the GPUs of one node and the final gather (SURVEY §8 e).

Every counter of the path is a commutative sum over fragments and a fragment touches one chromosome, so the work
shards by chromosome with NO data-path collective: each rank (one process per GPU) owns a disjoint set of BAM
references, counts only fragments on them and finalises only introns on them.  The only communication is the
end-of-run collection of the result tables on rank 0 over torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests):

  * fixed-shape tables (IR rows, SpansPoint / ROI / chromosome counters): rows a rank does not own are zero, so the
    collection is one reduce(SUM) per table (x + 0 = x exactly, also in fp64) — tens of MB, latency-bound;
  * junction tables have a rank-dependent length: all_gather of the lengths, then a padded gather.

The reference has no counterpart (single process); this is the new [DESIGN] item demanded by north_star.
"""
import numpy as np


def shard_chromosomes(weights, world):
    """Longest-processing-time bin packing of chromosomes onto `world` ranks.

    weights: expected work per BAM reference (e.g. fragments, or length as a proxy).  Returns a list of uint8 masks
    (one per rank, len = n_chr), deterministic so that every rank computes the same assignment."""
    w = np.asarray(weights, dtype=np.float64)
    order = np.argsort(-w, kind="stable")
    load = np.zeros(world)
    masks = [np.zeros(len(w), np.uint8) for _ in range(world)]
    for c in order:
        r = int(np.argmin(load))            # ties -> lowest rank
        masks[r][c] = 1
        load[r] += w[c]
    return masks


def feature_intervals(ref_texts, chr_name):
    """[start, end] of the introns, ref-sj junctions and read-continues points of one chromosome: a reference may be cut
    only outside all of them (ROIs may span a cut: ref-ROI.bed lists the intergenic regions themselves, and FragmentsInROI
    is counted by the owner of the fragment's start)."""
    cover, sj, rc, roi = ref_texts
    iv = []
    for text, a, b in ((cover, 1, 2), (sj, 1, 2)):
        for line in text.splitlines():
            f = line.split("\t")
            if len(f) > b and f[0] == chr_name:
                iv.append((int(f[a]), int(f[b])))
    for line in rc.splitlines():
        f = line.split("\t")
        if len(f) > 1 and f[0] == chr_name:
            iv.append((int(f[1]), int(f[1])))
    return iv


def intergenic_cut(ref_texts, chr_name, chr_len, target, margin=1000):
    """The position nearest to `target` that no feature of the reference files spans or touches (with `margin` bases of
    room on both sides: reads of the neighbouring features end well before it), or None if the chromosome has no such
    gap.  This is where irf_chr_map_update_ranges may cut the reference between two contexts."""
    iv = sorted(feature_intervals(ref_texts, chr_name))
    gaps, at = [], 0
    for a, b in iv:
        if a - margin > at + margin:
            gaps.append((at + margin, a - margin))
        at = max(at, b + 1)
    if chr_len - margin > at + margin:
        gaps.append((at + margin, chr_len - margin))
    best = None
    for a, b in gaps:
        c = min(max(int(target), a), b)
        if best is None or abs(c - target) < abs(best - target):
            best = c
    return best


def shard_with_split(weights, chr_lens, world, ref_texts, chr_names, weight_left_of=None):
    """shard_chromosomes, then the heaviest reference is cut in two at an intergenic gap when it alone weighs more than
    the mean load of a rank (a sample where one reference exceeds 1/N of the reads; SURVEY §8 e): its right part goes to
    the lightest rank.  weight_left_of(chr, pos) -> fraction of the reference's weight left of pos (default: pos / len).
    Returns (masks, own_start, own_end): per rank an uint8 mask and uint32 arrays for irf_chr_map_update_ranges."""
    w = np.asarray(weights, dtype=np.float64)
    masks = shard_chromosomes(w, world)
    n = len(w)
    starts = [np.zeros(n, np.uint32) for _ in range(world)]
    ends = [np.full(n, 0xffffffff, np.uint32) for _ in range(world)]
    heavy = int(np.argmax(w))
    if world < 2 or w[heavy] <= w.sum() / world:
        return masks, starts, ends
    owner = next(r for r in range(world) if masks[r][heavy])
    loads = np.array([float(np.dot(m, w)) for m in masks])
    other = int(np.argmin(np.where(np.arange(world) == owner, np.inf, loads)))
    # the owner keeps the left part up to where both ranks carry the same load
    keep = (loads[other] + w[heavy] - (loads[owner] - w[heavy])) / (2.0 * w[heavy])
    keep = min(max(keep, 0.05), 0.95)
    cut = intergenic_cut(ref_texts, chr_names[heavy], int(chr_lens[heavy]), keep * int(chr_lens[heavy]))
    if cut is None:
        return masks, starts, ends
    ends[owner][heavy] = cut
    masks[other][heavy] = 1
    starts[other][heavy] = cut
    return masks, starts, ends


def plane_bases(chr_lens, owned):
    """chr_base[] exactly as the engine lays out the depth planes of a rank (owned references only, len+1 rounded
    up to 64) — needed to decode that rank's packed junction keys."""
    base = np.zeros(len(chr_lens) + 1, np.uint64)
    off = 0
    for i, l in enumerate(chr_lens):
        base[i] = off
        if owned[i]:
            off += ((int(l) + 1 + 63) // 64) * 64
    base[len(chr_lens)] = off
    return base


JUNC_SPAN_BITS = 28
JUNC_LEFT_SHIFT = JUNC_SPAN_BITS + 1


def decode_junction_keys(keys, chr_lens, owned):
    """Packed keys of one rank -> (chr, left, right, dir) arrays."""
    keys = np.asarray(keys, np.uint64)
    base = plane_bases(chr_lens, owned)
    gl = keys >> np.uint64(JUNC_LEFT_SHIFT)
    span = (keys >> np.uint64(1)) & np.uint64((1 << JUNC_SPAN_BITS) - 1)
    d = (keys & np.uint64(1)).astype(np.uint8)
    owned_idx = np.nonzero(np.asarray(owned))[0]
    starts = base[owned_idx]
    k = np.searchsorted(starts, gl, side="right") - 1
    chr_id = owned_idx[k].astype(np.uint32) if len(keys) else np.zeros(0, np.uint32)
    left = (gl - starts[k]).astype(np.uint32) if len(keys) else np.zeros(0, np.uint32)
    right = (left.astype(np.uint64) + span).astype(np.uint32)
    return chr_id, left, right, d


def pack_tables(tables, keys):
    """The named tables (one dtype) as one flat tensor, in the order of `keys`."""
    import torch
    return torch.cat([tables[k].reshape(-1) for k in keys])


def unpack_tables(flat, tables, keys):
    """Views of `flat` with the shapes of the named tables (inverse of pack_tables)."""
    out, at = {}, 0
    for k in keys:
        n = tables[k].numel()
        out[k] = flat[at:at + n].view(tables[k].shape)
        at += n
    assert at == flat.numel()
    return out


def gather_tables(tables, dist=None, root=0):
    """Collect the per-rank result tables on `root`.

    tables: dict of torch tensors on the rank's device —
        fixed shape, zero where not owned: 'ir_f64' [8,n], 'ir_u32' [6,n] (int32 view), 'sp_cnt', 'roi_cnt', 'chr_cnt',
                                           'known_cnt' [2,n_ref_junction] (reads on ref-sj junctions)
        variable length (novel junctions): 'junc_keys' (int64 view of the packed keys), 'junc_counts' (int32)
    Returns on root: dict with the summed fixed tables and 'junc_keys'/'junc_counts' as lists (one entry per rank);
    on other ranks: None.  With dist None (single process) the input is returned in the same shape."""
    import torch
    fixed = [k for k in ("ir_f64", "ir_u32", "ir_dir_f64", "ir_dir_u32", "sp_cnt", "roi_cnt", "chr_cnt", "known_cnt")
             if k in tables]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = {k: tables[k] for k in fixed}
        out["junc_keys"] = [tables["junc_keys"]]
        out["junc_counts"] = [tables["junc_counts"]]
        return out
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {}
    # On the device (RCCL) the fixed-shape tables of one dtype travel as ONE flat buffer: a collective's cost there is its
    # launch, not its bytes (two reduces — fp64 columns, integer tables — instead of one per table).  Host tensors (gloo:
    # tests and rehearsals) go one by one: gloo is slower on the concatenated buffer (measured: 45 vs 20 ms, world 2).
    by_dtype = {}
    for k in fixed:
        if tables[k].is_cuda:
            by_dtype.setdefault(tables[k].dtype, []).append(k)
        else:
            t = tables[k].contiguous()
            dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
            out[k] = t
    for dt, ks in by_dtype.items():
        flat = pack_tables(tables, ks)
        dist.reduce(flat, dst=root, op=dist.ReduceOp.SUM)
        out.update(unpack_tables(flat, tables, ks))
    # junctions: lengths, then ONE padded gather of keys and counts (both as int64)
    keys, counts = tables["junc_keys"], tables["junc_counts"]
    n_local = torch.tensor([keys.numel()], dtype=torch.int64, device=keys.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(2 * m, dtype=torch.int64, device=keys.device)
    pad[: sizes[rank]] = keys
    pad[m: m + sizes[rank]] = counts.to(torch.int64)
    if rank == root:
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=root)
        out["junc_keys"] = [b[:s] for b, s in zip(bufs, sizes)]
        out["junc_counts"] = [b[m: m + s].to(torch.int32) for b, s in zip(bufs, sizes)]
    else:
        dist.gather(pad, None, dst=root)
    return out if rank == root else None


def global_directionality(votes, dist=None, ratio=10):
    """JunctionCount::Directional() over all shards: all-reduce(SUM) of the per-rank votes (same, opposite), then the
    same rule the engine applies locally.  votes: int64 tensor of 2 elements on the rank's device."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(votes, op=dist.ReduceOp.SUM)
    same, opp = int(votes[0].item()), int(votes[1].item())
    if same > opp * ratio:
        return 1
    if opp > same * ratio:
        return -1
    return 0


def mask_unowned_rows(ir_f64, ir_u32, intron_owned):
    """Zero the IR rows of introns this rank does not own (in place), so reduce(SUM) acts as a gather."""
    m = intron_owned.to(ir_f64.device)
    ir_f64.mul_(m.to(ir_f64.dtype).unsqueeze(0))
    ir_u32.mul_(m.to(ir_u32.dtype).unsqueeze(0))
    return ir_f64, ir_u32

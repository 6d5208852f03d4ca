"""Template-bank sharding across the GPUs of one node (BASELINE.json north_star).

Not present in the reference (its inference is single-GPU): rank r keeps templates
[lo_r, hi_r) of EVERY object.  Per step each rank
  1. runs the ViT on its own B crops, all-gathers the matcher-normalised query features, patch
     masks and labels                                             (exchange #1, RCCL all-gather)
  2. matches all W*B crops against its shard (gp_match_tiles) and takes a local top-k
  3. sends every other rank the k per-shard candidates of THAT rank's crops -- global template id, score and
     the 256-patch record (idx u8, score f32, mask f32), packed into one byte row   (exchange #2, RCCL all-to-all)
  4. merges the W*k candidates of its own crops: "higher score, then lower global id" -- exactly
     gp_topk's order over all N templates, so results equal the unsharded path bit-for-bit
  5. continues locally (IST regressor, RANSAC, recovery) with the small replicated banks.
Collectives go through torch.distributed ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist

import os

P = 256


def _always_collective():
    """GIGAPOSE_FORCE_COLLECTIVES=1: issue the all-gathers even with one rank (the single-GPU box exercises the
    RCCL calls, streams and layouts of the N>1 path this way; tests/test_gpu_e2e.py)."""
    return os.environ.get("GIGAPOSE_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()


REC_BYTES = 8 + 4 + P + 4 * P + 4 * P  # id i64, score f32, idx u8[256], score f32[256], mask f32[256]


def shard_bounds(n_templates, world, rank):
    """Contiguous, near-equal split of template indices: rank r owns [lo, hi)."""
    lo = (n_templates * rank + world - 1) // world
    hi = (n_templates * (rank + 1) + world - 1) // world
    return lo, hi


def _needs_host_staging(t, group):
    """gloo moves host memory only: device rows are staged through the host (the world-size-2 GPU test runs two ranks on ONE
    MI355X, which RCCL refuses -- "duplicate GPU" -- so that test drives the real kernels over gloo; RCCL takes device tensors)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _StagedWork:
    """Handle of a host-staged collective: wait() finishes it and copies the result to the device buffer."""

    def __init__(self, work, host, out):
        self.work, self.host, self.out = work, host, out

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.out.copy_(self.host)


def all_gather_rows(rows, group=None, async_op=False):
    """ONE collective: every rank contributes the same number of equal-length u8 rows (n, L); returns ((W*n, L) in rank
    order, work handle or None).  all_gather_into_tensor = a single RCCL ncclAllGather over the flat buffer (one launch
    instead of one per tensor)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not _always_collective():
        return rows, None
    rows = rows.contiguous()
    out = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    if _needs_host_staging(rows, group):
        host = torch.empty(out.shape, dtype=out.dtype)
        work = _StagedWork(dist.all_gather_into_tensor(host, rows.cpu(), group=group, async_op=async_op), host, out)
        if not async_op:
            work.wait()
        return out, (work if async_op else None)
    work = dist.all_gather_into_tensor(out, rows, group=group, async_op=async_op)
    return out, (work if async_op else None)


def all_to_all_rows(rows, group=None):
    """rows (W * n, ...): chunk r (n rows) goes to rank r.  Returns (W, n, ...): chunk s = what rank s sent to this rank.
    RCCL: one all_to_all_single (each rank receives only what it will use); gloo has no all-to-all, so there (CPU tests, and
    the two-ranks-on-one-GPU test) the same result is cut out of an all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rows.shape[0] % world:
        # all_to_all_single splits dim 0 evenly: a remainder would raise on this rank only and leave the others inside the collective
        raise ValueError(f"all_to_all_rows: {rows.shape[0]} rows do not split over {world} ranks (every rank must pass W * n rows)")
    n = rows.shape[0] // world
    if world == 1 and not _always_collective():
        return rows.reshape((1, n) + tuple(rows.shape[1:]))
    rows = rows.contiguous()
    if dist.get_backend(group) == "nccl":
        out = torch.empty_like(rows)
        dist.all_to_all_single(out, rows, group=group)
        return out.reshape((world, n) + tuple(rows.shape[1:]))
    rank = dist.get_rank(group)
    allrows, _ = all_gather_rows(rows, group)                                        # (W * W*n, ...)
    return allrows.reshape((world, world, n) + tuple(rows.shape[1:]))[:, rank].contiguous()


def require_same_batch(n_own, device, group=None):
    """Every rank must enter a sharded step with the SAME number of crops: the exchanges are fixed-size collectives
    (all_gather_into_tensor / all_to_all_single), and ranks that disagree on the size do not fail -- they hang, or pair rows of
    different shapes.  One tiny all-reduce (MAX of [B, -B]) + a host read: for callers that synchronise with the host anyway
    (GigaPose.eval_retrieval); raises ValueError on EVERY rank together when the sizes differ, so nobody is left inside a
    collective.  The raw predict() loop of bench.py passes a fixed B and skips this."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else device
    t = torch.tensor([n_own, -n_own], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise ValueError(f"template-sharded step: ranks passed different batch sizes ({lo}..{hi} crops; this rank {n_own}) -- pad the "
                         "last batch so that every rank holds the same number of crops")


def all_gather_cat(t, group=None):
    """all-gather equal-shaped tensors and concatenate along dim 0 (rank order); tuples element-wise."""
    if isinstance(t, (tuple, list)):
        return tuple(all_gather_cat(x, group) for x in t)
    return all_gather_rows(t, group)[0]


def _align16(n):
    return (n + 15) // 16 * 16


def pack_query(q, qmask, labels0, aux=None):
    """Exchange #1 payload, one byte row per crop: matcher-normalised features (f32 (B,C,256), or the f16 hi / lo planes
    (B,256,Cp) of the split numerics) | patch mask f32[256] | label i32 [| aux i32], every field at a 16-byte aligned offset
    of a preallocated row (one copy per field, no concatenation temporaries): 1 MB per crop at C = 1024.  `aux` (B,) int32 is
    an optional per-crop word the caller wants every rank to see (the accumulated drop-in flow sends its "this rank has nothing
    left" flag with it, gigapose_amd/sharded_flow.py); it is the LAST field and unpack_query skips it (unpack_aux reads it).
    Returns (rows (B, L) u8, layout)."""
    B = qmask.shape[0]
    feats = list(q) if isinstance(q, (tuple, list)) else [q]
    fields = [f.contiguous() for f in feats] + [qmask.contiguous().float(), labels0.to(torch.int32).contiguous()]
    if aux is not None:
        fields.append(aux.to(device=qmask.device, dtype=torch.int32).contiguous())
    layout, off = [], 0
    for j, f in enumerate(fields):
        nbytes = f[0].numel() * f.element_size() if B else 0
        layout.append((tuple(f.shape[1:]), f.dtype, off, nbytes) + (("aux",) if aux is not None and j == len(fields) - 1 else ()))
        off = _align16(off + nbytes)
    rows = torch.empty(B, off, dtype=torch.uint8, device=qmask.device)
    for f, (_, _, o, nbytes, *_tag) in zip(fields, layout):
        rows[:, o:o + nbytes] = f.view(torch.uint8).reshape(B, nbytes)
    return rows, layout


def unpack_query(rows, layout):
    """Inverse of pack_query for (n, L) u8 rows: (features or (hi, lo)), qmask (n,256) f32, labels (n,) i32 (contiguous
    copies: the kernels take dense arrays)."""
    n, out = rows.shape[0], []
    for shape, dtype, o, nbytes, *tag in layout:
        if not tag:
            out.append(rows[:, o:o + nbytes].contiguous().view(dtype).reshape(n, *shape))
    feats, qmask, labels = out[:-2], out[-2], out[-1]
    return (feats[0] if len(feats) == 1 else tuple(feats)), qmask, labels


def unpack_aux(rows, layout):
    """The optional per-crop int32 word of pack_query(aux=...) for (n, L) rows: (n,) int32, or None if the rows carry none."""
    for shape, dtype, o, nbytes, *tag in layout:
        if tag:
            return rows[:, o:o + nbytes].contiguous().view(dtype).reshape(rows.shape[0])
    return None


def pack_candidates(ids_global, scores, rec_idx, rec_score, rec_mask):
    """(B,k) i64, (B,k) f32, (B,k,256) u8/f32/f32 -> (B,k,REC_BYTES) u8 rows."""
    B, k = ids_global.shape
    parts = [ids_global.contiguous().view(torch.uint8).reshape(B, k, 8),
             scores.contiguous().view(torch.uint8).reshape(B, k, 4),
             rec_idx.reshape(B, k, P),
             rec_score.contiguous().view(torch.uint8).reshape(B, k, 4 * P),
             rec_mask.contiguous().view(torch.uint8).reshape(B, k, 4 * P)]
    return torch.cat(parts, dim=2).contiguous()


def unpack_candidates(rows):
    """Inverse of pack_candidates for (..., REC_BYTES) u8 rows."""
    lead = rows.shape[:-1]
    rows = rows.contiguous()
    o = 0
    ids = rows[..., o:o + 8].contiguous().view(torch.int64).reshape(lead); o += 8
    sc = rows[..., o:o + 4].contiguous().view(torch.float32).reshape(lead); o += 4
    ridx = rows[..., o:o + P].contiguous(); o += P
    rsc = rows[..., o:o + 4 * P].contiguous().view(torch.float32).reshape(*lead, P); o += 4 * P
    rma = rows[..., o:o + 4 * P].contiguous().view(torch.float32).reshape(*lead, P)
    return ids, sc, ridx, rsc, rma


def merge_topk(ids, scores, k):
    """ids, scores (B, W*k): select k per row by (score desc, id asc).  Returns positions (B,k)."""
    by_id = torch.sort(ids, dim=1, stable=True).indices
    sc_by_id = torch.gather(scores, 1, by_id)
    by_score = torch.sort(sc_by_id, dim=1, descending=True, stable=True).indices
    return torch.gather(by_id, 1, by_score)[:, :k]


def exchange_and_merge(local_rows, n_own, k, rank, group=None):
    """local_rows (W*B, k, REC_BYTES) u8: this rank's candidates for ALL crops (crop order = rank-major).  Exchange #2 is an
    all-to-all: rank r receives, from every rank, only the candidates of ITS OWN B crops (W * B * k records instead of the
    W * W * B * k an all-gather delivers: 5.9 MB instead of 47 MB per rank at W = 8, B = 64).
    Returns merged ids (B,k) i64, scores (B,k), rec_idx/rec_score/rec_mask (B,k,256) for OWN crops."""
    got = all_to_all_rows(local_rows, group)                                          # (W, B, k, REC): chunk s from rank s
    world = got.shape[0]
    mine = got.permute(1, 0, 2, 3).reshape(n_own, world * k, REC_BYTES)
    ids, sc, ridx, rsc, rma = unpack_candidates(mine)
    pos = merge_topk(ids, sc, k)
    pos3 = pos[:, :, None].expand(-1, -1, P)
    return (torch.gather(ids, 1, pos), torch.gather(sc, 1, pos), torch.gather(ridx, 1, pos3),
            torch.gather(rsc, 1, pos3), torch.gather(rma, 1, pos3))


class ShardedMatcher:
    """Wraps a LocalSimilarity (gigapose_amd.matching) and a MatchBank holding only this rank's
    template shard; produces the same PandasTensorCollection as LocalSimilarity.test_bank."""

    def __init__(self, metric, bank_shard, template_lo, group=None):
        self.metric, self.bank, self.lo, self.group = metric, bank_shard, template_lo, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.last_aux = None
        if bank_shard.N < metric.k:
            raise ValueError(f"shard of {bank_shard.N} templates is smaller than k={metric.k}")

    @torch.no_grad()
    def start_exchange(self, tar_feat, tar_mask, labels0, aux=None):
        """Exchange #1, asynchronous: pack this rank's matcher-normalised query features, patch masks and labels into one byte
        row per crop and all-gather them in ONE collective.  RCCL runs it on its own stream; the caller keeps launching
        independent work (the IST backbone, gigaPose.py) and calls finish() when it needs the matches.  `aux` (B,) int32:
        an optional word per crop that travels with the row; after finish() `self.last_aux` holds all ranks' words (W * B,)."""
        from .matching import patch_grid_mask

        rows, layout = pack_query(self.metric.normalize(tar_feat), patch_grid_mask(tar_mask), labels0, aux)
        allrows, work = all_gather_rows(rows, self.group, async_op=True)
        return dict(rows=allrows, work=work, layout=layout, n_own=tar_feat.shape[0])

    @torch.no_grad()
    def finish(self, h):
        import pandas as pd

        from .tensor_collection import PandasTensorCollection

        m = self.metric
        if h["work"] is not None:
            h["work"].wait()                                                          # current stream waits for the collective
        n_ranks = self.world if (self.world > 1 or _always_collective()) else 1
        if h["rows"].shape[0] != n_ranks * h["n_own"]:
            raise ValueError(f"exchange #1 returned {h['rows'].shape[0]} rows for {n_ranks} ranks x {h['n_own']} crops: every rank must pass the same batch size")
        q, qmask, labels_all = unpack_query(h["rows"], h["layout"])
        self.last_aux = unpack_aux(h["rows"], h["layout"])
        idx, sc, ma, avg = m.match_tiles(q, qmask, self.bank, labels_all)
        ids, score = m.topk(avg)
        rec_idx, rec_score, rec_mask = m.gather_records(ids, idx, sc, ma)
        rows = pack_candidates(ids.long() + self.lo, score, rec_idx, rec_score, rec_mask)
        gid, gsc, ridx, rsc, rma = exchange_and_merge(rows, h["n_own"], m.k, self.rank, self.group)  # exchange #2
        tar_pts, src_pts = m.format_points(ridx.contiguous(), rma.contiguous())
        return PandasTensorCollection(infos=pd.DataFrame(), id_src=gid, score_src=gsc, score_pts=rsc.contiguous(),
                                      tar_pts=tar_pts, src_pts=src_pts)

    def test_bank(self, tar_feat, tar_mask, labels0, aux=None):
        return self.finish(self.start_exchange(tar_feat, tar_mask, labels0, aux))

"""Multi-GPU sampling: independent utterances sharded over ranks, one all-gather to collate.

The reference pins inference to one GPU (convofusion/config.py:92-95).  Utterances are independent
(SURVEY.md section 8e), so each rank (one process per GPU) samples its own slice with its own Philox
sub-stream (``first_utterance`` = global id of its first utterance, so results do not depend on the
number of GPUs) and the final latents are collated with ONE ``all_gather`` (RCCL over xGMI on MI355X;
gloo in the CPU tests).  There is no per-step communication.
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous [start, stop) slice of ``total`` utterances owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_cfg_batch(t, start, stop, total, chunks=7):
    """Slice utterances [start, stop) out of a chunk-major guidance batch [chunks*total, ...]."""
    if t is None:
        return None
    v = t.reshape(chunks, total, *t.shape[1:])
    return v[:, start:stop].reshape(chunks * (stop - start), *t.shape[1:]).contiguous()


def gather_latents(local, total, group=None):
    """all_gather of per-rank latents [b_r, L, 128] -> [total, L, 128] (ragged shards are padded)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    ws = dist.get_world_size(group)
    sizes = [shard_range(total, r, ws) for r in range(ws)]
    mx = max(b - a for a, b in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad.contiguous(), group=group)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)


def sample_sharded(sample_fn, encoder_hidden_states, cond_masks, total_utterances, chunks=7, group=None):
    """Run ``sample_fn(enc_shard, masks_shard, B=<local>, first_utterance=<global id>)`` on this rank's
    utterances and return the gathered latents [total, L, 128] on every rank."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    a, b = shard_range(total_utterances, rank, ws)
    enc = [shard_cfg_batch(m, a, b, total_utterances, chunks) for m in encoder_hidden_states]
    masks = {k: shard_cfg_batch(v, a, b, total_utterances, chunks) for k, v in (cond_masks or {}).items()}
    local = sample_fn(enc, masks, B=b - a, first_utterance=a)
    return gather_latents(local, total_utterances, group)

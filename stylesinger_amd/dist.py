"""Data-parallel driver: utterances sharded over ranks, ONE collective to publish the mel batch.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm).  Utterances are
independent at inference, so the only exchange is a single `all_gather_into_tensor`
of the equal-sized padded buffer  [B_local, T_max, 80 mel + 1 f0] + lengths — after the mel diffusion, before
vocoding.  Each rank then vocodes its own shard (the gather publishes the full mel batch; the
waveforms stay sharded).  Sharding rule = the reference's dataloader rule `x[rank::num_replicas]`
(tasks/tts/tts_base.py:129-132).
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """tasks/tts/tts_base.py:132: items rank, rank+W, ..."""
    return list(range(rank, n_items, world))


def gather_mels(mel, f0, lens, group=None):
    """mel [B_local,T,M], f0 [B_local,T], lens int32 [B_local] -> (mel_all [W*B_local,T,M], f0_all, lens_all).
    Equal-shaped buffers on every rank (pad T to the global max before calling)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mel, f0, lens
    W = dist.get_world_size(group)
    Bl, T, M = mel.shape
    payload = torch.cat([mel, f0[:, :, None], lens.to(mel.dtype)[:, None, None].expand(-1, T, 1)], dim=-1).contiguous()
    out = torch.empty(W * Bl, T, M + 2, device=mel.device, dtype=mel.dtype)
    dist.all_gather_into_tensor(out, payload, group=group)
    return out[:, :, :M].contiguous(), out[:, :, M].contiguous(), out[:, 0, M + 1].round().to(torch.int32)


def global_max_int(x, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def run_sharded(infer_fn, vocode_fn, items, rank, world, pad_T, group=None):
    """Generic DP step used by bench.py and tested on CPU/gloo with fake compute functions.
    infer_fn(list_of_items) -> (mel [b,T,M], f0 [b,T], lens [b]); vocode_fn(mel, f0, lens) -> wav [b, T*hop]."""
    mine = [items[i] for i in shard_indices(len(items), rank, world)]
    mel, f0, lens = infer_fn(mine)
    if mel.shape[1] < pad_T:
        mel = torch.nn.functional.pad(mel, (0, 0, 0, pad_T - mel.shape[1]))
        f0 = torch.nn.functional.pad(f0, (0, pad_T - f0.shape[1]))
    mel_all, f0_all, lens_all = gather_mels(mel, f0, lens, group)
    wav = vocode_fn(mel, f0, lens)
    return dict(mel_all=mel_all, f0_all=f0_all, lens_all=lens_all, wav_local=wav, local_indices=shard_indices(len(items), rank, world))

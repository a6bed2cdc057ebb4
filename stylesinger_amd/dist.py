"""Data-parallel driver: utterances sharded over ranks, ONE collective to publish the mel batch.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm).  Utterances are
independent at inference, so the only exchange is a single `all_gather_into_tensor`
of the equal-sized padded buffer  [B_local, T_max, 80 mel + 1 f0 + 1 len] — after the mel diffusion, before
vocoding.  Each rank then vocodes its own shard (the gather publishes the full mel batch; the
waveforms stay sharded).  Sharding rule = the reference's dataloader rule `x[rank::num_replicas]`
(tasks/tts/tts_base.py:129-132).  When the item count is not a multiple of the world size the short
shards are padded with empty slots (len 0) so that every rank contributes the same shape to the collective.
"""
import os

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """tasks/tts/tts_base.py:132: items rank, rank+W, ..."""
    return list(range(rank, n_items, world))


def shard_slots(n_items, world):
    """Slots per rank = ceil(n/W): slot j of rank r holds global item r + j*W (valid iff < n_items)."""
    return (n_items + world - 1) // world


# A world of one rank has nothing to exchange, so the collective is skipped there - unless this flag is set (env SS_FORCE_COLLECTIVE=1
# or dist.FORCE_COLLECTIVE = True): then an initialised process group of ANY size, W = 1 included, takes the real
# all_gather_into_tensor branch. That is how a 1-GPU box drives the RCCL code path (tests/test_gpu_round4.py).
FORCE_COLLECTIVE = os.environ.get("SS_FORCE_COLLECTIVE", "0") == "1"


def _is_dist(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVE


_comm_streams = {}


def comm_stream(device):
    """The ONE stream of this process on which the data path's collectives are issued. Steps may run on several HIP streams
    (batches in flight), but every rank calls gather_mels in the same host order, and issuing them all from one stream keeps
    that order on the device too: in-flight batches can never interleave their collectives differently on different ranks."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _comm_streams:
        _comm_streams[key] = torch.cuda.Stream(device=key)
    return _comm_streams[key]


def gather_mels(mel, f0, lens, group=None):
    """mel [B_local,T,M], f0 [B_local,T], lens int32 [B_local] -> (mel_all [W*B_local,T,M], f0_all, lens_all int32).
    Equal-shaped buffers on every rank (pad T to the global max and B_local to shard_slots() before calling).
    The lengths ride in the same buffer bit-cast to fp32 (a collective moves bits), so they come back exact.
    The collective runs on the dedicated communication stream: it waits (event) for the calling stream's producers, and the
    calling stream waits (event) for it - callers on different step streams stay asynchronous to each other."""
    if not _is_dist(group):
        return mel, f0, lens
    W = dist.get_world_size(group)
    Bl, T, M = mel.shape
    payload = torch.empty(Bl, T, M + 2, device=mel.device, dtype=torch.float32)
    payload[:, :, :M] = mel
    payload[:, :, M] = f0
    payload[:, :, M + 1] = lens.to(torch.int32).view(torch.float32)[:, None]
    if not payload.is_cuda:
        out = torch.empty(W * Bl, T, M + 2, dtype=torch.float32)
        dist.all_gather_into_tensor(out, payload, group=group)
        return out[:, :, :M].contiguous(), out[:, :, M].contiguous(), out[:, 0, M + 1].contiguous().view(torch.int32)
    cur = torch.cuda.current_stream(mel.device)
    cs = comm_stream(mel.device)
    ready = torch.cuda.Event()
    ready.record(cur)
    out = torch.empty(W * Bl, T, M + 2, device=mel.device, dtype=torch.float32)
    with torch.cuda.stream(cs):
        cs.wait_event(ready)
        if dist.get_backend(group) == "gloo":
            # orchestration tests on one device run over gloo, whose collectives take host buffers (the copy synchronises the
            # communication stream only); RCCL (backend "nccl") gathers the device buffer directly over xGMI
            host = torch.empty(W * Bl, T, M + 2, dtype=torch.float32)
            dist.all_gather_into_tensor(host, payload.cpu(), group=group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, payload, group=group)
        done = torch.cuda.Event()
        done.record(cs)
    payload.record_stream(cs)
    out.record_stream(cur)
    cur.wait_event(done)
    return out[:, :, :M].contiguous(), out[:, :, M].contiguous(), out[:, 0, M + 1].contiguous().view(torch.int32)


def global_max_int(x, group=None):
    if not _is_dist(group):
        return int(x)
    t = torch.tensor([int(x)], dtype=torch.int64, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def pad_shard(mel, f0, lens, slots, pad_T):
    """Pad a local result to [slots, pad_T, ...] (empty slots have len 0 and zero payload)."""
    Bl, T, M = mel.shape
    if T < pad_T:
        mel = torch.nn.functional.pad(mel, (0, 0, 0, pad_T - T))
        f0 = torch.nn.functional.pad(f0, (0, pad_T - T))
    if Bl < slots:
        mel = torch.nn.functional.pad(mel, (0, 0, 0, 0, 0, slots - Bl))
        f0 = torch.nn.functional.pad(f0, (0, 0, 0, slots - Bl))
        lens = torch.nn.functional.pad(lens, (0, slots - Bl))
    return mel.contiguous(), f0.contiguous(), lens.contiguous()


def gathered_order(n_items, world):
    """Global item index of each row of a gathered buffer (row r*slots + j <-> item r + j*W), -1 for empty slots."""
    slots = shard_slots(n_items, world)
    return [(r + j * world) if (r + j * world) < n_items else -1 for r in range(world) for j in range(slots)]


def run_sharded(infer_fn, vocode_fn, items, rank, world, pad_T, group=None):
    """Generic DP step used by bench.py and the multi-process tests.
    infer_fn(list_of_items) -> (mel [b,T,M], f0 [b,T], lens int32 [b]); vocode_fn(mel, f0, lens) -> wav [b, T*hop].
    Returns the gathered mel batch in ITEM order (empty slots dropped) and this rank's waveforms."""
    idx = shard_indices(len(items), rank, world)
    mel, f0, lens = infer_fn([items[i] for i in idx])
    slots = shard_slots(len(items), world)
    mel_p, f0_p, lens_p = pad_shard(mel, f0, lens.to(torch.int32), slots, pad_T)
    mel_all, f0_all, lens_all = gather_mels(mel_p, f0_p, lens_p, group)
    if _is_dist(group):
        order = gathered_order(len(items), world)
        rows = sorted((g, r) for r, g in enumerate(order) if g >= 0)
        sel = torch.tensor([r for _, r in rows], device=mel_all.device)
        mel_all, f0_all, lens_all = mel_all[sel], f0_all[sel], lens_all[sel]
    else:
        mel_all, f0_all, lens_all = mel_all[:len(idx)], f0_all[:len(idx)], lens_all[:len(idx)]
    wav = vocode_fn(mel_p[:len(idx)], f0_p[:len(idx)], lens_p[:len(idx)])
    return dict(mel_all=mel_all, f0_all=f0_all, lens_all=lens_all, wav_local=wav, local_indices=idx)

"""Style-transfer sweep (BASELINE config 5): every reference voice/style x every target score.

The sweep is embarrassingly parallel over (reference, target) pairs, so there is no data-path collective: references are
sharded over ranks with the reference's dataloader rule (`x[rank::num_replicas]`, tasks/tts/tts_base.py:132), which also
keeps the per-reference cache local to one GPU:

  * the Residual Style Adaptor + RQ lookup + `l1` depend on the reference only -> `StyleSingerHIP.encode_style` runs once
    per reference (n_refs times instead of n_refs * n_targets) and its output stays resident in HBM;
  * a batch is ONE target score x `batch` references, so every item of a batch has the same frame count (no padding waste)
    and the hipGraphs of the f0 diffusions / DDIM mel sampler are keyed by (batch, T bucket, steps) and replayed;
  * the mel sampler is the strided deterministic DDIM (`ss_meldiff_sample_ddim`), default 50 network evaluations;
  * default batch = 32 references per target (the per-rank share of a 256-reference sweep on 8 GPUs): the denoiser GEMMs
    then run 4 rounds of blocks per launch instead of one, where they reach their large-size efficiency (DESIGN.md §5,
    "Size dependence": 73 % of the fp32 roof at 12 000 rows per launch, 101 % algorithmic at 180 000).
"""
import torch

from .dist import shard_indices


def sweep_plan(n_refs, n_targets, rank, world, batch):
    """-> list of (target index, [reference indices]) for this rank; each (ref, target) pair appears on exactly one rank."""
    mine = shard_indices(n_refs, rank, world)
    plan = []
    for t in range(n_targets):
        for i in range(0, len(mine), batch):
            plan.append((t, mine[i:i + batch]))
    return plan


def bucket_frames(T, bucket):
    """Frame count rounded up to the graph bucket (the extra frames are padding: mel2ph = 0)."""
    return T if bucket <= 1 else (T + bucket - 1) // bucket * bucket


class StyleCache:
    """Per-reference style encodings, resident on the device."""

    def __init__(self, model):
        self.model = model
        self.items = {}
        self.hits = 0      # lookups served from the cache (a reference seen before: every target after its first)
        self.encodes = 0   # style-encoder runs (one per reference)

    def get(self, idx, ref):
        if idx in self.items:
            self.hits += 1
        else:
            self.encodes += 1
            dev = next(iter(self.model.buffers())).device
            sc = self.model.encode_style(ref["ref_mels"][None].to(dev), ref["ref_f0"][None].to(dev))
            self.items[idx] = {k: v[0] for k, v in sc.items()}
        return self.items[idx]

    def batch(self, idxs, refs):
        """Stack the cached encodings of `idxs` (zero-padded to the longest reference; attention masks by lens_r)."""
        rows = [self.get(i, refs[i]) for i in idxs]
        Tr = max(r["sty"].shape[0] for r in rows)

        def pad(x):
            if x.shape[0] == Tr:
                return x
            return torch.nn.functional.pad(x, (0, 0) * (x.dim() - 1) + (0, Tr - x.shape[0]))
        return {k: torch.stack([r[k] if r[k].dim() == 0 else pad(r[k]) for r in rows]).contiguous() for k in rows[0]}


@torch.no_grad()
def style_transfer_sweep(infer, refs, targets, rank=0, world=1, batch=32, ddim_steps=50, t_bucket=1, vocode=True, emit=None, seed=None,
                         stats=None):
    """refs[i] = dict(ref_mels [Tr,80], ref_f0 [Tr], spk_embed [256], emo_embed [256]);
    targets[j] = dict(txt_tokens [Tp], note [Tp], note_dur [Tp], note_type [Tp], mel2ph [T]).
    Calls emit(ref_idx, target_idx, mel [T,80], f0 [T], wav [T*hop] or None) per pair; returns the number of pairs done and
    the number of mel frames produced by this rank. `stats` (a dict, optional) receives the per-reference style-cache accounting of this
    call: references on this rank, style-encoder runs, cache hits, hit rate."""
    model, dev = infer.model, infer.device
    cache = StyleCache(model)
    n_pairs = n_frames = 0
    for t_idx, ref_idxs in sweep_plan(len(refs), len(targets), rank, world, batch):
        tgt = targets[t_idx]
        nb = len(ref_idxs)
        T = tgt["mel2ph"].shape[0]
        Tb = bucket_frames(T, t_bucket)
        mel2ph = torch.nn.functional.pad(tgt["mel2ph"], (0, Tb - T)).to(dev)

        def rep(x):
            return x.to(dev)[None].expand(nb, *x.shape).contiguous()
        sc = cache.batch(ref_idxs, refs)
        out = model(rep(tgt["txt_tokens"]), mel2ph=mel2ph[None].expand(nb, -1).contiguous(),
                    spk_embed=torch.stack([refs[i]["spk_embed"] for i in ref_idxs]).to(dev),
                    emo_embed=torch.stack([refs[i]["emo_embed"] for i in ref_idxs]).to(dev),
                    ref_mels=None, ref_f0=None, global_steps=infer.hparams.get("diff_start", 0) + 1, infer=True,
                    note=rep(tgt["note"]), note_dur=rep(tgt["note_dur"]), note_type=rep(tgt["note_type"]),
                    style_cache=sc, sampler="ddim", ddim_steps=ddim_steps, **({} if seed is None else {"seed": seed}))
        mel, f0, lens = out["mel_out"], out["f0_denorm"], out["lens"]
        wav = infer.vocode(mel, f0, lens) if vocode else None
        n_pairs += nb
        n_frames += nb * T
        if emit is not None:
            for k, r_idx in enumerate(ref_idxs):
                emit(r_idx, t_idx, mel[k, :T], f0[k, :T], None if wav is None else wav[k, :T * infer.vocoder.model.hop])
    if stats is not None:
        look = cache.hits + cache.encodes
        stats.update(refs_on_rank=len(shard_indices(len(refs), rank, world)), style_encodes=cache.encodes, style_cache_hits=cache.hits,
                     style_cache_hit_rate=(cache.hits / look) if look else 0.0, pairs=n_pairs)
    return n_pairs, n_frames

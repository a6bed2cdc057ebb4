"""Inference entrypoint of the HIP path: mirror of `inference/StyleSinger.py::StyleSingerInfer`.

`StyleSingerInfer(hparams).forward_model(inp)` keeps the reference's single-utterance contract
(inference/StyleSinger.py:41-63: run the model, drop all-zero frames, clip the mel to
[mel_vmin, mel_vmax], vocode with the predicted f0).  `infer_batch` is the batched, device-resident
form the benchmark and the data-parallel driver use.  The feature extractors of `preprocess_input`
(:94-137) are wired in as far as their models are vendored: `preprocess_batch` computes the reference mel, the emotion
embedding and the normalised f0 contour on the device; the speaker embedding (resemblyzer) and the f0
tracker (parselmouth) are un-vendored third-party models and stay inputs.
"""
import numpy as np
import torch

from . import lib as L
from .config import make_hparams, make_vocoder_config
from .model import StyleSingerHIP
from .vocoder import get_vocoder_cls


class StyleSingerInfer:
    def __init__(self, hparams=None, device=None, model_state=None, vocoder_state=None, vocoder_config=None, dictionary=None):
        self.hparams = make_hparams(hparams)
        if device is None:
            if not torch.cuda.is_available():
                raise L.StyleSingerHipError("StyleSingerInfer (HIP) needs a GPU: there is no CPU path")
            device = "cuda"
        self.device = torch.device(device)
        self.model = self.build_model(dictionary, model_state)
        self.model.eval()
        self.model.to(self.device)
        vcfg = make_vocoder_config(vocoder_config)
        if not (vocoder_config and "mfma_precision" in vocoder_config):
            vcfg["mfma_precision"] = self.hparams.get("mfma_precision", "fp32")
        self.vocoder = get_vocoder_cls(self.hparams)(config=vcfg, state_dict=vocoder_state,
                                                     device=self.device, hparams=self.hparams)

    @classmethod
    def from_checkpoints(cls, hparams, exp_dir, vocoder_dir, device=None, dictionary=None):
        """Build from the reference's on-disk checkpoints (inference/StyleSinger.py:34-39 + hifigan_nsf.py:46-61):
        `exp_dir` = checkpoints/<exp_name> (newest model_ckpt_steps_*.ckpt), `vocoder_dir` = hparams['vocoder_ckpt']."""
        from . import ckpt
        state, _ = ckpt.read_state(exp_dir, "model")
        if state is None:
            raise FileNotFoundError(f"| ckpt not found in {exp_dir}.")
        vstate, vcfg = ckpt.load_vocoder_ckpt(vocoder_dir)
        return cls(hparams, device=device, model_state=state, vocoder_state=vstate, vocoder_config=vcfg, dictionary=dictionary)

    def build_model(self, dictionary=None, state=None):
        model = StyleSingerHIP(dictionary, hparams=self.hparams)
        if state is not None:
            model.load_state_dict(state, strict=False)
        return model

    # ---- batched, device resident -------------------------------------------------------------
    @torch.no_grad()
    def infer_batch(self, batch, noise=None, vocoder_noise=None, seed=None, vocode=True, plan_slot=0):
        """batch: dict of device tensors (txt_tokens, note, note_dur, note_type, spk_embed, emo_embed, ref_mels,
        ref_f0, optional mel2ph).  Returns dict(mel [B,T,80], f0 [B,T], lens int32 [B], wav [B,T*hop])."""
        hp = self.hparams
        seed = hp["seed"] if seed is None else seed
        out = self.model(batch["txt_tokens"], mel2ph=batch.get("mel2ph"), spk_embed=batch["spk_embed"], emo_embed=batch["emo_embed"],
                         ref_mels=batch["ref_mels"], ref_f0=batch["ref_f0"], global_steps=320000, infer=True, note=batch["note"],
                         note_dur=batch["note_dur"], note_type=batch["note_type"], noise=noise, seed=seed, plan_slot=plan_slot)
        res = dict(mel=out["mel_out"], f0=out["f0_denorm"], lens=out["lens"], model_out=out)
        if vocode:
            res["wav"] = self.vocode(out["mel_out"], out["f0_denorm"], out["lens"], noise=vocoder_noise, seed=seed + 101)
        return res

    @torch.no_grad()
    def infer_batches(self, batches, in_flight=3, seed=None, vocode=True):
        """Throughput form of infer_batch for a sequence of INDEPENDENT batches: batch i runs on HIP stream i % in_flight with
        its own workspace / hipGraph set (`plan_slot`), so up to `in_flight` batches overlap on the device - one batch's
        single-round kernel launches leave ramps and tails that the others' blocks fill (-10 % wall at C2).
        Yields the result dicts in order; each result is complete (its stream has been waited for) when it is yielded."""
        in_flight = max(1, int(in_flight))
        if not hasattr(self, "_flight_streams") or len(self._flight_streams) < in_flight:
            self._flight_streams = [torch.cuda.Stream(device=self.device) for _ in range(in_flight)]
        seed = self.hparams["seed"] if seed is None else seed
        main = torch.cuda.current_stream(self.device)
        pending = []

        def finish(entry):
            res, strm = entry
            main.wait_stream(strm)
            return res
        for i, batch in enumerate(batches):
            strm = self._flight_streams[i % in_flight]
            strm.wait_stream(main)   # the batch's inputs were produced on the caller's stream
            with torch.cuda.stream(strm):
                res = self.infer_batch(batch, seed=seed + i, vocode=vocode, plan_slot=i % in_flight)
                for v in batch.values():
                    if torch.is_tensor(v):
                        v.record_stream(strm)
            pending.append((res, strm))
            if len(pending) >= in_flight:
                yield finish(pending.pop(0))
        while pending:
            yield finish(pending.pop(0))

    @torch.no_grad()
    def vocode(self, mel, f0, lens, noise=None, seed=1234):
        hp = self.hparams
        mel_c = torch.empty_like(mel)
        L.check(L.load().ss_clip(L.ptr(mel), L.ptr(mel_c), mel.numel(), float(hp["mel_vmin"]), float(hp["mel_vmax"]), L.stream_ptr()), "ss_clip")
        return self.vocoder.spec2wav_batch(mel_c, f0, lens=lens, noise=noise, seed=seed)

    @torch.no_grad()
    def infer_batch_to_files(self, batch, names, writer, seed=None):
        """Batched form of the reference's test step + after_infer (tasks/StyleSinger/stylesinger.py:186-275), which
        is limited to batch size 1: run the batch, vocode it, quantise to PCM16 on the device, crop each item to its own
        frame count and queue the files on `writer` (a writer.WavWriter)."""
        from .writer import wav_to_pcm16
        res = self.infer_batch(batch, seed=seed)
        hop = self.vocoder.model.hop
        pcm = wav_to_pcm16(res["wav"], res["lens"], hop, norm=bool(self.hparams.get("out_wav_norm", False)))
        writer.submit_batch(names, pcm, res["lens"], hop)
        return res

    # ---- the reference's single-utterance surface ---------------------------------------------
    def input_to_batch(self, item):
        """inference/StyleSinger.py:139-172: `item['f0']` is the tracker's contour in Hz (0 = unvoiced) and goes through
        `norm_interp_f0` (utils/pitch_utils.py:47-62) exactly as there (:152)."""
        from .pitch import norm_interp_f0
        d = self.device
        t = lambda x, dt: torch.as_tensor(np.asarray(x), dtype=dt)[None].to(d)
        f0, _uv = norm_interp_f0(np.asarray(item["f0"]), self.hparams)
        return dict(txt_tokens=t(item["ph_token"], torch.long), ref_mels=t(item["mel"], torch.float32),
                    spk_embed=t(item["spk_embed"], torch.float32), emo_embed=t(item["emo_embed"], torch.float32),
                    note=t(item["note"], torch.long), note_dur=t(item["note_dur"], torch.float32),
                    note_type=t(item["note_type"], torch.long), ref_f0=f0[None].to(d),
                    **({"mel2ph": t(item["mel2ph"], torch.long)} if "mel2ph" in item else {}))

    def forward_model(self, inp, noise=None, vocoder_noise=None):
        sample = self.input_to_batch(inp)
        res = self.infer_batch(sample, noise=noise, vocoder_noise=None, vocode=False)
        mel_pred = res["mel"].cpu().numpy()
        f0_pred = res["f0"].cpu().numpy()
        mask = np.abs(mel_pred).sum(-1) > 0
        mel_pred = np.clip(mel_pred[mask], self.hparams["mel_vmin"], self.hparams["mel_vmax"])
        f0_pred = f0_pred[mask]
        return self.vocoder.spec2wav(mel_pred, f0=f0_pred, noise=vocoder_noise)

    def infer_once(self, inp):
        return self.forward_model(inp)

"""Inference entrypoint of the HIP path: mirror of `inference/StyleSinger.py::StyleSingerInfer`.

`StyleSingerInfer(hparams).forward_model(inp)` keeps the reference's single-utterance contract
(inference/StyleSinger.py:41-63: run the model, drop all-zero frames, clip the mel to
[mel_vmin, mel_vmax], vocode with the predicted f0).  `infer_batch` is the batched, device-resident
form the benchmark and the data-parallel driver use.  The feature extractors of `preprocess_input`
(:94-137) run on the device: `preprocess_batch` computes the reference mel, the emotion embedding, the speaker
embedding (resemblyzer's published algorithm on the emotion encoder's kernels, `speaker.py`; parity unpinned: un-vendored package),
the f0 contour (Praat's autocorrelation method, `f0track.py`; parity unpinned: parselmouth is un-vendored) and its normalisation.
`trim_long_silences` runs on the device AROUND webrtcvad's per-window decisions (a fixed-point GMM whose tables are not in the reference tree):
they are computed on the host when the package is importable, or given by the caller; skipping the trim is an explicit opt-out
(`vad_flags=False`), never a silent default.
`infer_once(inp)` = the reference's entry point (inference/StyleSinger.py:175-179) with the features kept on the device between the producers and
the model; `python -m stylesinger_amd.infer` = `example_run` (:181-331).
"""
import json
import os
import warnings

import numpy as np
import torch

from . import lib as L
from .config import make_hparams, make_vocoder_config
from .model import StyleSingerHIP
from .vocoder import get_vocoder_cls


class StyleSingerInfer:
    def __init__(self, hparams=None, device=None, model_state=None, vocoder_state=None, vocoder_config=None, dictionary=None,
                 emotion_state=None, speaker_state=None, phone_set=None):
        """`emotion_state`: state_dict of the reference's emotion encoder checkpoint (`EmotionEncoder.load_model`,
        inference/StyleSinger.py:101) - enables the emotion branch of `preprocess_batch`.
        `speaker_state`: `model_state` of resemblyzer's `pretrained.pt` (`VoiceEncoder()`, inference/StyleSinger.py:100) - enables the
        speaker branch."""
        self.hparams = make_hparams(hparams)
        self._front_hparams = hparams
        # inference/StyleSinger.py:27-28: ph_encoder = build_token_encoder(f"{processed_data_dir}/phone_set.json"); `phone_set` overrides the path
        # (the released checkpoint ships it as ZH_checkpoint_phone_set.json). Without one the caller passes inp['ph_token'] or sets self.ph_encoder.
        self.ph_encoder = None
        if phone_set is None and hparams and hparams.get("processed_data_dir"):
            cand = os.path.join(str(hparams["processed_data_dir"]), "phone_set.json")
            phone_set = cand if os.path.exists(cand) else None
        if phone_set is not None:
            from .text_encoder import build_token_encoder
            self.ph_encoder = build_token_encoder(phone_set)
            if dictionary is None:
                dictionary = self.ph_encoder
        if hparams and hparams.get("loud_norm"):
            # process_audio passes loud_norm to librosa_wav2spec (inference/StyleSinger.py:85; utils/audios/__init__.py:55-59: pyloudnorm, un-vendored)
            raise NotImplementedError("hparams['loud_norm'] is set: the pyloudnorm loudness normalisation of the reference mel is not implemented "
                                      "(the released config leaves it off); refusing rather than computing a different mel")
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
        self._mel_frontend = None
        self._emo_frontend = None
        self.emotion_encoder = None
        if emotion_state is not None:
            from .emotion import EmotionEncoderHIP
            self.emotion_encoder = EmotionEncoderHIP(emotion_state, device=self.device)
        self.speaker_encoder = None
        if speaker_state is not None:
            from .speaker import SpeakerEncoderHIP
            self.speaker_encoder = SpeakerEncoderHIP(speaker_state, device=self.device)

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
            if self.model.f16:   # the fp16 modes' range check, where this path waits for the batch anyway
                strm.synchronize()
                self.model.check_finite(res["model_out"])
            # the results were allocated from the side stream's pool and are handed to the caller's stream: tell the allocator, or
            # the next batch on that side stream could reuse the memory while `main` still reads it
            for v in list(res.values()) + list(res.get("model_out", {}).values()):
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            return res
        # `batches` may be a lazy producer (a dataset loop placing batches on the GPU): it is consumed one batch at a time, so at most
        # `in_flight` batches are resident. Lazily built shared state is created here, on the caller's stream, not inside a forward on a
        # side stream: a default size up front, and a batch longer than that grows it when the batch is pulled (the tables only grow).
        warm = 2048
        self.model.warm_caches(warm, self.device)
        for i, batch in enumerate(batches):
            need = max(int(batch["mel2ph"].shape[1]) if batch.get("mel2ph") is not None else 0, int(batch["ref_mels"].shape[1]))
            if need > warm:
                warm = need
                self.model.warm_caches(warm, self.device)
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
        self.model.check_finite(res["model_out"])
        hop = self.vocoder.model.hop
        pcm = wav_to_pcm16(res["wav"], res["lens"], hop, norm=bool(self.hparams.get("out_wav_norm", False)))
        writer.submit_batch(names, pcm, res["lens"], hop)
        return res

    # ---- input producers on the device ------------------------------------------
    @staticmethod
    def align_f0_to_mel(f0, n_mel, hop_size=256):
        """The tracker-output alignment of preprocess_input (inference/StyleSinger.py:120-136): left pad 2 * pad_size frames, right
        pad to the mel length, repeat the last value / crop when still off (|delta| <= 8 asserted there). numpy in, numpy out."""
        pad_size = {128: 4, 256: 2}[hop_size]
        f0 = np.asarray(f0)
        lpad = pad_size * 2
        rpad = n_mel - len(f0) - lpad
        f0 = np.pad(f0, [[lpad, max(rpad, 0)]], mode="constant") if rpad >= 0 else np.pad(f0, [[lpad, 0]], mode="constant")
        delta = n_mel - len(f0)
        assert abs(delta) <= 8, delta
        if delta > 0:
            f0 = np.concatenate([f0, [f0[-1]] * delta], 0)
        return f0[:n_mel]

    def _partials_batch(self, wavs, lens, slicer):
        """Shared front half of the two utterance encoders: per item zero-pad to the end of its last partial window, 40-mel power
        spectrogram of the whole batch (EmotionMelFrontendHIP: the same librosa.feature.melspectrogram parameters in both packages),
        gather the 160-frame partial windows of ALL items. -> (frames [sum P_b, 160, 40], counts [P_b])"""
        from .frontend import EmotionMelFrontendHIP
        if self._emo_frontend is None:
            self._emo_frontend = EmotionMelFrontendHIP(self.device)
        lens = [int(v) for v in lens]
        B = len(lens)
        slices = [slicer(n) for n in lens]
        need = [max(n, ws[-1].stop) for n, (ws, _) in zip(lens, slices)]   # `if max_wave_length >= len(wav): pad` (inference.py:129-131)
        buf = torch.zeros(B, max(need), device=self.device, dtype=torch.float32)
        buf[:, :wavs.shape[1]] = wavs.to(self.device).float()[:, :max(need)]
        mel40, _ = self._emo_frontend.wav2mel(buf, need)
        idx_b, idx_t = [], []
        counts = []
        for b, (_, ms) in enumerate(slices):
            counts.append(len(ms))
            for sl in ms:
                idx_b.append(torch.full((sl.stop - sl.start,), b, dtype=torch.long))
                idx_t.append(torch.arange(sl.start, sl.stop, dtype=torch.long))
        ib = torch.cat(idx_b).to(self.device)
        it = torch.cat(idx_t).to(self.device)
        frames = mel40[ib, it].reshape(sum(counts), 160, mel40.shape[-1]).contiguous()
        return frames, counts

    def _mean_l2norm_per_item(self, part, counts):
        out = torch.empty(len(counts), part.shape[1], device=self.device, dtype=torch.float32)
        lib, o = L.load(), 0
        for b, c in enumerate(counts):
            L.check(lib.ss_mean_l2norm(L.ptr(part[o:o + c]), L.ptr(out[b]), c, part.shape[1], L.stream_ptr()), "ss_mean_l2norm")
            o += c
        return out

    @torch.no_grad()
    def embed_emotion_batch(self, wavs, lens):
        """`Embed_utterance(wav, using_partials=True)` (data_gen/tts/emotion/inference.py:111-151) for a batch of PREPROCESSED
        waveforms (`preprocess_wav` output, zero beyond lens[b]; lens are host ints): per item zero-pad to the last partial's end,
        40-mel power spectrogram (EmotionMelFrontendHIP), the partial windows of ALL items through the LSTM in one pass, mean + L2
        norm per item. -> [B, 256] on the device."""
        from .emotion import compute_partial_slices
        if self.emotion_encoder is None:
            raise L.StyleSingerHipError("embed_emotion_batch: construct StyleSingerInfer(..., emotion_state=<emotion encoder state_dict>)")
        frames, counts = self._partials_batch(wavs, lens, compute_partial_slices)
        return self._mean_l2norm_per_item(self.emotion_encoder.embed_frames_batch(frames), counts)

    @torch.no_grad()
    def embed_speaker_batch(self, wavs, lens, rate=1.3, min_coverage=0.75):
        """`VoiceEncoder().embed_utterance(wav)` (inference/StyleSinger.py:100,104; resemblyzer 0.1.1.dev0, un-vendored: parity UNPINNED,
        `speaker.py`) for a batch of waveforms [B, L] (zero beyond lens[b]): partial windows of 160 frames every round(16000 / rate / 160)
        frames of the 40-mel, VoiceEncoder.forward on all of them in one pass (3 x LSTM, ReLU(Linear), L2 norm per partial), L2-normalised
        mean per item. The reference hands it the 48 kHz samples of `process_audio` rounded to float16 (:87,104) and the package reads
        them as 16 kHz audio - `preprocess_batch` reproduces exactly that. -> [B, 256] on the device."""
        from .speaker import compute_partial_slices as spk_slices
        if self.speaker_encoder is None:
            raise L.StyleSingerHipError("embed_speaker_batch: construct StyleSingerInfer(..., speaker_state=<resemblyzer model_state>)")
        frames, counts = self._partials_batch(wavs, lens, lambda n: spk_slices(n, rate, min_coverage))
        return self._mean_l2norm_per_item(self.speaker_encoder.forward(frames), counts)

    def process_audio_wav(self, ref_wavs, frames, valid_lens=None):
        """The waveform `process_audio` returns next to the mel (inference/StyleSinger.py:86-88): the audio zero-padded to
        n_mel * hop samples (utils/audios/__init__.py:76-78) and rounded to float16. -> ([B, max n_mel * hop] fp32 holding
        float16-representable values, zero beyond each item's length; lengths as host ints). `valid_lens`: the items' own sample
        counts - samples of the batch buffer past them are padding whatever they hold (as MelFrontendHIP.wav2mel treats them)."""
        hop = int(self.hparams["hop_size"])
        lens = [int(f) * hop for f in frames]
        x = ref_wavs.to(self.device).float().contiguous()
        out = torch.empty(x.shape[0], max(lens), device=self.device, dtype=torch.float32)
        n_out = torch.tensor(lens, dtype=torch.int32).to(self.device)
        n_in = None if valid_lens is None else torch.tensor([int(v) for v in valid_lens], dtype=torch.int32).to(self.device)
        L.check(L.load().ss_round_f16_rows(L.ptr(x), x.shape[1], x.shape[1], L.ptr(n_in), L.ptr(n_out), L.ptr(out), out.shape[1], x.shape[0], L.stream_ptr()),
                "ss_round_f16_rows")
        for t_ in (n_out, n_in):
            if t_ is not None:
                t_.record_stream(torch.cuda.current_stream(self.device))
        return out, lens

    @torch.no_grad()
    def preprocess_batch(self, ref_wavs, ref_lens, spk_embed, f0_hz, txt_tokens, note, note_dur, note_type, mel2ph=None,
                         emo_embed=None, emo_wavs=None, emo_lens=None, emo_vad_flags=None):
        """Batched device form of `preprocess_input` + `input_to_batch` (inference/StyleSinger.py:94-172): from reference audio to
        the dict `infer_batch` takes, with no host round trip of the data.
          ref_wavs [B, L] fp32 48 kHz reference audio (zero beyond ref_lens[b]; ref_lens host ints)   -> ref_mels  (process_audio, :106-118)
          f0_hz    [B, Tr] tracker contour in Hz aligned to the mel frames (align_f0_to_mel), 0 = unvoiced -> ref_f0 (norm_interp_f0, :152);
                   None -> tracked on the device from `process_audio`'s waveform as :112-135 does with parselmouth (`f0track.py`: Praat's
                   published autocorrelation method, 80-800 Hz, voicing threshold 0.6; parity UNPINNED - parselmouth is un-vendored)
          emo_wavs [B, Le] `preprocess_wav` output for the emotion encoder (zero beyond emo_lens[b])  -> emo_embed (Embed_utterance, :104)
                   default: the reference audio itself, volume-normalised on the device. `trim_long_silences` (audio.py:58-100) runs on the
                   device AROUND the caller's decisions: pass `emo_vad_flags` [B, nW] = webrtcvad's `is_speech` per 30 ms window of the
                   volume-normalised 16-bit PCM (the decision itself is an un-vendored fixed-point GMM: `vadtrim.py`); without flags the
                   audio goes untrimmed; `emo_vad_flags="webrtc"` computes them on the host with the webrtcvad package from the device-normalised
                   audio (`vadtrim.webrtc_flags`: the reference's own call). Pass `emo_embed` [B, 256] instead to skip this branch.
        The returned dict also carries `ref_f0_hz` [B, Tr] (the tracker's contour on the mel grid, before normalisation) for callers that mirror
        `preprocess_input`'s `inp['f0']`.
          spk_embed [B, 256], or None -> `VoiceEncoder().embed_utterance(wav)` (:100,104) on the device (`embed_speaker_batch`;
                   needs `speaker_state`) from what the reference hands it: `process_audio`'s waveform, i.e. the reference audio
                   zero-padded to n_mel * hop samples and rounded to float16 (:87; utils/audios/__init__.py:76-78)."""
        from .frontend import MelFrontendHIP
        from .pitch import norm_interp_f0_device
        d = self.device
        if self._mel_frontend is None:
            self._mel_frontend = MelFrontendHIP(self._front_hparams, device=d)
        ref_lens_h = [int(v) for v in ref_lens]
        ref_wavs = ref_wavs.to(d).float()
        ref_mels, frames = self._mel_frontend.wav2mel(ref_wavs, torch.tensor(ref_lens_h, dtype=torch.int64))
        Tr = ref_mels.shape[1]
        hop = int(self.hparams["hop_size"])
        wav16 = None
        if f0_hz is None or spk_embed is None:   # the waveform the reference hands both third-party producers (:87)
            wav16, wav16_lens = self.process_audio_wav(ref_wavs, [n // hop + 1 for n in ref_lens_h], ref_lens_h)   # frames of a centred STFT
        if f0_hz is None:
            from .f0track import track_f0_device
            f0_hz = track_f0_device(wav16, wav16_lens, Tr, sr=int(self.hparams["audio_sample_rate"]), hop_size=hop)
        f0_hz = f0_hz.to(d).float()
        if f0_hz.shape[1] != Tr:
            raise ValueError(f"preprocess_batch: f0_hz has {f0_hz.shape[1]} frames, the reference mel {Tr} (use align_f0_to_mel)")
        ref_f0, _uv = norm_interp_f0_device(f0_hz, frames, self.hparams)
        if emo_embed is None:
            if emo_wavs is None:
                if self._emo_frontend is None:
                    from .frontend import EmotionMelFrontendHIP
                    self._emo_frontend = EmotionMelFrontendHIP(d)
                emo_wavs = self._emo_frontend.normalize_volume(ref_wavs, torch.tensor(ref_lens_h))
                emo_lens = ref_lens_h
                if isinstance(emo_vad_flags, str):
                    if emo_vad_flags != "webrtc":
                        raise ValueError(f"emo_vad_flags={emo_vad_flags!r}: expected flags, None or 'webrtc'")
                    from .vadtrim import webrtc_flags
                    emo_vad_flags = webrtc_flags(emo_wavs, emo_lens)
                if emo_vad_flags is not None:   # preprocess_wav's second step (audio.py:38), around the VAD flags
                    from .vadtrim import trim_long_silences_device
                    emo_wavs, kept = trim_long_silences_device(emo_wavs, emo_lens, emo_vad_flags)
                    emo_lens = [int(v) for v in kept.cpu()]   # the partial slicing below is host arithmetic on the lengths
            emo_embed = self.embed_emotion_batch(emo_wavs, emo_lens)
        if spk_embed is None:
            spk_embed = self.embed_speaker_batch(wav16, wav16_lens)
        batch = dict(txt_tokens=txt_tokens.to(d), note=note.to(d), note_dur=note_dur.to(d).float(), note_type=note_type.to(d),
                     spk_embed=spk_embed.to(d).float(), emo_embed=emo_embed.to(d).float(), ref_mels=ref_mels, ref_f0=ref_f0, ref_f0_hz=f0_hz)
        if mel2ph is not None:
            batch["mel2ph"] = mel2ph.to(d)
        return batch

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

    def _wav_from_result(self, res, vocoder_noise=None):
        """inference/StyleSinger.py:53-63: drop all-zero frames, clip the mel, vocode with the predicted f0 (one item)."""
        mel_pred = res["mel"].cpu().numpy()
        self.model.check_finite(res["model_out"])   # (the copy above synchronised)
        f0_pred = res["f0"].cpu().numpy()
        mask = np.abs(mel_pred).sum(-1) > 0
        mel_pred = np.clip(mel_pred[mask], self.hparams["mel_vmin"], self.hparams["mel_vmax"])
        f0_pred = f0_pred[mask]
        return self.vocoder.spec2wav(mel_pred, f0=f0_pred, noise=vocoder_noise)

    def forward_model(self, inp, noise=None, vocoder_noise=None):
        sample = self.input_to_batch(inp)
        return self._wav_from_result(self.infer_batch(sample, noise=noise, vocoder_noise=None, vocode=False), vocoder_noise)

    @staticmethod
    def _load_wav(path, want_sr):
        """A reference-audio FILE: 16-bit PCM WAV at the model's sample rate (the reference resamples through librosa, un-vendored: other rates
        are refused, not approximated). -> float32 mono in [-1, 1)."""
        import wave
        with wave.open(os.fsdecode(path), "rb") as wf:
            if wf.getsampwidth() != 2 or wf.getframerate() != want_sr:
                raise ValueError(f"{path}: need 16-bit PCM at {want_sr} Hz (got {8 * wf.getsampwidth()} bit, {wf.getframerate()} Hz)")
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32).reshape(-1, wf.getnchannels())
        return pcm.mean(axis=1) / 32768.0

    _warned_untrimmed = False

    def _resolve_vad(self, vad_flags):
        """`preprocess_wav` ALWAYS trims long silences (data_gen/tts/emotion/audio.py:36-38). None = do as the reference does: webrtcvad's decisions,
        computed on the host - an ImportError where the package is missing (it is un-vendored), never a silent skip. False = explicit opt-out
        (untrimmed audio; warns once: the emotion embedding of a recording with long pauses then differs from the reference's). Otherwise the
        caller's flags [nW]."""
        if vad_flags is None:
            from .vadtrim import have_webrtcvad
            if not have_webrtcvad():
                raise ImportError("preprocess_input: the reference trims long silences with webrtcvad before the emotion encoder, and the package is "
                                  "not importable here. Pass vad_flags=<webrtcvad's is_speech per 30 ms window> or vad_flags=False to skip the trim "
                                  "explicitly (the emotion embedding then differs from the reference's for audio with long pauses).")
            return "webrtc"
        if vad_flags is False:
            if not StyleSingerInfer._warned_untrimmed:
                StyleSingerInfer._warned_untrimmed = True
                warnings.warn("StyleSingerInfer: trim_long_silences skipped on request (vad_flags=False): emo_embed is computed from untrimmed audio")
            return None
        return np.asarray(vad_flags)[None]

    @torch.no_grad()
    def _device_batch(self, inp, vad_flags=None):
        """`preprocess_input` + `input_to_batch` (inference/StyleSinger.py:94-172) for ONE item with every producer on the device: the dict
        `infer_batch` takes, device tensors only (+ `ref_f0_hz`, `n_mel`). ONE pass of the f0 tracker."""
        sr, hop = int(self.hparams["audio_sample_rate"]), int(self.hparams["hop_size"])
        audio = inp["ref_audio"]
        wav = self._load_wav(audio, sr) if isinstance(audio, (str, bytes)) or hasattr(audio, "__fspath__") else np.asarray(audio, dtype=np.float32)
        if "ph_token" not in inp:
            if self.ph_encoder is None:
                raise ValueError("preprocess_input: give inp['ph_token'], construct StyleSingerInfer(..., phone_set=<phone_set.json>) or set "
                                 "self.ph_encoder (the reference's build_token_encoder(f'{processed_data_dir}/phone_set.json'))")
            inp["ph_token"] = self.ph_encoder.encode(" ".join(inp["ph"]))
        t = lambda x, dt: torch.as_tensor(np.asarray(x), dtype=dt)[None]
        batch = self.preprocess_batch(torch.from_numpy(wav)[None], [len(wav)], None, None, t(inp["ph_token"], torch.long), t(inp["note"], torch.long),
                                      t(inp["note_dur"], torch.float32), t(inp["note_type"], torch.long),
                                      mel2ph=t(inp["mel2ph"], torch.long) if "mel2ph" in inp else None, emo_vad_flags=self._resolve_vad(vad_flags))
        batch["n_mel"] = len(wav) // hop + 1
        return batch

    @torch.no_grad()
    def preprocess_input(self, inp, vad_flags=None):
        """Mirror of `StyleSingerInfer.preprocess_input` (inference/StyleSinger.py:94-137) with every producer on the device: fills `mel`,
        `spk_embed`, `emo_embed`, `f0` (the tracker's contour in Hz on the mel grid) as numpy arrays, `ph_token`, and `item_name` / `wav_fn` from
        `inp['ref_audio']` (a float waveform at the model's sample rate, or the path of a 16-bit PCM WAV at that rate). Needs `emotion_state` and
        `speaker_state` (the two encoders' checkpoints). `vad_flags`: see `_resolve_vad` (None = webrtcvad on the host, False = opt out)."""
        batch = self._device_batch(inp, vad_flags)
        n_mel, audio = batch["n_mel"], inp["ref_audio"]
        inp.update(item_name=inp.get("name"), wav_fn=os.fsdecode(audio) if isinstance(audio, (str, bytes)) or hasattr(audio, "__fspath__") else None,
                   mel=batch["ref_mels"][0, :n_mel].cpu().numpy(), spk_embed=batch["spk_embed"][0].cpu().numpy(),
                   emo_embed=batch["emo_embed"][0].cpu().numpy(), f0=batch["ref_f0_hz"][0, :n_mel].double().cpu().numpy())
        return inp

    def postprocess_output(self, output):
        return output

    def infer_once(self, inp, vad_flags=None, noise=None, vocoder_noise=None):
        """inference/StyleSinger.py:175-179: preprocess_input -> forward_model -> postprocess_output. The features stay on the device between the
        producers and the model (no numpy detour; `preprocess_input` is the form that returns them). An `inp` that already carries the features
        (`mel`, `spk_embed`, `emo_embed`, `f0`, `ph_token`) skips the producers, as before."""
        if all(k in inp for k in ("mel", "spk_embed", "emo_embed", "f0", "ph_token")):
            return self.postprocess_output(self.forward_model(inp, noise=noise, vocoder_noise=vocoder_noise))
        batch = self._device_batch(inp, vad_flags)
        res = self.infer_batch({k: v for k, v in batch.items() if k not in ("n_mel", "ref_f0_hz")}, noise=noise, vocode=False)
        return self.postprocess_output(self._wav_from_result(res, vocoder_noise))

    @classmethod
    def example_run(cls, hparams=None, ref_audio="test/test.wav", out_path="infer_out/test.wav", vad_flags=None, **ctor):
        """inference/StyleSinger.py:181-331: the example score (stylesinger_amd/example_input.json = that method's input dict, extracted by
        `python -m oracle.gen_golden --round6`) sung in the style of `ref_audio`, written to `out_path` as 16-bit PCM (utils/audio.py:12-17).
        `ctor`: how to build the instance - `exp_dir` + `vocoder_dir` (the reference's checkpoints, `from_checkpoints`) or explicit
        `model_state` / `vocoder_state` / `emotion_state` / `speaker_state` / `phone_set`."""
        from .writer import save_wav
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "example_input.json")) as fh:
            inp = {k: v for k, v in json.load(fh).items() if k != "source"}
        inp["ref_audio"] = ref_audio
        if "exp_dir" in ctor:
            ins = cls.from_checkpoints(hparams, ctor.pop("exp_dir"), ctor.pop("vocoder_dir"), **ctor)
        else:
            ins = cls(hparams, **ctor)
        out = ins.infer_once(inp, vad_flags=vad_flags)
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        save_wav(out, out_path, int(ins.hparams["audio_sample_rate"]), norm=bool(ins.hparams.get("out_wav_norm", False)))
        print(f"Save at {out_path}.")
        return out


def main(argv=None):
    """`python -m stylesinger_amd.infer --exp-dir checkpoints/<exp> --vocoder-dir <hifigan dir> --emotion-ckpt <pt> --speaker-ckpt <pt>
    --phone-set ZH_checkpoint_phone_set.json [--ref-audio test/test.wav] [--out infer_out/test.wav] [--no-vad-trim]` = the reference's
    `python inference/StyleSinger.py` (StyleSingerInfer.example_run)."""
    import argparse
    ap = argparse.ArgumentParser(description="StyleSinger example_run on the HIP path")
    ap.add_argument("--exp-dir", required=True)
    ap.add_argument("--vocoder-dir", required=True)
    ap.add_argument("--emotion-ckpt", required=True, help="the reference's emotion encoder checkpoint (hparams['emotion_encoder_path'])")
    ap.add_argument("--speaker-ckpt", required=True, help="resemblyzer's pretrained.pt")
    ap.add_argument("--phone-set", required=True)
    ap.add_argument("--ref-audio", default="test/test.wav")
    ap.add_argument("--out", default="infer_out/test.wav")
    ap.add_argument("--no-vad-trim", action="store_true", help="explicit opt-out of trim_long_silences (webrtcvad missing)")
    a = ap.parse_args(argv)
    emo = torch.load(a.emotion_ckpt, map_location="cpu", weights_only=False)
    spk = torch.load(a.speaker_ckpt, map_location="cpu", weights_only=False)
    from . import ckpt
    state, _ = ckpt.read_state(a.exp_dir, "model")
    if state is None:
        raise FileNotFoundError(f"| ckpt not found in {a.exp_dir}.")
    vstate, vcfg = ckpt.load_vocoder_ckpt(a.vocoder_dir)
    StyleSingerInfer.example_run(None, a.ref_audio, a.out, vad_flags=False if a.no_vad_trim else None, model_state=state, vocoder_state=vstate,
                                 vocoder_config=vcfg, emotion_state=emo.get("model_state", emo), speaker_state=spk.get("model_state", spk),
                                 phone_set=a.phone_set)


if __name__ == "__main__":
    main()

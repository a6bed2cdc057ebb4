"""Speaker encoder (`resemblyzer.VoiceEncoder().embed_utterance`, inference/StyleSinger.py:100-104) on the kernels of the emotion
encoder, against the oracle's restatement of the package's published algorithm (parity UNPINNED: the package is an un-vendored dependency).
Every launch is one tests/test_gpu_round2.py::test_emotion_encoder_matches_reference_golden already exercises, only the host-side slicing /
composition is new. First run on an MI355X in round 5 (profiles/r05_session1_tests.log: partial embeds 6.7e-8, utterance embed 2.2e-8)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402
from stylesinger_amd.speaker import SpeakerEncoderHIP, compute_partial_slices  # noqa: E402


def test_speaker_encoder_matches_the_oracle_restatement():
    ssd = synth.synth_emotion_state_dict(11)              # same parameter names and shapes as resemblyzer's model_state (lstm.*, linear.*)
    enc = SpeakerEncoderHIP(ssd, device="cuda:0")
    n_samples = 16000 * 5 + 3000
    _, sl = compute_partial_slices(n_samples)
    mel = synth.synth_emotion_frames(1, n_frames=sl[-1].stop, seed=12)[0]
    got = enc.embed_utterance_frames(mel, n_samples=n_samples).cpu()
    frames = torch.stack([torch.as_tensor(mel)[s] for s in sl]).float()
    with torch.no_grad():
        want, part = R.speaker_embed(ssd, frames)
    e_p = (enc.forward(frames).cpu() - part).abs().max().item()
    e_e = (got - want).abs().max().item()
    print(f"speaker encoder: {len(sl)} partials, partial embeds max err {e_p:.3e}, utterance embed {e_e:.3e}")
    assert abs(float(got.norm()) - 1.0) <= 1e-5
    assert e_p <= 1e-5 and e_e <= 1e-5

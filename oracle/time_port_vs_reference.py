"""How conservative is bench.py's `cpu_baseline` (kind "port")? Times the REAL reference (imported unmodified from /root/reference) and the
restatement (oracle/restatement.py) on the same inputs and threads: BASELINE config C1 - one 4 s utterance (T = 750 frames), 100 mel + 2 x 100
f0 diffusion steps + HiFi-GAN-NSF. Runs in the build container only (the reference is not on the GPU box). TEST INFRASTRUCTURE.

    python -m oracle.time_port_vs_reference [--threads 8] [--repeat 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gen_golden, refimport, restatement as R  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--frames", type=int, default=750)
    a = ap.parse_args()
    assert refimport.available(), "the reference is not mounted here"
    torch.set_num_threads(a.threads)
    S = 100
    model, hp, sd = gen_golden.build_reference(S, S, 1234)
    T, Tp, Tr = a.frames, max(2, a.frames * 28 // 1500), min(a.frames, 1500)
    batch = synth.synth_batch(1, T, Tp, Tr, hp, 1234)
    ref_cls = refimport.load()["HifiGanGenerator"]
    vcfg = config.make_vocoder_config()
    gen = ref_cls(vcfg)
    vsd = synth.synth_vocoder_state_dict(vcfg, 1234)
    gen.load_state_dict(vsd, strict=True)
    gen.remove_weight_norm()
    gen.eval()

    def run_reference():
        with torch.no_grad():
            ret = model(batch["txt_tokens"], mel2ph=batch["mel2ph"], spk_embed=batch["spk_embed"], emo_embed=batch["emo_embed"],
                        ref_mels=batch["ref_mels"], ref_f0=batch["ref_f0"], global_steps=320000, infer=True, note=batch["note"],
                        note_dur=batch["note_dur"], note_type=batch["note_type"])
            mel = ret["mel_out"].clamp(hp["mel_vmin"], hp["mel_vmax"])
            return gen(mel.transpose(1, 2), ret["f0_denorm"])

    def run_port():
        with torch.no_grad():
            ret = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(7), mel2ph=batch["mel2ph"])
            mel = ret["mel_out"].clamp(hp["mel_vmin"], hp["mel_vmax"])
            return R.hifigan_forward(vsd, vcfg, mel, ret["f0_denorm"], synth.NoiseTape(8))[0]

    out = {}
    for name, fn in (("reference", run_reference), ("port", run_port)):
        fn()   # warm-up (allocator, oneDNN primitives)
        ts = []
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = ts[len(ts) // 2]
        print(f"{name:10s} median of {a.repeat}: {out[name]:.2f} s = {T / out[name]:.1f} mel-frames/s ({a.threads} threads)", flush=True)
    print(f"port / reference speed ratio: {out['reference'] / out['port']:.2f}x (> 1: the port is the FASTER of the two, i.e. GPU / cpu_baseline ratios are conservative)")


if __name__ == "__main__":
    main()

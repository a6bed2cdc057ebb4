"""Shared helpers for the parity tests (TEST INFRASTRUCTURE ONLY): rebuild a golden case's inputs and
run the CPU restatement on them."""
import os

import torch

from stylesinger_amd import config, synth
from . import restatement as R

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_case(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def case_setup(meta):
    hp = config.make_hparams(dict(timesteps=meta["steps_mel"], K_step=meta["steps_mel"], f0_timesteps=meta["steps_f0"],
                                  **meta.get("hp_over", {})))
    sd = synth.synth_acoustic_state_dict(hp, meta["seed"])
    batch = synth.synth_batch(meta["B"], meta["T"], meta["Tp"], meta["Tr"], hp, meta["seed"])
    return hp, sd, batch


def run_restatement_case(meta):
    hp, sd, batch = case_setup(meta)
    tape = synth.NoiseTape(meta["tape_seed"])
    with torch.no_grad():
        ret = R.acoustic_forward(sd, hp, batch, tape, mel2ph=batch["mel2ph"] if meta["give_mel2ph"] else None)
    return ret, tape


def vocoder_case_setup(meta):
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, meta["seed"])
    return cfg, vsd

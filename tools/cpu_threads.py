"""How does the CPU oracle scale with threads on this host? (picks cpu_baseline's thread count)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restatement as R
from stylesinger_amd import config, synth
hp = config.make_hparams(dict(timesteps=10, K_step=10, f0_timesteps=10))
sd = synth.synth_acoustic_state_dict(hp, 1234)
T = 375
batch = synth.synth_batch(1, T, 7, T, hp, 1234)
print("cpus", os.cpu_count())
for n in (8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 8):
        break
    torch.set_num_threads(n)
    with torch.no_grad():
        R.acoustic_forward(sd, hp, batch, synth.NoiseTape(1), mel2ph=batch["mel2ph"])
        t0 = time.time()
        R.acoustic_forward(sd, hp, batch, synth.NoiseTape(1), mel2ph=batch["mel2ph"])
        dt = time.time() - t0
    print(f"threads={n:4d}  {dt:.2f} s for 10-step T={T}")

"""Checkpoint intake: the reference's on-disk formats -> the HIP path's modules.

Mirrors the loading rules of the reference so its released checkpoints drop in unchanged:
  * acoustic model: `utils/commons/ckpt_utils.py:26-67 load_ckpt` - newest `model_ckpt_steps_*.ckpt` of an experiment
    directory (or an explicit file); `checkpoint["state_dict"]` is either flat (`"model.<param>"` keys) or nested
    (`{"model": {<param>: tensor}}`); `strict=False` drops shape-mismatched entries before loading.
  * vocoder: `tasks/tts/vocoder_infer/hifigan_nsf.py:24-61` - `config.yaml` + `state_dict.model_gen`, or `config.json`
    + `generator` (the original HiFi-GAN layout, checkpoint file `generator_v1`).
Parameter names/shapes are the contract; weight-norm stays in the (g, v) parametrisation on disk and is
folded on the device at pack time.  `python -m stylesinger_amd.ckpt strip <in.ckpt> <out.pt>` writes an inference-only
copy (optimizer states dropped).
"""
import glob
import json
import os
import re
import sys

import torch


def get_all_ckpts(work_dir, steps=None):
    pat = f"{work_dir}/model_ckpt_steps_{'*' if steps is None else steps}.ckpt"
    return sorted(glob.glob(pat), key=lambda p: -int(re.findall(r".*steps_(\d+)\.ckpt", p)[0]))


def _select(state_dict, model_name):
    """The sub-dict of `model_name` from either layout of checkpoint["state_dict"]."""
    if any("." in k for k in state_dict):
        pre = model_name + "."
        return {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
    if "." not in model_name:
        return dict(state_dict[model_name])
    base, rest = model_name.split(".", 1)
    pre = rest + "."
    return {k[len(pre):]: v for k, v in state_dict[base].items() if k.startswith(pre)}


def read_state(ckpt_base_dir, model_name="model"):
    """-> (state_dict of model_name, path) or (None, None) when the directory holds no checkpoint."""
    if os.path.isfile(ckpt_base_dir):
        path = ckpt_base_dir
    else:
        found = get_all_ckpts(ckpt_base_dir)
        if not found:
            return None, None
        path = found[0]
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    return _select(ckpt["state_dict"], model_name), path


def load_ckpt(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True):
    """Same contract as the reference's load_ckpt (ckpt_utils.py:26-67)."""
    state, path = read_state(ckpt_base_dir, model_name)
    if state is None:
        msg = f"| ckpt not found in {ckpt_base_dir}."
        if force:
            raise FileNotFoundError(msg)
        print(msg)
        return None
    if not strict:
        have = cur_model.state_dict()
        for k in [k for k, v in state.items() if k in have and have[k].shape != v.shape]:
            print("| Unmatched keys: ", k, tuple(have[k].shape), tuple(state[k].shape))
            del state[k]
    cur_model.load_state_dict(state, strict=strict)
    print(f"| load '{model_name}' from '{path}'.")
    return path


def load_vocoder_ckpt(base_dir):
    """-> (generator state_dict, config dict) from a reference vocoder directory (hifigan_nsf.py:24-61)."""
    if os.path.exists(f"{base_dir}/config.yaml"):
        import yaml
        cfg = yaml.safe_load(open(f"{base_dir}/config.yaml"))
        paths = sorted(glob.glob(f"{base_dir}/model_ckpt_steps_*.ckpt"), key=lambda p: int(re.findall(r"steps_(\d+)\.ckpt", p)[0]))
        if not paths:
            raise FileNotFoundError(f"no model_ckpt_steps_*.ckpt in {base_dir}")
        state = torch.load(paths[-1], map_location="cpu", weights_only=False)["state_dict"]["model_gen"]
    elif os.path.exists(f"{base_dir}/config.json"):
        cfg = json.load(open(f"{base_dir}/config.json"))
        state = torch.load(f"{base_dir}/generator_v1", map_location="cpu", weights_only=False)["generator"]
    else:
        raise FileNotFoundError(f"no config.yaml / config.json in {base_dir}")
    return state, cfg


def strip(src, dst, model_name="model"):
    """Inference-only copy of a training checkpoint: {"state_dict": {model_name: tensors}} and nothing else."""
    state, _ = read_state(src, model_name)
    if state is None:
        raise FileNotFoundError(src)
    torch.save({"state_dict": {model_name: {k: v.detach().clone().contiguous() for k, v in state.items()}}}, dst)
    return len(state)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "strip":
        n = strip(sys.argv[2], sys.argv[3], *(sys.argv[4:5]))
        print(f"wrote {n} tensors to {sys.argv[3]}")
    else:
        print(__doc__)

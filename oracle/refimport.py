"""Import the *real* reference (AaronZ345/StyleSinger, mounted read-only at /root/reference).

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` (in the build container, where
/root/reference exists) to pin the CPU restatement in ``oracle/restatement.py`` and to generate
the fixtures under ``tests/golden/``.  Nothing here runs on the GPU box and nothing in the product
(`stylesinger_amd/`) may import this module.

Recipe follows SURVEY.md §8(c): stub the absent third-party modules, chdir into the reference so the
cwd-relative yaml ``base_config`` chain resolves, call ``set_hparams`` *before* importing
``modules.StyleSinger.stylesinger`` (import-time default binding of ``max_beta``,
modules/diff/shallow_diffusion_tts.py:41).
"""
import os
import sys
import types

REF = os.environ.get("STYLESINGER_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "modules", "StyleSinger"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_loaded = {}


def load(overrides=None):
    """Returns dict(hparams=..., StyleSinger=cls, HifiGanGenerator=cls, torch=torch)."""
    if _loaded:
        if overrides:
            _loaded["hparams"].update(overrides)
        return _loaded
    assert available(), f"reference not found at {REF}"
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    for name in ("chardet", "pyloudnorm", "webrtcvad", "parselmouth", "resemblyzer"):
        _stub(name)
    lib = _stub("librosa")
    lib.filters = _stub("librosa.filters")
    lib.core = _stub("librosa.core")
    _stub("pycwt", wavelet=types.SimpleNamespace())
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    import tqdm as _tqdm
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from utils.hparams import set_hparams, hparams
        old_argv = sys.argv
        sys.argv = [old_argv[0]]
        try:
            set_hparams(config="egs/stylesinger.yaml", exp_name="", print_hparams=False)
        finally:
            sys.argv = old_argv
        if overrides:
            hparams.update(overrides)
        import modules.diff.shallow_diffusion_tts as sdt
        import modules.diff.gaussian_multinomial_diffusion as gmd
        # silence the progress bars (they are the reference's only "tracing")
        quiet = lambda it, **kw: it
        sdt.tqdm = quiet
        gmd.tqdm = quiet
        from modules.StyleSinger.stylesinger import StyleSinger
        from modules.hifigan.hifigan_nsf import HifiGanGenerator
    finally:
        os.chdir(cwd)
    import torch
    _loaded.update(hparams=hparams, StyleSinger=StyleSinger, HifiGanGenerator=HifiGanGenerator, torch=torch)
    return _loaded


class FakeDict:
    """Stands in for utils/text/text_encoder.py:TokenTextEncoder: the model only needs len() and pad()."""

    def __init__(self, n=61):
        self.n = n

    def __len__(self):
        return self.n

    def pad(self):
        return 0

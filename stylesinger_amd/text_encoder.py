"""Phoneme token encoder of the inference entry point: `build_token_encoder(phone_set.json)` / `TokenTextEncoder.encode`
(reference: utils/text/text_encoder.py:107-147,257-259, used at inference/StyleSinger.py:28,96 and as the model's `dictionary`,
modules/StyleSinger/stylesinger.py:46 -> `len(dictionary)` sizes the phoneme embedding).

Contract mirrored (pinned bit-exactly against the imported reference class in tests/test_host_cpu.py):
  * ids 0, 1, 2 are `<pad>`, `<EOS>`, `<UNK>`; the tokens of the list follow in list order from id 3 (reserved tokens inside the list are
    skipped, not renumbered);
  * `encode(s)` splits on whitespace, maps every token that is not in the vocabulary to `replace_oov` when one is set (a KeyError otherwise),
    and reverses the ids when `reverse` is set;
  * `decode(ids)` joins the tokens with single blanks, `ID_<n>` for an unknown id, optionally cut at the first pad / EOS.
Host-side integer work: nothing here touches the device.
"""
import json

PAD, EOS, UNK, SEG = "<pad>", "<EOS>", "<UNK>", "|"
RESERVED_TOKENS = [PAD, EOS, UNK]
NUM_RESERVED_TOKENS = len(RESERVED_TOKENS)
PAD_ID, EOS_ID, UNK_ID = 0, 1, 2


class TokenTextEncoder:
    def __init__(self, vocab_filename=None, reverse=False, vocab_list=None, replace_oov=None, num_reserved_ids=NUM_RESERVED_TOKENS):
        self._num_reserved_ids = num_reserved_ids
        self._reverse = bool(reverse)
        self._replace_oov = replace_oov
        if vocab_filename:   # a vocabulary FILE carries its reserved tokens itself: one token per line, ids = line numbers
            with open(vocab_filename) as fh:
                tokens = [line.strip() for line in fh.readlines()]
        else:
            if vocab_list is None:
                raise ValueError("TokenTextEncoder: give vocab_filename or vocab_list")
            tokens = RESERVED_TOKENS + [t for t in vocab_list if t not in RESERVED_TOKENS]
        self.id_to_token = dict(enumerate(tokens))
        self.token_to_id = {tok: i for i, tok in self.id_to_token.items()}   # a repeated token keeps its LAST id, as a dict built in id order does
        self.pad_index = self.token_to_id[PAD]
        self.eos_index = self.token_to_id[EOS]
        self.unk_index = self.token_to_id[UNK]
        self.seg_index = self.token_to_id.get(SEG, self.eos_index)

    # ---- the reference's surface
    @property
    def num_reserved_ids(self):
        return self._num_reserved_ids

    @property
    def vocab_size(self):
        return len(self.id_to_token)

    def __len__(self):
        return self.vocab_size

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def seg(self):
        return self.seg_index

    def encode(self, s):
        tokens = s.strip().split()
        if self._replace_oov is not None:
            tokens = [t if t in self.token_to_id else self._replace_oov for t in tokens]
        ids = [self.token_to_id[t] for t in tokens]
        return ids[::-1] if self._reverse else ids

    def decode_list(self, ids):
        seq = reversed(list(ids)) if self._reverse else ids
        return [self.id_to_token.get(int(i), "ID_%d" % int(i)) for i in seq]

    def decode(self, ids, strip_eos=False, strip_padding=False):
        ids = list(ids)
        if strip_padding and self.pad_index in ids:
            ids = ids[:ids.index(self.pad_index)]
        if strip_eos and self.eos_index in ids:
            ids = ids[:ids.index(self.eos_index)]
        return " ".join(self.decode_list(ids))

    def store_to_file(self, filename):
        with open(filename, "w") as fh:
            for i in range(len(self.id_to_token)):
                fh.write(self.id_to_token[i] + "\n")

    def sil_phonemes(self):
        return [p for p in self.id_to_token.values() if is_sil_phoneme(p)]


def is_sil_phoneme(p):
    return p == "" or not p[0].isalpha()


def build_token_encoder(token_list_file):
    """utils/text/text_encoder.py:257-259: the JSON list of phonemes -> encoder with `<UNK>` for out-of-vocabulary tokens."""
    with open(token_list_file) as fh:
        return TokenTextEncoder(None, vocab_list=json.load(fh), replace_oov=UNK)

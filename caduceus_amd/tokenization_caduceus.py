"""Character-level DNA tokenizer with the vocabulary / complement map of the reference
(/root/reference/caduceus/tokenization_caduceus.py:10-135): specials 0-6, then one id per character starting at 7;
`complement_map` maps ids to the ids of the complementary base (A<->T, C<->G, N->N, specials to themselves); left padding.
Host-side only (no kernel): it feeds `config.complement_map`, which the RCPS embedding / LM-head kernels index with.
"""
from typing import Dict, List, Optional, Sequence, Tuple

from transformers import PreTrainedTokenizer

_SPECIALS = ["[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]"]


class CaduceusTokenizer(PreTrainedTokenizer):
    model_input_names = ["input_ids"]

    def __init__(self, model_max_length: int, characters: Sequence[str] = ("A", "C", "G", "T", "N"),
                 complement_map=None, bos_token="[BOS]", eos_token="[SEP]", sep_token="[SEP]", cls_token="[CLS]",
                 pad_token="[PAD]", mask_token="[MASK]", unk_token="[UNK]", **kwargs):
        if complement_map is None:
            complement_map = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
        self.characters = characters
        self.model_max_length = model_max_length
        self._vocab_str_to_int = {tok: i for i, tok in enumerate(_SPECIALS)}
        self._vocab_str_to_int.update({ch: i + len(_SPECIALS) for i, ch in enumerate(self.characters)})
        self._vocab_int_to_str = {v: k for k, v in self._vocab_str_to_int.items()}
        self._complement_map = {
            idx: self._vocab_str_to_int[complement_map[tok]] if tok in complement_map else idx
            for tok, idx in self._vocab_str_to_int.items()
        }
        add_prefix_space = kwargs.pop("add_prefix_space", False)
        padding_side = kwargs.pop("padding_side", "left")
        kwargs.pop("add_special_tokens", None)  # collides with a method name on transformers >= 5
        super().__init__(bos_token=bos_token, eos_token=eos_token, sep_token=sep_token, cls_token=cls_token,
                         pad_token=pad_token, mask_token=mask_token, unk_token=unk_token,
                         add_prefix_space=add_prefix_space, model_max_length=model_max_length,
                         padding_side=padding_side, **kwargs)

    @property
    def vocab_size(self) -> int:
        return len(self._vocab_str_to_int)

    @property
    def complement_map(self) -> Dict[int, int]:
        return self._complement_map

    def _tokenize(self, text: str, **kwargs) -> List[str]:
        return list(text.upper())

    def _convert_token_to_id(self, token: str) -> int:
        return self._vocab_str_to_int.get(token, self._vocab_str_to_int["[UNK]"])

    def _convert_id_to_token(self, index: int) -> str:
        return self._vocab_int_to_str[index]

    def convert_tokens_to_string(self, tokens):
        return "".join(tokens)

    def get_special_tokens_mask(self, token_ids_0: List[int], token_ids_1: Optional[List[int]] = None,
                                already_has_special_tokens: bool = False) -> List[int]:
        if already_has_special_tokens:
            return super().get_special_tokens_mask(token_ids_0=token_ids_0, token_ids_1=token_ids_1,
                                                   already_has_special_tokens=True)
        mask = [0] * len(token_ids_0) + [1]
        if token_ids_1 is not None:
            mask += [0] * len(token_ids_1) + [1]
        return mask

    def build_inputs_with_special_tokens(self, token_ids_0: List[int],
                                         token_ids_1: Optional[List[int]] = None) -> List[int]:
        out = token_ids_0 + [self.sep_token_id]
        if token_ids_1 is not None:
            out += token_ids_1 + [self.sep_token_id]
        return out

    def get_vocab(self) -> Dict[str, int]:
        return self._vocab_str_to_int

    def save_vocabulary(self, save_directory: str, filename_prefix: Optional[str] = None) -> Tuple:
        return ()  # fixed vocabulary, nothing to write

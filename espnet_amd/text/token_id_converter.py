"""TokenIDConverter / tokenizers: host glue the reference re-uses unchanged around the hot path
(espnet2/text/token_id_converter.py:11-66, espnet2/text/{word,char,sentencepieces}_tokenizer.py).
Re-stated minimally so the drop-in is importable without espnet2 on the GPU box."""
from pathlib import Path
from typing import Iterable, List, Union


class TokenIDConverter:
    def __init__(self, token_list: Union[Path, str, Iterable[str]], unk_symbol: str = "<unk>"):
        if isinstance(token_list, (Path, str)):
            with Path(token_list).open("r", encoding="utf-8") as f:
                self.token_list = [line.rstrip("\n") for line in f]
        else:
            self.token_list = list(token_list)
        self.token2id = {}
        for i, t in enumerate(self.token_list):
            if t in self.token2id:
                raise RuntimeError(f'Symbol "{t}" is duplicated')
            self.token2id[t] = i
        self.unk_symbol = unk_symbol
        if unk_symbol not in self.token2id:
            raise RuntimeError(f"Unknown symbol '{unk_symbol}' doesn't exist in the token_list")
        self.unk_id = self.token2id[unk_symbol]

    def get_num_vocabulary_size(self) -> int:
        return len(self.token_list)

    def ids2tokens(self, integers) -> List[str]:
        tl = self.token_list  # ints, numpy / torch integer scalars all index a list
        return [tl[i] for i in integers]

    def tokens2ids(self, tokens: Iterable[str]) -> List[int]:
        return [self.token2id.get(t, self.unk_id) for t in tokens]


class WordTokenizer:
    def __init__(self, delimiter: str = None):
        self.delimiter = delimiter

    def text2tokens(self, line: str) -> List[str]:
        return line.split(self.delimiter)

    def tokens2text(self, tokens: Iterable[str]) -> str:
        return (" " if self.delimiter is None else self.delimiter).join(tokens)


class CharTokenizer:
    def __init__(self, space_symbol: str = "<space>"):
        self.space_symbol = space_symbol

    def text2tokens(self, line: str) -> List[str]:
        return [self.space_symbol if c == " " else c for c in line]

    def tokens2text(self, tokens: Iterable[str]) -> str:
        return "".join(" " if t == self.space_symbol else t for t in tokens)


class SentencepiecesTokenizer:
    def __init__(self, model: Union[Path, str]):
        import sentencepiece as spm

        self.sp = spm.SentencePieceProcessor()
        self.sp.load(str(model))

    def text2tokens(self, line: str) -> List[str]:
        return self.sp.EncodeAsPieces(line)

    def tokens2text(self, tokens: Iterable[str]) -> str:
        return self.sp.DecodePieces(list(tokens))


def build_tokenizer(token_type: str, bpemodel=None, delimiter: str = None, space_symbol: str = "<space>"):
    """espnet2/text/build_tokenizer.py:15-98 for the token types ASR recipes use."""
    if token_type == "bpe":
        if bpemodel is None:
            raise ValueError('bpemodel is required if token_type = "bpe"')
        return SentencepiecesTokenizer(bpemodel)
    if token_type == "word":
        return WordTokenizer(delimiter=delimiter)
    if token_type == "char":
        return CharTokenizer(space_symbol=space_symbol)
    raise NotImplementedError(f"token_type={token_type} is outside the MI355X hot path")

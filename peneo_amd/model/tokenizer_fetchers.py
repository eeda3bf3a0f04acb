"""Token -> source-substring alignment ("tokenizer fetchers", reference: model/backbone_mapping.py:35-249).

The decoder's output is token indices; the key/value STRINGS of the metric are rebuilt by concatenating, per predicted
line, the source substring every token stands for (``text`` in the dataset item, consumed by pipeline/decode.py).  A fetcher
takes the line's text and the tokenizer's token strings and returns one substring per token whose concatenation is the
line text.  Three tokenizer families are used by the registered backbones:

  * byte-level BPE (RoBERTa / LayoutLMv3: ``Ġ`` marks a leading space)       -> ``fetcher_RobertaTokenizer``,
                                                                                ``fetcher_LayoutLMv3Tokenizer``
  * sentencepiece (XLM-R / LayoutXLM / LayoutLMv3-chinese: ``▁`` marks it)   -> ``fetcher_XLMTokenizer``
  * WordPiece (LayoutLMv2: ``##`` continues a word, ``[UNK]``)               -> ``fetcher_LayoutLMv2Tokenizer``

The three "scan" fetchers share one cursor walk: for every character of the token, copy source characters up to and
including the first one that equals it (or its upper case); unknown tokens take the next non-space character with the
spaces before it; what is left of the source goes to the last token.  Running off the end of the source raises IndexError
exactly where the reference does (a token the source cannot account for is a data error, not something to paper over)."""
from __future__ import annotations

from typing import List, Optional


_LATIN1_REPAIRS = (("Â°", "°"), ("Î¿", "o"))  # byte-level tokens of two non-ASCII characters RFUND contains
_ACCENT_FOLD = str.maketrans("áéíóúü", "aeiouu")


def _scan(source: str, tokens: List[str], unk: str, surface) -> List[str]:
    """The shared cursor walk; ``surface(k, token)`` maps a token string to the characters to look for."""
    if len(source) == 0 or source.isspace():
        return []
    pos = 0
    out: List[str] = []
    for k, token in enumerate(tokens):
        piece = []
        if token == unk:
            while source[pos] == " ":
                piece.append(source[pos])
                pos += 1
                if pos >= len(source):
                    break
            piece.append(source[pos])
            pos += 1
        else:
            for ch in surface(k, token):
                while ch != source[pos] and ch.upper() != source[pos]:
                    piece.append(source[pos])
                    pos += 1
                    if pos >= len(source):
                        break
                piece.append(source[pos])
                pos += 1
        out.append("".join(piece))
    out[-1] += source[pos:]
    return out


def _repair(token: str) -> str:
    for bad, good in _LATIN1_REPAIRS:
        token = token.replace(bad, good)
    return token


def fetcher_RobertaTokenizer(orig_text: str, tokens: List[str]) -> List[str]:
    """model/backbone_mapping.py:143-194: a token that starts with ``Ġ`` has all its ``Ġ`` read as spaces."""
    def surface(k: int, token: str) -> str:
        return token.replace("Ġ", " ") if token.startswith("Ġ") else token
    return _scan(orig_text, [_repair(t) for t in tokens], "<unk>", surface)


def fetcher_LayoutLMv3Tokenizer(orig_text: str, tokens: List[str]) -> List[str]:
    """model/backbone_mapping.py:197-250: as above, but the FIRST token of a line drops its space marker (``Ġ`` / ``ĠÂ``)
    instead of reading it as a space."""
    def surface(k: int, token: str) -> str:
        blank = " " if k > 0 else ""
        if token.startswith("ĠÂ"):
            token = token.replace("ĠÂ", blank)
        if token.startswith("Ġ"):
            token = token.replace("Ġ", blank)
        return token
    return _scan(orig_text, [_repair(t) for t in tokens], "<unk>", surface)


def fetcher_LayoutLMv2Tokenizer(orig_text: str, tokens: List[str]) -> List[str]:
    """model/backbone_mapping.py:86-140: WordPiece; accents folded on the source first, ``##`` prefixes dropped."""
    def surface(k: int, token: str) -> str:
        return token[2:] if token.startswith("##") else token
    return _scan(orig_text.translate(_ACCENT_FOLD), tokens, "[UNK]", surface)


def fetcher_XLMTokenizer(orig_text: str, tokens: List[str]) -> List[str]:
    """model/backbone_mapping.py:35-83: sentencepiece; ``▁`` reads as a space, a token character that is not the next source
    character (compared after full-width folding too) produces nothing, a doubled space in the source is swallowed by the
    token that produced the first one, and the last token takes what is left."""
    from ..data.data_utils import string_f2h  # late: the data package imports the model package
    out: List[str] = []
    pos = 0
    for k, token in enumerate(tokens):
        piece = []
        for ch in token.replace("\u2581", " "):
            src = orig_text[pos]
            if ch != src and string_f2h(ch) != string_f2h(src):
                continue
            piece.append(src)
            pos += 1
            if src == " " and orig_text[pos] == " ":
                pos += 1
                piece.append(" ")
        if k == len(tokens) - 1:
            piece.append(orig_text[pos:])
            pos = len(orig_text)
        out.append("".join(piece))
    return out


def fetcher_for(backbone_name: str) -> Optional[object]:
    """The fetcher registered for a backbone key of ``BACKBONE_MAPPING`` (model/backbone_mapping.py:277-348)."""
    return {
        "lilt-infoxlm-base": fetcher_XLMTokenizer,
        "lilt-roberta-en-base": fetcher_RobertaTokenizer,
        "layoutxlm-base": fetcher_XLMTokenizer,
        "layoutlmv2-base-uncased": fetcher_LayoutLMv2Tokenizer,
        "layoutlmv3-base-chinese": fetcher_XLMTokenizer,
        "layoutlmv3-base": fetcher_LayoutLMv3Tokenizer,
    }.get(backbone_name)

"""CPU oracle (TEST INFRASTRUCTURE ONLY - never imported by the product path): the tokenizer text side of the hot path.

Restates, in plain Python, what WhisperKit does with token ids once they leave the decoder:

  * `decode(tokens:)`                 ArgmaxCore/External/Tokenizers/Tokenizer.swift:510-525 (ids -> token strings ->
                                      decoder -> join -> cleanUp) with the ByteLevel decoder of
                                      ArgmaxCore/External/Tokenizers/Decoder.swift:126-165 and the byte table of ByteEncoder.swift
                                      (vendored swift-transformers 1.1.6)
  * `cleanUp(text:)`                  Tokenizer.swift:433-449 (clean_up_tokenization_spaces, default true :407)
  * special tokens / language tokens  WhisperTokenizerWrapper.init, WhisperKit/Core/Models.swift:1198-1224
  * `splitToWordTokens`               Core/Models.swift:1226-1306 (splitTokensOnUnicode / splitTokensOnSpaces)

Pinned by the reference's own known-answer tests (Tests/WhisperKitTests/UnitTests.swift:1288-1375) through the fixture
vocabulary `whisperkit_amd.synth.kat_tokenizer_vocab`, which carries the real id -> text pairs those tests disclose, and
cross-checked against the HF `tokenizers` 0.22 library (the implementation swift-transformers mirrors) in
tests/test_tokenizer_text.py.

Deviation that cannot be restated: `splitToWordTokens` picks the splitter with Apple's NLLanguageRecognizer
(Core/Models.swift:1296-1299).  Here the language code is an argument (the transcription's language), which is what
openai/whisper's own `split_to_word_tokens` does.
"""
import json
import os
import unicodedata
from typing import Dict, List, Optional, Sequence, Tuple

from .decode import SpecialTokens, is_swift_whitespace, trim_whitespaces  # noqa: F401

REPLACEMENT = "�"
UNICODE_SPLIT_LANGUAGES = ("zh", "ja", "th", "lo", "my", "yue")      # Core/Models.swift:1301


def _byte_decoder() -> Dict[str, int]:
    """Inverse of the GPT-2 byte -> character table (ByteEncoder.swift `byteEncoder` / Decoder.swift `byteDecoder`)."""
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


def is_punctuation_scalar(ch: str) -> bool:
    """CharacterSet.punctuationCharacters: Unicode general category P*."""
    return unicodedata.category(ch).startswith("P")


class Tokenizer:
    """AutoTokenizer.from(modelFolder:) for a ByteLevel-BPE `tokenizer.json` (the only kind Whisper ships),
    Utilities/ModelUtilities.swift:175-203."""

    def __init__(self, tokenizer_json: str):
        with open(tokenizer_json, encoding="utf-8") as f:
            tj = json.load(f)
        if (tj.get("decoder") or {}).get("type") != "ByteLevel" or tj["model"]["type"] != "BPE":
            raise ValueError("only ByteLevel BPE tokenizers are supported (WhisperError.tokenizerUnavailable)")
        self.token_to_id: Dict[str, int] = dict(tj["model"]["vocab"])
        self.added_tokens = set()
        self.special_ids = set()
        for a in tj.get("added_tokens", []):
            self.token_to_id[a["content"]] = a["id"]
            self.added_tokens.add(a["content"])
            if a.get("special"):
                self.special_ids.add(a["id"])
        self.id_to_token: Dict[int, str] = {i: t for t, i in self.token_to_id.items()}
        self.clean_up = True                                                # Tokenizer.swift:407 `.boolean(or: true)`
        cfg = os.path.join(os.path.dirname(tokenizer_json), "tokenizer_config.json")
        if os.path.exists(cfg):
            with open(cfg, encoding="utf-8") as f:
                v = json.load(f).get("clean_up_tokenization_spaces")
            if isinstance(v, bool):
                self.clean_up = v
        self._bd = _byte_decoder()

    # -- swift-transformers pass-through ------------------------------------------------------------------------------
    def convertTokenToId(self, token: str) -> Optional[int]:
        return self.token_to_id.get(token)

    def convertIdToToken(self, i: int) -> Optional[str]:
        return self.id_to_token.get(i)

    def _bytelevel(self, tokens: Sequence[str]) -> List[str]:
        """ByteLevelDecoder.decode (Decoder.swift:137-165): runs of ordinary tokens are mapped back to bytes and decoded as
        UTF-8 with repair (String(decoding:as:) == errors="replace"); added tokens are passed through and end a run."""
        out, cur = [], []

        def flush():
            if cur:
                out.append(bytes(self._bd[c] for c in "".join(cur)).decode("utf-8", errors="replace"))
                cur.clear()
        for t in tokens:
            if t in self.added_tokens:
                flush()
                out.append(t)
            else:
                cur.append(t)
        flush()
        return out

    def cleanUp(self, text: str) -> str:
        """Tokenizer.swift:433-449."""
        if not self.clean_up:
            return text
        for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"),
                     (" 's", "'s"), (" 've", "'ve"), (" 're", "'re")):
            text = text.replace(a, b)
        return text

    def decode(self, tokens: Sequence[int], skipSpecialTokens: bool = False) -> str:
        """Tokenizer.swift:510-525; unknown ids are dropped (compactMap)."""
        ids = [t for t in tokens if not (skipSpecialTokens and t in self.special_ids)]
        strs = [self.id_to_token[t] for t in ids if t in self.id_to_token]
        return self.cleanUp("".join(self._bytelevel(strs)))

    # -- WhisperTokenizerWrapper ---------------------------------------------------------------------------------------
    def specialTokens(self) -> SpecialTokens:
        """Core/Models.swift:1198-1211 with the defaults of :1309-1322."""
        g = lambda tok, default: self.token_to_id.get(tok, default)
        return SpecialTokens(
            endToken=g("<|endoftext|>", 50257), englishToken=g("<|en|>", 50259), noSpeechToken=g("<|nospeech|>", 50362),
            noTimestampsToken=g("<|notimestamps|>", 50363), specialTokenBegin=g("<|endoftext|>", 50257),
            startOfPreviousToken=g("<|startofprev|>", 50361), startOfTranscriptToken=g("<|startoftranscript|>", 50258),
            timeTokenBegin=g("<|0.00|>", 50364), transcribeToken=g("<|transcribe|>", 50359),
            translateToken=g("<|translate|>", 50358), whitespaceToken=g(" ", 220))

    def allLanguageTokens(self, language_codes: Sequence[str]) -> set:
        """Core/Models.swift:1219-1223."""
        begin = self.specialTokens().specialTokenBegin
        ids = (self.token_to_id.get(f"<|{c}|>") for c in language_codes)
        return {i for i in ids if i is not None and i > begin}

    def splitTokensOnUnicode(self, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
        """Core/Models.swift:1226-1254.  `decoded.range(of:)` yields an index into `decoded`; subscripting `decodedFull` with
        it addresses the same UTF-8 offset of the full string (native Swift strings), NOT `unicodeOffset + index` as
        openai/whisper does - restated as written."""
        full = self.decode(tokens).encode("utf-8")
        rep = REPLACEMENT.encode("utf-8")
        words, wordTokens, cur = [], [], []
        for t in tokens:
            cur.append(t)
            dec = self.decode(cur)
            db = dec.encode("utf-8")
            at = db.find(rep)
            inFull = at >= 0 and full[at:at + len(rep)] == rep
            if at < 0 or inFull:
                words.append(dec)
                wordTokens.append(cur)
                cur = []
        return words, wordTokens

    def splitTokensOnSpaces(self, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
        """Core/Models.swift:1256-1279."""
        sub, subTok = self.splitTokensOnUnicode(tokens)
        begin = self.specialTokens().specialTokenBegin
        words, wordTokens = [], []
        for s, st in zip(sub, subTok):
            special = st[0] >= begin
            withSpace = s.startswith(" ") and not (len(s) > 1 and unicodedata.category(s[1]) in ("Mn", "Me"))
            stripped = trim_whitespaces(s)
            punctuation = len(stripped) == 1 and is_punctuation_scalar(stripped)
            if special or withSpace or punctuation or not words:
                words.append(s)
                wordTokens.append(list(st))
            else:
                words[-1] += s
                wordTokens[-1] += st
        return words, wordTokens

    def splitToWordTokens(self, tokenIds: Sequence[int], language: str = "en") -> Tuple[List[str], List[List[int]]]:
        """Core/Models.swift:1293-1306 with the language code supplied by the caller (see module docstring)."""
        if language in UNICODE_SPLIT_LANGUAGES:
            return self.splitTokensOnUnicode(tokenIds)
        return self.splitTokensOnSpaces(tokenIds)


def trimming_special_token_characters(s: str) -> str:
    """String.trimmingSpecialTokenCharacters (Constants.specialTokenCharacters = "<|>", Core/Models.swift:1330)."""
    return s.strip("<|>")

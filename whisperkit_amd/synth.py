"""Synthetic 30 s / 16 kHz PCM chunks: 0.05 N(0,1) noise + three sinusoids, clipped to [-1, 1].

Used by bench.py and the parity tests so both sides see byte-identical inputs; there are no
datasets in the image (no network).
"""
import numpy as np

WINDOW_SAMPLES = 480000
SAMPLE_RATE = 16000


def synthetic_chunk(seed: int, n: int = WINDOW_SAMPLES) -> np.ndarray:
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    x = 0.05 * rng.standard_normal(n)
    for f in (220.0, 440.0, 1760.0):
        x += 0.1 * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    # slow amplitude envelope so that the spectrogram is not stationary
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * 0.37 * t + rng.uniform(0, 2 * np.pi))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def bench_chunk_seed(chunk_index: int, audio_set: int = 0) -> int:
    """Seed of chunk `chunk_index` of audio set `audio_set`.  bench.py cycles its run steps through a small pool of sets, so the packed steps
    of one device batch and the sessions in flight carry DIFFERENT audio (VERDICT r05 "what's weak" 9: identical halves let the embedding
    kernel read one row for two slots).  Set 0 keeps the seeds of rounds 1 - 5 (1234 + chunk index): the CPU baseline's token check and
    tests/test_gpu_fulldepth.py use it."""
    return 1234 + chunk_index + 1000 * audio_set


# ------------------------------------------------------------------------------------------------ tokenizer fixture
# Whisper language codes in openai/whisper token order (same set as Constants.languages, Core/Models.swift:1335-1449).
LANGUAGE_CODES = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi "
                  "ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd "
                  "gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su yue").split()

# id -> text pairs of the REAL multilingual Whisper vocabulary that the reference's own tests disclose
# (Tests/WhisperKitTests/UnitTests.swift:1289-1375: testTokenizerOutput, testSplitToWordTokens{,Spanish,Japanese}).
KAT_VOCAB = {
    400: " And", 370: " so", 452: " my", 7177: " fellow", 6280: " Americans", 1029: " ask", 406: " not", 437: " what",
    428: " your", 1941: " country", 393: " can", 360: " do", 337: " for", 291: " you",
    2425: " Hello", 1002: " world", 639: " This", 307: " is", 257: " a", 31636: "test", 1943: " isn", 380: "'t", 309: " it",
    24364: "¡", 48529: "Hola", 376: " M", 6043: "undo", 20547: " Esta", 785: " es", 2002: " una", 48241: " prueba",
    3841: " ¿", 1771: "no",
    38088: "こんにちは", 1231: "、", 24486: "世界", 25212: "これは", 22985: "テ", 40498: "スト", 4767: "です", 30346: "よね",
}


def bytes_to_unicode():
    """GPT-2 byte <-> printable-character table (the byteEncoder of ArgmaxCore/External/Tokenizers/ByteEncoder.swift);
    ids 0..255 of every Whisper vocabulary are these characters in this order."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, map(chr, cs)))


def kat_tokenizer_vocab(n_vocab: int):
    """(vocab token->id over the BPE range, added_tokens list) of a Whisper-shaped byte-level vocabulary of size n_vocab:
    ids 0..255 are the byte characters (exact), the ids in KAT_VOCAB carry the real texts, every other BPE id is filler
    (pseudo-words with / without a leading space, digits, CJK, punctuation runs and lone UTF-8 lead/continuation byte pairs so
    that the replacement-character logic of splitTokensOnUnicode is exercised), specials sit at the ids of
    openai/whisper's tokenizer for that vocabulary size."""
    b2u = bytes_to_unicode()
    enc = lambda s: "".join(b2u[b] for b in s.encode("utf-8"))
    eot = 50256 if n_vocab == 51864 else 50257
    n_lang = 100 if n_vocab == 51866 else 99
    if n_vocab not in (51864, 51865, 51866):
        raise ValueError(f"unknown Whisper vocabulary size {n_vocab}")
    byte_order = list(b2u.values())
    id2tok = {i: byte_order[i] for i in range(256)}
    used = set(id2tok.values())
    for i, text in KAT_VOCAB.items():
        id2tok[i] = enc(text)
        used.add(id2tok[i])
    syll = ["ka", "to", "mi", "ne", "ru", "so", "la", "vi", "en", "or", "an", "th", "ch", "ou", "st", "er"]
    cjk = "日本語中文字漢話時間人年月水火木金土山川田"
    punct = ["...", "--", "?!", ").", "\",", "('", "::", ";;"]
    for i in range(256, eot):
        if i in id2tok:
            continue
        k, kind = i, i % 16
        if kind == 13:       # two raw bytes that are not valid UTF-8 on their own: lead byte + one continuation byte of a 3-byte char
            tok = b2u[0xE0 + (k // 16) % 16] + b2u[0x80 + (k // 256) % 64]
        elif kind == 14:
            tok = enc(cjk[(k // 16) % len(cjk)] + cjk[(k // 400) % len(cjk)])
        elif kind == 15:
            tok = enc((" " if (k // 16) % 2 else "") + punct[(k // 32) % len(punct)])
        elif kind == 12:
            tok = enc((" " if (k // 16) % 2 else "") + str(k))
        else:
            w, q = "", k
            for _ in range(3):
                w += syll[q % 16]
                q //= 16
            tok = enc((" " if kind < 8 else "") + w)
        n = 0
        while tok in used:            # keep the vocabulary injective
            n += 1
            tok = tok + enc(syll[(k + n) % 16])
        id2tok[i] = tok
        used.add(tok)
    vocab = {id2tok[i]: i for i in range(eot)}
    specials = ["<|endoftext|>", "<|startoftranscript|>"] + [f"<|{c}|>" for c in LANGUAGE_CODES[:n_lang]] + \
        ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"] + \
        [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
    added = [{"id": eot + j, "content": s, "single_word": False, "lstrip": False, "rstrip": False, "normalized": False,
              "special": True} for j, s in enumerate(specials)]
    assert eot + len(specials) == n_vocab, (eot, len(specials), n_vocab)
    return vocab, added


def write_kat_tokenizer(folder: str, n_vocab: int = 51865, clean_up_tokenization_spaces: bool = True) -> str:
    """Write `tokenizer.json` + `tokenizer_config.json` (HF tokenizers layout: ByteLevel BPE, added special tokens, ByteLevel
    decoder) for kat_tokenizer_vocab(n_vocab) into `folder`; returns the tokenizer.json path.  There is no real Whisper
    tokenizer.json in the image (no network); this fixture is exact wherever the reference's tests pin the real one."""
    import json, os
    vocab, added = kat_tokenizer_vocab(n_vocab)
    tj = {
        "version": "1.0", "truncation": None, "padding": None, "added_tokens": added, "normalizer": None,
        "pre_tokenizer": {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True},
        "post_processor": None,
        "decoder": {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": True, "use_regex": True},
        "model": {"type": "BPE", "dropout": None, "unk_token": None, "continuing_subword_prefix": "", "end_of_word_suffix": "",
                  "fuse_unk": False, "byte_fallback": False, "ignore_merges": False, "vocab": vocab, "merges": []},
    }
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "tokenizer.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(tj, f, ensure_ascii=False)
    with open(os.path.join(folder, "tokenizer_config.json"), "w", encoding="utf-8") as f:
        json.dump({"tokenizer_class": "WhisperTokenizer", "clean_up_tokenization_spaces": clean_up_tokenization_spaces,
                   "bos_token": "<|endoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>"}, f)
    return path

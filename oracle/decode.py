"""Restatement of the reference's Swift host logic for the decode path - test infrastructure only.

Each function cites the reference file:line (relative to /root/reference/Sources/WhisperKit) it
follows.  Arithmetic notes where this restatement knowingly differs from an arm64 run of the
reference (and why that cannot be pinned here) are marked NOTE(parity).

NOTE(parity) logits: the reference receives logits as Float16 from CoreML (Core/Models.swift:1041)
and runs the timestamp log-softmax in Float16 on arm64 (Core/Text/LogitsFilter.swift:152-237).
The oracle keeps logits in float32/float64; the product does the same (DESIGN.md "numerics").
NOTE(parity) sampling at T>0 uses the unseeded system RNG in the reference
(Core/Text/TokenSampler.swift:61,169); oracle and product share a seeded counter-based
generator (`uniform01`) so that T>0 decoding is reproducible.
NOTE(parity) beam search (BeamSearchTokenSampler, decode_text_beam): PARITY UNPINNED.  The reference's sampler of that name is
fatalError (Core/Text/TokenSampler.swift:254-290); the restatement follows openai/whisper's BeamSearchDecoder, which is not in
this container either, so no golden vector exists - the properties it is checked on are in tests/test_beam_search.py.
"""
from __future__ import annotations

import dataclasses
import math
import zlib
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

SAMPLE_RATE = 16000            # WhisperKit.sampleRate       (Core/WhisperKit.swift:38)
SECONDS_PER_TIME_TOKEN = 0.02  # WhisperKit.secondsPerTimeToken (Core/WhisperKit.swift:40)
WINDOW_SAMPLES = 480000        # Constants.defaultWindowSamples (Core/Models.swift:1457)
MAX_TOKEN_CONTEXT = 448 // 2   # Constants.maxTokenContext    (Core/Models.swift:1334)
NEG_INF = -np.inf


# ----------------------------------------------------------------------------- types
@dataclasses.dataclass(frozen=True)
class SpecialTokens:
    """Core/Models.swift:1111-1149."""
    endToken: int = 50257
    englishToken: int = 50259
    noSpeechToken: int = 50362
    noTimestampsToken: int = 50363
    specialTokenBegin: int = 50257
    startOfPreviousToken: int = 50361
    startOfTranscriptToken: int = 50258
    timeTokenBegin: int = 50364
    transcribeToken: int = 50359
    translateToken: int = 50358
    whitespaceToken: int = 220


def special_tokens_for_vocab(n_vocab: int) -> Tuple[SpecialTokens, List[int]]:
    """Token ids of the three Whisper vocabularies (openai/whisper tokenizer.py special-token order:
    eot, sot, languages, translate, transcribe, startoflm, startofprev, nospeech, notimestamps, <|0.00|>...).
    Multilingual defaults equal Core/Models.swift:1309-1322.  Returns (SpecialTokens, language token ids)."""
    if n_vocab == 51864:      # *.en (gpt2 vocab): eot 50256; language slots exist but only <|en|> is meaningful
        eot, n_lang = 50256, 99
    elif n_vocab == 51865:
        eot, n_lang = 50257, 99
    elif n_vocab == 51866:    # large-v3 (+ <|yue|>)
        eot, n_lang = 50257, 100
    else:
        raise ValueError(f"unknown Whisper vocabulary size {n_vocab}")
    sot = eot + 1
    lang0 = sot + 1
    translate = lang0 + n_lang
    st = SpecialTokens(endToken=eot, englishToken=lang0, noSpeechToken=translate + 4,
                       noTimestampsToken=translate + 5, specialTokenBegin=eot,
                       startOfPreviousToken=translate + 3, startOfTranscriptToken=sot,
                       timeTokenBegin=translate + 6, transcribeToken=translate + 1,
                       translateToken=translate, whitespaceToken=220)
    assert st.timeTokenBegin + 1501 == n_vocab
    return st, list(range(lang0, lang0 + n_lang))


@dataclasses.dataclass
class DecodingOptions:
    """Core/Configurations.swift:155-247 (same names, same defaults)."""
    verbose: bool = False
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    temperatureIncrementOnFallback: float = 0.2
    temperatureFallbackCount: int = 5
    sampleLength: int = MAX_TOKEN_CONTEXT
    topK: int = 5
    usePrefillPrompt: bool = True
    detectLanguage: Optional[bool] = None
    skipSpecialTokens: bool = False
    withoutTimestamps: bool = False
    wordTimestamps: bool = False
    maxInitialTimestamp: Optional[float] = None
    maxWindowSeek: Optional[int] = None
    clipTimestamps: List[float] = dataclasses.field(default_factory=list)
    windowClipTime: float = 1.0
    promptTokens: Optional[List[int]] = None
    prefixTokens: Optional[List[int]] = None
    suppressBlank: bool = False
    suppressTokens: List[int] = dataclasses.field(default_factory=list)
    compressionRatioThreshold: Optional[float] = 2.4
    logProbThreshold: Optional[float] = -1.0
    firstTokenLogProbThreshold: Optional[float] = -1.5
    noSpeechThreshold: Optional[float] = 0.6
    concurrentWorkerCount: int = 4
    chunkingStrategy: Optional[str] = None
    # Not a DecodingOptions field of the reference: the reference-numerics switch of the C ABI (wh_decoding_options.float16_logits).
    # True emulates the arm64 FloatType = Float16 path: logits rounded to Float16 (Core/Models.swift:1041) and the
    # TimestampRulesFilter comparison evaluated on Float16 log-probabilities (Core/Text/LogitsFilter.swift:144-242).
    float16Logits: bool = False
    beamSize: int = 0             # > 1: beam search at T = 0 (BeamSearchTokenSampler below; no reference behaviour)
    beamPatience: float = 1.0

    def __post_init__(self):
        if self.detectLanguage is None:   # Configurations.swift:222
            self.detectLanguage = not self.usePrefillPrompt

    def prepareSeekClips(self, contentFrames: int) -> List[Tuple[int, int]]:
        """Utilities/Extensions+Internal.swift:112-130."""
        seek_points = [int(_swift_round(np.float32(t) * np.float32(SAMPLE_RATE))) for t in self.clipTimestamps]
        if len(seek_points) == 0:
            seek_points.append(0)
        if len(seek_points) % 2 == 1:
            seek_points.append(contentFrames)
        clips = []
        for i in range(0, len(seek_points), 2):
            start = seek_points[i]
            end = seek_points[i + 1] if i + 1 < len(seek_points) else contentFrames
            clips.append((start, end))
        return clips


def _swift_round(x: float) -> float:
    """Swift `round()` = schoolbook rounding (half away from zero)."""
    return math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1)


def _rounded(x: float, places: int) -> float:
    """Float.rounded(_ places:) (ArgmaxCore FloatType helpers): round(x * 10^p) / 10^p in Float."""
    m = np.float32(10.0 ** places)
    return float(np.float32(_swift_round(float(np.float32(x) * m))) / m)


@dataclasses.dataclass
class DecodingFallback:
    needsFallback: bool
    fallbackReason: str


def decoding_fallback(options: DecodingOptions, isFirstTokenLogProbTooLow: bool, noSpeechProb: float,
                      compressionRatio: float, avgLogProb: float) -> Optional[DecodingFallback]:
    """Core/Models.swift:357-381 - NOTE: order matters."""
    if isFirstTokenLogProbTooLow:
        return DecodingFallback(True, "firstTokenLogProbThreshold")
    if options.noSpeechThreshold is not None and noSpeechProb > options.noSpeechThreshold:
        return DecodingFallback(False, "silence")
    if options.compressionRatioThreshold is not None and compressionRatio > options.compressionRatioThreshold:
        return DecodingFallback(True, "compressionRatioThreshold")
    if options.logProbThreshold is not None and avgLogProb < options.logProbThreshold:
        return DecodingFallback(True, "logProbThreshold")
    return None


@dataclasses.dataclass
class DecodingResult:
    """Core/Models.swift:383-439 (text omitted unless a tokenizer is supplied)."""
    language: str
    tokens: List[int]
    tokenLogProbs: List[Dict[int, float]]
    avgLogProb: float
    noSpeechProb: float
    temperature: float
    compressionRatio: float
    fallback: Optional[DecodingFallback]
    alignment: Optional[np.ndarray] = None   # DecodingCache.alignmentWeights [224, 1500]
    text: str = ""
    languageProbs: Dict[str, float] = dataclasses.field(default_factory=dict)
    isFirstTokenLogProbTooLow: bool = False
    steps: int = 0


@dataclasses.dataclass
class WordTiming:
    word: str
    tokens: List[int]
    start: float
    end: float
    probability: float

    @property
    def duration(self) -> float:
        return float(np.float32(self.end) - np.float32(self.start))


@dataclasses.dataclass
class TranscriptionSegment:
    """Core/Models.swift TranscriptionSegment."""
    id: int
    seek: int
    start: float
    end: float
    text: str
    tokens: List[int]
    tokenLogProbs: List[Dict[int, float]]
    temperature: float
    avgLogprob: float
    compressionRatio: float
    noSpeechProb: float
    words: Optional[List[WordTiming]] = None


# ----------------------------------------------------------------------------- compression ratio
def compression_ratio(tokens: Sequence[int]) -> float:
    """Utilities/TextUtilities.swift:14-30: len(int32 bytes) / len(NSData.compressed(using: .zlib)).
    Apple's `.zlib` is a raw DEFLATE stream (RFC 1951, no zlib header) at level 5.
    NOTE(parity): Apple's encoder is not bit-identical to zlib's; sizes can differ by a few bytes.
    Empty input: NSData.compressed throws on empty data -> +inf (TextUtilities.swift:26-29)."""
    data = np.asarray(list(tokens), dtype="<i4").tobytes()
    if len(data) == 0:
        return float("inf")
    c = zlib.compressobj(5, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return float(np.float32(len(data)) / np.float32(len(comp)))


# ----------------------------------------------------------------------------- logits filters
def fill_indexes(logits: np.ndarray, indexes: Sequence[Sequence[int]], value) -> np.ndarray:
    """MLMultiArray.fill(indexes:with:) (ArgmaxCore/MLMultiArrayExtensions.swift:102-109): every multi-dimensional index of `indexes` is set to `value`, in place
    (the filters' logits are a [1, 1, n] array; an empty list changes nothing).  KAT: UnitTests.swift:1903-1920."""
    for idx in indexes:
        logits[tuple(int(i) for i in idx)] = value
    return logits


def fill_last_dimension(logits: np.ndarray, indexes: range, value) -> np.ndarray:
    """MLMultiArray.fillLastDimension(indexes:with:) (ArgmaxCore/MLMultiArrayExtensions.swift:90-99): a half-open range of the last dimension of a [1, 1, n] array (what
    TimestampRulesFilter uses, Core/Text/LogitsFilter.swift:91-126); the oracle's filters address the same ranges as slices of their 1-D logits row."""
    assert logits.ndim == 3 and logits.shape[0] == 1 and logits.shape[1] == 1, "Must have [1, 1, n] shape"
    logits[0, 0, indexes.start:indexes.stop] = value
    return logits


def batched(items: Sequence, size: int) -> List[list]:
    """Array.batched(into:) (ArgmaxCore/FoundationExtensions.swift:41-47): how WhisperKit.transcribeWithOptions cuts the audio list into groups of concurrentWorkerCount
    (Core/WhisperKit.swift:739); the last group may be shorter.  KAT: UnitTests.swift:1922-1931."""
    return [list(items[i:i + size]) for i in range(0, len(items), size)]


class SuppressTokensFilter:
    """Core/Text/LogitsFilter.swift:12-25."""
    def __init__(self, suppressTokens: Sequence[int]):
        self.suppressTokens = list(suppressTokens)

    def filterLogits(self, logits: np.ndarray, tokens: Sequence[int]) -> np.ndarray:
        for t in self.suppressTokens:
            logits[t] = NEG_INF
        return logits


class SuppressBlankFilter:
    """Core/Text/LogitsFilter.swift:27-51."""
    def __init__(self, specialTokens: SpecialTokens, sampleBegin: int):
        self.specialTokens, self.sampleBegin = specialTokens, sampleBegin

    def filterLogits(self, logits, tokens):
        if len(tokens) != self.sampleBegin:
            return logits
        logits[self.specialTokens.whitespaceToken] = NEG_INF
        logits[self.specialTokens.endToken] = NEG_INF
        return logits


class TimestampRulesFilter:
    """Core/Text/LogitsFilter.swift:54-243."""
    def __init__(self, specialTokens: SpecialTokens, sampleBegin: int, maxInitialTimestampIndex: Optional[int],
                 isModelMultilingual: bool, float16: bool = False):
        self.specialTokens, self.sampleBegin = specialTokens, sampleBegin
        self.maxInitialTimestampIndex, self.isModelMultilingual = maxInitialTimestampIndex, isModelMultilingual
        self.float16 = float16

    def _sampleBegin(self, tokens) -> Optional[int]:       # :131-142
        st = self.specialTokens
        if self.isModelMultilingual:
            for i, t in enumerate(tokens[:3]):
                if t == st.transcribeToken or t == st.translateToken:
                    return max(i + 1, self.sampleBegin)
            return None
        return self.sampleBegin

    def filterLogits(self, logits, tokens):
        st = self.specialTokens
        sb = self._sampleBegin(tokens)
        if sb is None or not (sb <= len(tokens)):
            return logits
        logits[st.noTimestampsToken] = NEG_INF                                   # :81
        if len(tokens) > sb:                                                      # :83-110
            sampled = list(tokens[sb:])
            lastWasTimestamp = len(sampled) >= 1 and sampled[-1] >= st.timeTokenBegin
            penultimateWasTimestamp = len(sampled) < 2 or sampled[-2] >= st.timeTokenBegin
            if lastWasTimestamp:
                if penultimateWasTimestamp:
                    logits[st.timeTokenBegin:] = NEG_INF
                else:
                    logits[: st.endToken] = NEG_INF
            timestamps = [t for t in sampled if t >= st.timeTokenBegin]
            if timestamps:
                last = timestamps[-1]
                timestampLast = last if (lastWasTimestamp and not penultimateWasTimestamp) else last + 1
                logits[st.timeTokenBegin: timestampLast] = NEG_INF
        # :112-122 initial-timestamp rule is commented out in the reference - restated as absent.
        if self._sumOfProbabilityOverTimestampsIsAboveAnyOtherToken(logits, st.timeTokenBegin, self.float16):   # :125
            logits[: st.timeTokenBegin] = NEG_INF
        return logits

    @staticmethod
    def _sumOfProbabilityOverTimestampsIsAboveAnyOtherToken(logits, timeTokenBegin, float16: bool = False) -> bool:
        """:144-242  logsumexp(logprobs[tb:]) > max(logprobs[:tb]); the log-softmax normaliser is common
        to both sides, so it is evaluated on the logits directly (float64).
        float16: the reference's arithmetic type is FloatType = Float16 there (BNNS logSoftmax -> logSumExp / max, each a
        Float16 result); BNNS' internal precision is unspecified, so the emulation evaluates in float64 and rounds the two
        log-probabilities that are compared to Float16."""
        x = np.asarray(logits, dtype=np.float64)
        ts, tx = x[timeTokenBegin:], x[:timeTokenBegin]
        mts = ts.max() if ts.size else NEG_INF
        if not np.isfinite(mts):
            return False if mts == NEG_INF else True
        timestampLogProb = mts + math.log(float(np.exp(ts - mts).sum()))
        maxText = tx.max() if tx.size else NEG_INF
        if float16:
            if maxText == NEG_INF:
                return True
            m = x.max()
            lse = m + math.log(float(np.exp(x - m).sum()))
            return bool(np.float16(timestampLogProb - lse) > np.float16(maxText - lse))
        return bool(timestampLogProb > maxText)


class LanguageLogitsFilter:
    """Core/Text/LogitsFilter.swift:245-276."""
    def __init__(self, allLanguageTokens: Sequence[int], logitsDim: int, sampleBegin: int):
        self.allLanguageTokens, self.logitsDim, self.sampleBegin = set(allLanguageTokens), logitsDim, sampleBegin

    def filterLogits(self, logits, tokens):
        if not (len(tokens) >= self.sampleBegin):
            return logits
        keep = np.zeros(self.logitsDim, dtype=bool)
        keep[list(self.allLanguageTokens)] = True
        logits[~keep] = NEG_INF
        return logits


def create_logits_filters(options: DecodingOptions, prefilledIndex: int, initialPromptIndex: int,
                          st: SpecialTokens, isModelMultilingual: bool, custom=()):
    """Core/TextDecoder.swift:857-899 (order: custom, SuppressBlank, SuppressTokens, TimestampRules)."""
    fs = list(custom)
    if options.suppressBlank:
        fs.append(SuppressBlankFilter(st, sampleBegin=prefilledIndex))
    if len(options.suppressTokens) > 0:
        fs.append(SuppressTokensFilter([t for t in options.suppressTokens if t < st.specialTokenBegin]))
    if not options.withoutTimestamps:
        mi = None
        if options.maxInitialTimestamp is not None:
            mi = int(np.float32(options.maxInitialTimestamp) / np.float32(SECONDS_PER_TIME_TOKEN))
        fs.append(TimestampRulesFilter(st, sampleBegin=initialPromptIndex, maxInitialTimestampIndex=mi,
                                       isModelMultilingual=isModelMultilingual, float16=options.float16Logits))
    return fs


# ----------------------------------------------------------------------------- sampler
def uniform01(seed: int, counter: int) -> float:
    """Counter-based uniform in [0,1): splitmix64(seed + counter * golden) top 24 bits.
    Shared (by specification, not by code) with the HIP sampler kernel."""
    M = (1 << 64) - 1
    z = (seed + (counter + 1) * 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return (z >> 40) * (1.0 / (1 << 24))


class GreedyTokenSampler:
    """Core/Text/TokenSampler.swift:29-252.  T == 0: argmax + log softmax; T != 0: softmax(logits / T),
    top-k, multinomial over the k (BNNS path :140-180)."""
    def __init__(self, temperature: float, eotToken: int, decodingOptions: DecodingOptions, seed: int = 0):
        self.temperature = float(np.float16(temperature))   # FloatType on arm64
        self.eotToken, self.decodingOptions, self.seed = eotToken, decodingOptions, seed

    def sample(self, logits: np.ndarray, counter: int = 0) -> Tuple[int, float]:
        x = np.asarray(logits, dtype=np.float64)
        if self.temperature != 0.0:
            x = x * float(np.float32(1.0) / np.float32(self.temperature))     # :116-121 alpha = Float(1/temperature)
        m = x.max()
        lse = m + math.log(float(np.exp(x - m).sum()))
        if self.temperature != 0.0:
            k = self.decodingOptions.topK
            order = np.argsort(-x, kind="stable")[:k]                         # top-k, descending, ties -> lower id
            probs = np.exp(x[order] - lse)
            total = float(probs.sum())
            rnd = uniform01(self.seed, counter) * total                       # Float.random(in: 0..<sum)
            acc, chosen = 0.0, 0
            for i in range(len(order)):
                acc += float(probs[i])
                if rnd < acc:
                    chosen = i
                    break
            tok = int(order[chosen])
        else:
            tok = int(np.argmax(x))                                           # first maximum
        return tok, float(x[tok] - lse)

    def update(self, tokens: List[int], logits: np.ndarray, logProbs: List[float], counter: int = 0):
        tok, lp = self.sample(logits, counter)
        return tokens + [tok], logProbs + [lp], tok == self.eotToken          # :215-240

    def finalize(self, tokens: List[int], logProbs: List[float]):             # :242-251
        t, l = list(tokens), list(logProbs)
        if not t or t[-1] != self.eotToken:
            t.append(self.eotToken)
            l.append(0.0)
        return t, l


# ----------------------------------------------------------------------------- prompt
def prefill_prompt(options: Optional[DecodingOptions], st: SpecialTokens, isModelMultilingual: bool,
                   languageToken: Optional[int] = None) -> List[int]:
    """Core/TextDecoder.swift:163-216.  `languageToken`: id of "<|{options.language ?? en}|>" resolved by the
    caller's tokenizer; falls back to englishToken like the reference (:184)."""
    prefill = [st.startOfTranscriptToken]
    if options is not None:
        if isModelMultilingual:
            prefill.append(languageToken if languageToken is not None else st.englishToken)
            prefill.append(st.translateToken if options.task == "translate" else st.transcribeToken)
        prefill.append(st.noTimestampsToken if options.withoutTimestamps else st.timeTokenBegin)
        if options.promptTokens is not None:
            maxPromptLen = (MAX_TOKEN_CONTEXT // 2) - 1
            trimmed = [t for t in options.promptTokens[-maxPromptLen:] if t < st.specialTokenBegin]
            prefill = [st.startOfPreviousToken] + trimmed + prefill
        if options.prefixTokens is not None:
            trimmedPrefix = [t for t in options.prefixTokens[-(MAX_TOKEN_CONTEXT // 2):] if t < st.specialTokenBegin]
            prefill += trimmedPrefix
    return prefill


# ----------------------------------------------------------------------------- decodeText
StepFn = Callable[[int, int], np.ndarray]    # (token, cacheLength/tokenIndex) -> logits[V]  (float32)


def decode_text(step: StepFn, initialPrompt: List[int], sampler: GreedyTokenSampler, options: DecodingOptions,
                st: SpecialTokens, isModelMultilingual: bool, languageTokens: Sequence[int] = (),
                prefilledIndex: int = 0, custom_filters=(), alignment: Optional[np.ndarray] = None,
                record_logits: Optional[list] = None) -> DecodingResult:
    """Core/TextDecoder.swift:541-855."""
    initialPromptIndex = len(initialPrompt)
    currentTokens = list(initialPrompt)
    nextToken = initialPrompt[-1]
    logProbs = [0.0] * len(currentTokens)
    filters = create_logits_filters(options, prefilledIndex, initialPromptIndex, st, isModelMultilingual, custom_filters)
    loopCount = min(options.sampleLength, MAX_TOKEN_CONTEXT - 1)
    isFirstTokenLogProbTooLow = False
    steps = 0
    for tokenIndex in range(prefilledIndex, loopCount):
        isPrefill = tokenIndex < initialPromptIndex - 1
        isLastPrefillToken = tokenIndex == initialPromptIndex - 1
        isFirstToken = tokenIndex == prefilledIndex
        if tokenIndex < initialPromptIndex:                                                    # :581-594
            isTimestampToken = currentTokens[tokenIndex] >= st.timeTokenBegin
            modelPredictedTimestamp = nextToken >= st.timeTokenBegin
            if not (isLastPrefillToken and isTimestampToken and modelPredictedTimestamp):
                nextToken = currentTokens[tokenIndex]
            else:
                currentTokens[tokenIndex] = nextToken
        logits = np.array(step(nextToken, tokenIndex), dtype=np.float32, copy=True)             # :616
        if options.float16Logits:                        # MLMultiArray of FloatType (Core/Models.swift:1041)
            logits = logits.astype(np.float16).astype(np.float32)
        steps += 1
        raw = logits.copy() if record_logits is not None else None
        for f in filters:                                                                       # :641-643
            logits = f.filterLogits(logits, currentTokens)
        if record_logits is not None:
            record_logits.append((tokenIndex, nextToken, raw, logits.copy()))
        tok, lp = sampler.sample(logits, counter=tokenIndex)                                     # :652
        nextToken, nextTokenLogProb = tok, lp
        completed = tok == sampler.eotToken
        isFirstTokenLogProbTooLow = bool(isFirstToken and options.firstTokenLogProbThreshold is not None
                                         and nextTokenLogProb < options.firstTokenLogProbThreshold)   # :662-667
        isSegmentCompleted = completed or len(currentTokens) >= MAX_TOKEN_CONTEXT - 1 or isFirstTokenLogProbTooLow
        if isSegmentCompleted:
            break
        if not isPrefill:                                                                       # :682-686
            currentTokens.append(nextToken)
            logProbs.append(nextTokenLogProb)
    segmentTokens, segmentLogProbs = sampler.finalize(currentTokens, logProbs)                   # :776
    startIndex = segmentTokens.index(st.startOfTranscriptToken) if st.startOfTranscriptToken in segmentTokens else 0
    endIndex = segmentTokens.index(st.endToken) if st.endToken in segmentTokens else len(segmentTokens)
    filteredTokens = segmentTokens[startIndex: endIndex + 1]
    filteredLogProbs = segmentLogProbs[startIndex: endIndex + 1]
    s = np.float32(0)
    for v in filteredLogProbs:
        s = np.float32(s + np.float32(v))
    avgLogProbs = float(s / np.float32(len(filteredLogProbs)))
    tokenProbs = [{t: float(np.float32(l))} for t, l in zip(filteredTokens, filteredLogProbs)]
    wordTokens = [t for t in filteredTokens if t < st.specialTokenBegin]
    finalCompressionRatio = compression_ratio(wordTokens)
    temperature = _rounded(float(np.float16(sampler.temperature)), 3)                            # :796-800
    noSpeechProb = 0.0                                                                           # :802 (TODO in reference)
    language = options.language or "en"
    if options.language is None:                                                                 # :807-822
        language = "en"
        for t in filteredTokens:
            if t in set(languageTokens):
                language = f"<lang:{t}>"     # text needs a tokenizer; ids are the parity target
                break
    fallback = decoding_fallback(options, isFirstTokenLogProbTooLow, noSpeechProb, finalCompressionRatio, avgLogProbs)
    return DecodingResult(language=language, tokens=filteredTokens, tokenLogProbs=tokenProbs, avgLogProb=avgLogProbs,
                          noSpeechProb=noSpeechProb, temperature=temperature, compressionRatio=finalCompressionRatio,
                          fallback=fallback, alignment=alignment, isFirstTokenLogProbTooLow=isFirstTokenLogProbTooLow,
                          steps=steps)


# ----------------------------------------------------------------------------- beam search (NO REFERENCE BEHAVIOUR)
class BeamSearchTokenSampler:
    """The reference declares this class (Core/Text/TokenSampler.swift:254-290: beamSize, eotToken, patience, maxCandidates =
    Int(Float(beamSize) * patience), finishedSequences) but `update` and `finalize` are `fatalError("Not implemented")`.  The
    behaviour restated here is openai/whisper's BeamSearchDecoder (whisper/decoding.py, v20231117: update :343-404, finalize
    :406-424) for ONE audio - the only published semantics for this sampler.  PARITY UNPINNED: neither openai/whisper nor a
    golden vector of it is in this container; BASELINE configs[4] / SURVEY 8(d) c5 ask for it labelled "no reference behaviour".
    Additions over openai's decoder: the per-token log-probabilities travel with a beam (DecodingResult.tokenLogProbs needs them).
    Ties: top-k and the candidate ranking are stable (lower token id / earlier candidate first), where torch.topk is unspecified."""
    def __init__(self, beamSize: int, eotToken: int, patience: float = 1.0):
        self.beamSize, self.eotToken, self.patience = int(beamSize), int(eotToken), float(patience)
        self.maxCandidates = int(np.float32(beamSize) * np.float32(patience))
        if self.maxCandidates <= 0 or self.beamSize <= 0:
            raise ValueError(f"Invalid beam size {beamSize} or patience {patience}")            # fatalError in the reference
        self.reset()

    def reset(self):
        self.finishedSequences: Dict[tuple, Tuple[float, tuple]] = {}     # tokens -> (sum of log-probs (f32), per-token log-probs)
        self.minMargin = math.inf         # test instrumentation: smallest score gap that decided a ranking (near-tie detector)

    def update(self, beams: List[Tuple[List[int], List[float], float]], logprobRows: List[np.ndarray]):
        """beams: (tokens, logProbs, sumLogProb) per live beam; logprobRows[j]: log-softmax of beam j's filtered logits (f32).
        Returns (new beams, source index per new beam, completed)."""
        scores: Dict[tuple, Tuple[np.float32, int, np.float32]] = {}
        for j, (toks, _lps, sm) in enumerate(beams):                                             # STEP 1 (:358-366)
            row = np.asarray(logprobRows[j], dtype=np.float32)
            for t in np.argsort(-row, kind="stable")[: self.beamSize + 1]:
                scores[tuple(toks) + (int(t),)] = (np.float32(np.float32(sm) + row[t]), j, row[t])
        ranked = sorted(scores, key=lambda q: scores[q][0], reverse=True)                        # STEP 2 (:368-381), stable
        newBeams, sources, finished = [], [], {}
        cut = None
        for r_i, seq in enumerate(ranked):
            sc, j, lp = scores[seq]
            if seq[-1] == self.eotToken:
                finished[seq] = (float(sc), tuple(beams[j][1]) + (float(lp),))
            else:
                newBeams.append((list(seq), list(beams[j][1]) + [float(lp)], float(sc)))
                sources.append(j)
                if len(newBeams) == self.beamSize:
                    cut = r_i
                    break
        vals = [float(scores[q][0]) for q in ranked[: (cut + 2 if cut is not None else len(ranked))]]
        for a, b in zip(vals, vals[1:]):
            if math.isfinite(a) and math.isfinite(b):
                self.minMargin = min(self.minMargin, a - b)
        for seq in sorted(finished, key=lambda q: finished[q][0], reverse=True):                 # :391-396
            if len(self.finishedSequences) >= self.maxCandidates:
                break
            self.finishedSequences[seq] = finished[seq]
        return newBeams, sources, len(self.finishedSequences) >= self.maxCandidates             # :398-402 (one audio)

    def finalize(self, beams: List[Tuple[List[int], List[float], float]]):
        """:406-424: when fewer than beamSize sequences finished, the live beams follow (best sum first) with EOT appended."""
        if len(self.finishedSequences) < self.beamSize:
            sums = np.array([b[2] for b in beams], dtype=np.float32)
            for j in list(np.argsort(sums, kind="stable"))[::-1]:
                toks, lps, sm = beams[j]
                self.finishedSequences[tuple(toks) + (self.eotToken,)] = (float(np.float32(sm)), tuple(lps) + (0.0,))
                if len(self.finishedSequences) >= self.beamSize:
                    break
        return [(list(k), list(v[1]), v[0]) for k, v in self.finishedSequences.items()]

    def rank(self, candidates, sampleBegin: int) -> int:
        """MaximumLikelihoodRanker with length_penalty None (:236-255): sum of log-probs / number of sampled tokens before EOT."""
        best, bestScore = 0, -math.inf
        vals = []
        for i, (toks, _lps, sm) in enumerate(candidates):
            text = toks[sampleBegin:]
            if self.eotToken in text:
                text = text[: text.index(self.eotToken)]
            sc = float(np.float32(np.float32(sm) / np.float32(max(len(text), 1))))
            vals.append(sc)
            if sc > bestScore:
                best, bestScore = i, sc
        o = sorted(vals, reverse=True)
        if len(o) > 1 and math.isfinite(o[0]) and math.isfinite(o[1]):
            self.minMargin = min(self.minMargin, o[0] - o[1])
        return best


def decode_text_beam(new_state: Callable[[], object], initialPrompt: List[int], beamSize: int, patience: float,
                     options: DecodingOptions, st: SpecialTokens, isModelMultilingual: bool, languageTokens: Sequence[int] = (),
                     prefilledIndex: int = 0, sampler_out: Optional[list] = None) -> DecodingResult:
    """Beam search at T = 0 for ONE audio on the reference's decodeText skeleton (Core/TextDecoder.swift:541-855): the prompt is
    pre-filled exactly like decode_text (one decoder call per prompt token, greedy prediction kept for the last-prompt-timestamp
    replacement and the first-token threshold), then every position expands the beams with BeamSearchTokenSampler.
    new_state(): a fresh decoder state with .step(token, pos) -> logits and .copy_from(other) (key/value cache copy: openai's
    rearrange_kv_cache).  NO REFERENCE BEHAVIOUR (see BeamSearchTokenSampler)."""
    initialPromptIndex = len(initialPrompt)
    currentTokens = list(initialPrompt)
    nextToken = initialPrompt[-1]
    greedy = GreedyTokenSampler(0.0, st.endToken, options)
    beam = BeamSearchTokenSampler(beamSize, st.endToken, patience)
    if sampler_out is not None:
        sampler_out.append(beam)
    loopCount = min(options.sampleLength, MAX_TOKEN_CONTEXT - 1)
    filters = create_logits_filters(options, prefilledIndex, initialPromptIndex, st, isModelMultilingual)
    state0 = new_state()
    isFirstTokenLogProbTooLow = False
    steps = 0
    early = False

    def filtered(state, token, tokenIndex, tokens):
        logits = np.array(state.step(token, tokenIndex), dtype=np.float32, copy=True)
        if options.float16Logits:
            logits = logits.astype(np.float16).astype(np.float32)
        for f in filters:
            logits = f.filterLogits(logits, tokens)
        return logits

    def resolve(tokenIndex):                                                     # :581-594, the loop top of decode_text
        nonlocal nextToken
        if tokenIndex < initialPromptIndex:
            isTimestampToken = currentTokens[tokenIndex] >= st.timeTokenBegin
            modelPredictedTimestamp = nextToken >= st.timeTokenBegin
            if not (tokenIndex == initialPromptIndex - 1 and isTimestampToken and modelPredictedTimestamp):
                nextToken = currentTokens[tokenIndex]
            else:
                currentTokens[tokenIndex] = nextToken

    for tokenIndex in range(prefilledIndex, min(initialPromptIndex - 1, loopCount)):       # pre-fill: decode_text's greedy steps
        resolve(tokenIndex)
        logits = filtered(state0, nextToken, tokenIndex, currentTokens)
        steps += 1
        tok, lp = greedy.sample(logits, counter=tokenIndex)
        nextToken = tok
        isFirstTokenLogProbTooLow = bool(tokenIndex == prefilledIndex and options.firstTokenLogProbThreshold is not None
                                         and lp < options.firstTokenLogProbThreshold)
        if tok == st.endToken or len(currentTokens) >= MAX_TOKEN_CONTEXT - 1 or isFirstTokenLogProbTooLow:
            early = True
            break
    chosen = (list(currentTokens), [0.0] * len(currentTokens), 0.0)
    if not early and initialPromptIndex - 1 < loopCount:
        resolve(initialPromptIndex - 1)
        beams = [(list(currentTokens), [0.0] * len(currentTokens), 0.0) for _ in range(beamSize)]
        states = [state0] + [new_state() for _ in range(beamSize - 1)]
        for sj in states[1:]:
            sj.copy_from(state0)
        nextTokens = [nextToken] * beamSize
        live = True
        spare = None
        for tokenIndex in range(initialPromptIndex - 1, loopCount):
            rows = []
            for j in range(len(beams)):
                logits = filtered(states[j], nextTokens[j], tokenIndex, beams[j][0]).astype(np.float64)
                m = logits.max()
                lse = m + math.log(float(np.exp(logits - m).sum()))
                rows.append((logits - lse).astype(np.float32))
            steps += 1
            if tokenIndex == prefilledIndex and options.firstTokenLogProbThreshold is not None:
                isFirstTokenLogProbTooLow = bool(float(rows[0].max()) < options.firstTokenLogProbThreshold)
                if isFirstTokenLogProbTooLow:
                    live = False
                    break
            if len(beams[0][0]) >= MAX_TOKEN_CONTEXT - 1:
                break
            beams, sources, completed = beam.update(beams, rows)
            if spare is None:
                spare = [new_state() for _ in range(beamSize)]       # rearrange_kv_cache: copy into the other pool, swap
            for sj, src in zip(spare, sources):
                sj.copy_from(states[src])
            states, spare = spare, states
            nextTokens = [b[0][-1] for b in beams]
            if completed:
                break
        if live:
            cands = beam.finalize(beams)
            chosen = cands[beam.rank(cands, initialPromptIndex)]
    segmentTokens, segmentLogProbs = greedy.finalize(chosen[0], chosen[1])
    startIndex = segmentTokens.index(st.startOfTranscriptToken) if st.startOfTranscriptToken in segmentTokens else 0
    endIndex = segmentTokens.index(st.endToken) if st.endToken in segmentTokens else len(segmentTokens)
    filteredTokens = segmentTokens[startIndex: endIndex + 1]
    filteredLogProbs = segmentLogProbs[startIndex: endIndex + 1]
    sacc = np.float32(0)
    for v in filteredLogProbs:
        sacc = np.float32(sacc + np.float32(v))
    avgLogProbs = float(sacc / np.float32(len(filteredLogProbs)))
    wordTokens = [t for t in filteredTokens if t < st.specialTokenBegin]
    finalCompressionRatio = compression_ratio(wordTokens)
    language = options.language or "en"
    if options.language is None:
        language = "en"
        for t in filteredTokens:
            if t in set(languageTokens):
                language = f"<lang:{t}>"
                break
    fallback = decoding_fallback(options, isFirstTokenLogProbTooLow, 0.0, finalCompressionRatio, avgLogProbs)
    return DecodingResult(language=language, tokens=filteredTokens,
                          tokenLogProbs=[{t: float(np.float32(l))} for t, l in zip(filteredTokens, filteredLogProbs)],
                          avgLogProb=avgLogProbs, noSpeechProb=0.0, temperature=0.0, compressionRatio=finalCompressionRatio,
                          fallback=fallback, alignment=None, isFirstTokenLogProbTooLow=isFirstTokenLogProbTooLow, steps=steps)


def detect_language(step: StepFn, sampler: GreedyTokenSampler, st: SpecialTokens, languageTokens: Sequence[int],
                    logitsSize: int) -> Tuple[int, float]:
    """Core/TextDecoder.swift:420-539: one step on SOT at position 0, LanguageLogitsFilter, sample.
    Returns (language token id, logprob)."""
    currentTokens = [st.startOfTranscriptToken]
    logits = np.array(step(currentTokens[0], 0), dtype=np.float32, copy=True)
    logits = LanguageLogitsFilter(languageTokens, logitsSize, sampleBegin=0).filterLogits(logits, currentTokens)
    tok, lp = sampler.sample(logits, counter=0)
    return tok, lp


def fallback_temperatures(options: DecodingOptions) -> List[float]:
    """Core/TranscribeTask.swift:327 - built in FloatType (Float16 on arm64)."""
    t0, inc = np.float16(options.temperature), np.float16(options.temperatureIncrementOnFallback)
    return [float(np.float16(t0 + np.float16(i) * inc)) for i in range(options.temperatureFallbackCount + 1)]


# ----------------------------------------------------------------------------- segments
def find_seek_point_and_segments(res: DecodingResult, options: DecodingOptions, allSegmentsCount: int,
                                 currentSeek: int, segmentSize: int, st: SpecialTokens,
                                 decode_fn: Optional[Callable[[List[int]], str]] = None,
                                 sampleRate: int = SAMPLE_RATE):
    """Core/Text/SegmentSeeker.swift:41-189."""
    timeToken = st.timeTokenBegin
    decode_fn = decode_fn or (lambda toks: "")
    seek = currentSeek
    f32 = np.float32
    timeOffset = f32(seek) / f32(sampleRate)
    spt = f32(SECONDS_PER_TIME_TOKEN)
    if options.noSpeechThreshold is not None:
        shouldSkip = res.noSpeechProb > options.noSpeechThreshold
        if options.logProbThreshold is not None and res.avgLogProb > options.logProbThreshold:
            shouldSkip = False
        if shouldSkip:
            return seek + segmentSize, None
    segs: List[TranscriptionSegment] = []
    cur, curLP = res.tokens, res.tokenLogProbs
    isTs = [t >= timeToken for t in cur]
    last3 = isTs[-3:]
    singleTimestampEnding = last3 == [False, True, False]
    noTimestampEnding = last3 == [False, False, False]
    sliceIndexes = []
    prev = False
    for i, c in enumerate(isTs):
        if prev and c:
            sliceIndexes.append(i)
        prev = c

    def mk(tokens, lps, start, end):
        wordTokens = [t for t in tokens if t < st.specialTokenBegin]
        text = decode_fn(wordTokens if options.skipSpecialTokens else tokens)
        return TranscriptionSegment(id=allSegmentsCount + len(segs), seek=seek0, start=float(start), end=float(end),
                                    text=text, tokens=list(tokens), tokenLogProbs=list(lps), temperature=res.temperature,
                                    avgLogprob=res.avgLogProb, compressionRatio=res.compressionRatio,
                                    noSpeechProb=res.noSpeechProb)

    seek0 = seek
    if sliceIndexes:
        if singleTimestampEnding:
            sliceIndexes.append(max(i for i, c in enumerate(isTs) if c) + 1)
        elif noTimestampEnding:
            sliceIndexes.append(len(cur))
        lastSliceStart = 0
        for currentSliceEnd in sliceIndexes:
            sl, slp = cur[lastSliceStart:currentSliceEnd], curLP[lastSliceStart:currentSliceEnd]
            tts = [t for t in sl if t >= timeToken]
            startS = f32(tts[0] - timeToken) * spt
            endS = f32(tts[-1] - timeToken) * spt
            segs.append(mk(sl, slp, timeOffset + startS, timeOffset + endS))
            lastSliceStart = currentSliceEnd
        if not noTimestampEnding:
            lastTimestampToken = cur[lastSliceStart - (1 if singleTimestampEnding else 0)] - timeToken
            lastTimestampSeconds = f32(lastTimestampToken) * spt
            seek += int(lastTimestampSeconds * f32(sampleRate))
        else:
            seek += segmentSize
    else:
        durationSeconds = f32(segmentSize) / f32(sampleRate)
        tts = [t for t in cur if t > timeToken]
        if tts:
            durationSeconds = f32(tts[-1] - timeToken) * spt
        segs.append(mk(cur, curLP, timeOffset, timeOffset + durationSeconds))
        seek += segmentSize
    return seek, segs


# ----------------------------------------------------------------------------- DTW / word timestamps
def dynamic_time_warping(matrix: np.ndarray) -> Tuple[List[int], List[int]]:
    """Core/Text/SegmentSeeker.swift:195-278: DP over -matrix in Double with the reference's exact
    tie-breaking (strict `<` for diagonal and up; otherwise left) and backtrace."""
    m = -np.asarray(matrix, dtype=np.float64)
    n_rows, n_cols = m.shape
    cost = np.full((n_rows + 1, n_cols + 1), np.inf)
    trace = np.full((n_rows + 1, n_cols + 1), -1, dtype=np.int64)
    cost[0, 0] = 0
    trace[0, 1:] = 2
    trace[1:, 0] = 1
    for r in range(1, n_rows + 1):
        row_prev, row_cur, mv = cost[r - 1], cost[r], m[r - 1]
        tr = trace[r]
        for c in range(1, n_cols + 1):
            v = mv[c - 1]
            c0, c1, c2 = row_prev[c - 1] + v, row_prev[c] + v, row_cur[c - 1] + v
            if c0 < c1 and c0 < c2:
                row_cur[c], tr[c] = c0, 0
            elif c1 < c0 and c1 < c2:
                row_cur[c], tr[c] = c1, 1
            else:
                row_cur[c], tr[c] = c2, 2
    i, j = n_rows, n_cols
    ti, tj = [], []
    while i > 0 or j > 0:
        ti.append(i - 1)
        tj.append(j - 1)
        t = trace[i, j]
        if t == 0:
            i -= 1; j -= 1
        elif t == 1:
            i -= 1
        elif t == 2:
            j -= 1
        else:
            break
    return ti[::-1], tj[::-1]


def default_split_to_word_tokens(tokenIds: List[int], st: SpecialTokens):
    """Stand-in for WhisperTokenizerWrapper.splitToWordTokens (Core/Models.swift:1226-1306) when no
    tokenizer.json exists (none in this image): every token is its own word, text is "<id>"."""
    return [f"<{t}>" for t in tokenIds], [[t] for t in tokenIds]


def find_alignment(wordTokenIds: List[int], alignmentWeights: np.ndarray, tokenLogProbs: List[float],
                   split_fn: Callable) -> List[WordTiming]:
    """Core/Text/SegmentSeeker.swift:340-408."""
    textIndices, timeIndices = dynamic_time_warping(alignmentWeights)
    words, wordTokens = split_fn(wordTokenIds)
    if len(wordTokens) <= 1:
        return []
    f32 = np.float32
    spt = f32(SECONDS_PER_TIME_TOKEN)
    startTimes, endTimes = [f32(0.0)], []
    currentTokenIndex = textIndices[0] if textIndices else 0
    for idx in range(len(textIndices)):
        if textIndices[idx] != currentTokenIndex:
            currentTokenIndex = textIndices[idx]
            t = f32(timeIndices[idx]) * spt
            startTimes.append(t)
            endTimes.append(t)
    endTimes.append(f32(timeIndices[-1] if timeIndices else 1500) * spt)
    out = []
    currentTokenIndex = 0
    for index, wta in enumerate(wordTokens):
        startIndex = currentTokenIndex
        wordStart = startTimes[currentTokenIndex]
        currentTokenIndex += len(wta) - 1
        wordEnd = endTimes[currentTokenIndex]
        currentTokenIndex += 1
        probs = tokenLogProbs[startIndex:currentTokenIndex]
        s = f32(0)
        for p in probs:
            s = f32(s + f32(p))
        prob = s / f32(len(probs))
        out.append(WordTiming(words[index], list(wta), float(wordStart), float(wordEnd), float(np.exp(f32(prob)))))
    return out


def calculate_word_duration_constraints(alignment: List[WordTiming]) -> Tuple[float, float]:
    """Core/Text/SegmentSeeker.swift:498-507."""
    durs = sorted(d for d in (w.duration for w in alignment) if d > 0)
    median = durs[len(durs) // 2] if durs else 0.0
    cm = min(np.float32(0.7), np.float32(median))
    return float(cm), float(cm * np.float32(2))


def truncate_long_words_at_sentence_boundaries(alignment: List[WordTiming], maxDuration: float) -> List[WordTiming]:
    """Core/Text/SegmentSeeker.swift:509-526."""
    marks = [".", "。", "!", "！", "?", "？"]
    out = [dataclasses.replace(w) for w in alignment]
    for i in range(1, len(out)):
        if out[i].duration > maxDuration:
            if out[i].word in marks:
                out[i].end = float(np.float32(out[i].start) + np.float32(maxDuration))
            elif out[i - 1].word in marks:
                out[i].start = float(np.float32(out[i].end) - np.float32(maxDuration))
    return out


def is_swift_whitespace(ch: str) -> bool:
    """CharacterSet.whitespaces: Unicode general category Zs plus CHARACTER TABULATION."""
    import unicodedata
    return ch == "\t" or unicodedata.category(ch) == "Zs"


def trim_whitespaces(s: str) -> str:
    """String.trimmingCharacters(in: .whitespaces)."""
    a, b = 0, len(s)
    while a < b and is_swift_whitespace(s[a]):
        a += 1
    while b > a and is_swift_whitespace(s[b - 1]):
        b -= 1
    return s[a:b]


def merge_punctuations(alignment: List[WordTiming], prepended: str = "\"'“¡¿([{-",
                       appended: str = "\"'.。,，!！?？:：”)]}、") -> List[WordTiming]:
    """Core/Text/SegmentSeeker.swift:280-338 (defaults: Constants.defaultPrepend/AppendPunctuations)."""
    if not alignment:
        return []
    al = [dataclasses.replace(w) for w in alignment]
    pre: List[WordTiming] = []
    app: List[WordTiming] = []
    if trim_whitespaces(al[0].word) not in prepended or trim_whitespaces(al[0].word) == "":
        # Swift `String.contains("")` is false for the empty string
        pre.append(al[0])
    for i in range(1, len(al)):
        cur, prev = dataclasses.replace(al[i]), al[i - 1]
        pw = trim_whitespaces(prev.word)
        if prev.word[:1] != "" and is_swift_whitespace(prev.word[0]) and pw != "" and pw in prepended:
            cur.word = prev.word + cur.word
            cur.tokens = prev.tokens + cur.tokens
            if not pre:
                pre.append(cur)
            else:
                pre[-1] = cur
        else:
            pre.append(cur)
    if pre:
        app.append(pre[0])
    for i in range(1, len(pre)):
        cur, prev = pre[i], dataclasses.replace(pre[i - 1])
        cw = trim_whitespaces(cur.word)
        if not prev.word.endswith(" ") and cw != "" and cw in appended:
            prev.word = prev.word + cur.word
            prev.tokens = prev.tokens + cur.tokens
            app[-1] = prev
        else:
            app.append(cur)
    return [w for w in app if w.word != "" and not (w.word in appended) and not (w.word in prepended)]


def update_segments_with_word_timings(segments: List[TranscriptionSegment], mergedAlignment: List[WordTiming], seek: int,
                                      lastSpeechTimestamp: float, constrainedMedianDuration: float, maxDuration: float,
                                      specialTokenBegin: int, decode_fn: Optional[Callable[[List[int]], str]] = None
                                      ) -> List[TranscriptionSegment]:
    """Core/Text/SegmentSeeker.swift:528-659 (Float arithmetic; `.rounded(2)` = _rounded).  `decode_fn` stands in for
    tokenizer.decode when a merged word loses special tokens (:552-554)."""
    f = np.float32
    timeOffset = f(seek) / f(SAMPLE_RATE)
    cmd, mx = f(constrainedMedianDuration), f(maxDuration)
    lastSpeech = f(lastSpeechTimestamp)
    wordIndex = 0
    updated: List[TranscriptionSegment] = []
    for segmentIndex, segment in enumerate(segments):
        savedTokens = 0
        textTokens = [t for t in segment.tokens if t < specialTokenBegin]
        words: List[WordTiming] = []
        while wordIndex < len(mergedAlignment) and savedTokens < len(textTokens):
            timing = mergedAlignment[wordIndex]
            wordIndex += 1
            timingTokens = [t for t in timing.tokens if t < specialTokenBegin]
            if not timingTokens:
                continue
            if len(timingTokens) < len(timing.tokens):
                if decode_fn is None:
                    raise ValueError("a merged word lost special tokens: decode_fn (tokenizer.decode) is required")
                word = decode_fn(timingTokens)
            else:
                word = timing.word
            start = f(_rounded(float(timeOffset + f(timing.start)), 2))
            end = f(_rounded(float(timeOffset + f(timing.end)), 2))
            if end - start < cmd / f(4):
                if words:
                    previousEnd = f(words[-1].end)
                    if start > previousEnd:
                        desired = min(start - previousEnd, cmd / f(2))
                        start = f(_rounded(float(start - desired), 2))
                elif segmentIndex > 0 and len(updated) > segmentIndex - 1 and start > f(updated[segmentIndex - 1].end):
                    desired = min(start - f(updated[segmentIndex - 1].end), cmd / f(2))
                    start = f(_rounded(float(start - desired), 2))
            words.append(WordTiming(word, timingTokens, float(start), float(end), _rounded(timing.probability, 2)))
            savedTokens += len(timingTokens)
        seg = dataclasses.replace(segment)
        if words:
            first = words[0]
            pauseLength = f(first.end) - lastSpeech
            firstWordTooLong = f(first.duration) > mx
            bothWordsTooLong = len(words) > 1 and f(words[1].end) - f(first.start) > mx * f(2)
            if pauseLength > cmd * f(4) and (firstWordTooLong or bothWordsTooLong):
                if len(words) > 1 and f(words[1].duration) > mx:
                    boundary = max(f(words[1].end) / f(2), f(words[1].end) - mx)
                    words[0].end = float(boundary)
                    words[1].start = float(boundary)
                words[0].start = float(max(lastSpeech, f(words[0].end) - mx))
            if f(segment.start) < f(words[0].end) and f(segment.start) - f(0.5) > f(words[0].start):
                words[0].start = float(max(f(0), min(f(words[0].end) - cmd, f(segment.start))))
            else:
                seg.start = words[0].start
            last = words[-1]
            if f(seg.end) > f(last.start) and f(segment.end) + f(0.5) < f(last.end):
                words[-1].end = float(max(f(last.start) + cmd, f(segment.end)))
            else:
                seg.end = last.end
            lastSpeech = f(seg.end)
        seg.words = words
        updated.append(seg)
    return updated


# ----------------------------------------------------------------------------- VAD / chunking
def calculate_voice_activity_in_chunks(signal: np.ndarray, chunkCount: int, frameLengthSamples: int,
                                       frameOverlapSamples: int = 0, energyThreshold: float = 0.022) -> List[bool]:
    """Core/Audio/AudioProcessor.swift:673-702 (vDSP_rmsqv per chunk > threshold)."""
    out = []
    n = len(signal)
    for i in range(chunkCount):
        s = i * frameLengthSamples
        e = min(s + frameLengthSamples + frameOverlapSamples, n)
        chunk = np.asarray(signal[s:e], dtype=np.float32)
        rms = np.float32(math.sqrt(float(np.mean(chunk.astype(np.float64) ** 2)))) if len(chunk) else np.float32(0)
        out.append(bool(rms > np.float32(energyThreshold)))
    return out


class EnergyVAD:
    """Core/Audio/EnergyVAD.swift + VoiceActivityDetector.swift."""
    def __init__(self, sampleRate: int = SAMPLE_RATE, frameLength: float = 0.1, frameOverlap: float = 0.0,
                 energyThreshold: float = 0.02, frameLengthSamples: Optional[int] = None,
                 frameOverlapSamples: Optional[int] = None):
        self.sampleRate = sampleRate
        self.frameLengthSamples = frameLengthSamples if frameLengthSamples is not None else int(np.float32(frameLength) * np.float32(sampleRate))
        self.frameOverlapSamples = frameOverlapSamples if frameOverlapSamples is not None else int(np.float32(frameOverlap) * np.float32(sampleRate))
        self.energyThreshold = energyThreshold

    def voiceActivity(self, waveform) -> List[bool]:
        count = int(math.ceil(len(waveform) / self.frameLengthSamples))
        return calculate_voice_activity_in_chunks(waveform, count, self.frameLengthSamples, self.frameOverlapSamples,
                                                  self.energyThreshold)

    def calculateActiveChunks(self, waveform) -> List[Tuple[int, int]]:
        vad = self.voiceActivity(waveform)
        result: List[List[int]] = []
        cur = None
        for i, v in enumerate(vad):
            if v:
                cs = i * self.frameLengthSamples
                ce = min(cs + self.frameLengthSamples, len(waveform))
                if cur is not None:
                    result[-1][1] = ce
                else:
                    cur = cs
                    result.append([cs, ce])
            else:
                cur = None
        return [(a, b) for a, b in result]

    def voiceActivityIndexToAudioSampleIndex(self, index: int) -> int:
        return index * self.frameLengthSamples

    def voiceActivityIndexToSeconds(self, index: int) -> float:
        return float(np.float32(self.voiceActivityIndexToAudioSampleIndex(index)) / np.float32(self.sampleRate))

    @staticmethod
    def findLongestSilence(vadResult: Sequence[bool]) -> Optional[Tuple[int, int]]:
        best, bestCount, i = None, 0, 0
        while i < len(vadResult):
            if vadResult[i]:
                i += 1
            else:
                e = i
                while e < len(vadResult) and not vadResult[e]:
                    e += 1
                if e - i > bestCount:
                    bestCount, best = e - i, (i, e)
                i = e
        return best

    def voiceActivityClipTimestamps(self, waveform) -> List[float]:
        out = []
        for s, e in self.calculateActiveChunks(waveform):
            out += [float(np.float32(s) / np.float32(self.sampleRate)), float(np.float32(e) / np.float32(self.sampleRate))]
        return out

    def calculateNonSilentSeekClips(self, waveform) -> List[Tuple[int, int]]:
        return DecodingOptions(clipTimestamps=self.voiceActivityClipTimestamps(waveform)).prepareSeekClips(len(waveform))


def vad_chunk_all(audio: np.ndarray, maxChunkLength: int = WINDOW_SAMPLES, options: Optional[DecodingOptions] = None,
                  windowPadding: int = 16000, vad: Optional[EnergyVAD] = None) -> List[Tuple[int, np.ndarray]]:
    """Core/Audio/AudioChunker.swift:43-107 -> [(seekOffsetIndex, samples)]."""
    vad = vad or EnergyVAD()
    n = len(audio)
    if n <= maxChunkLength:
        return [(0, audio)]
    options = options or DecodingOptions()
    out = []
    for clipStart, clipEnd in options.prepareSeekClips(n):
        startIndex = clipStart
        while startIndex < clipEnd - windowPadding:
            if not (0 <= startIndex < n):
                raise ValueError("startIndex is outside the buffer size")
            endIndex = clipEnd
            if startIndex + maxChunkLength < endIndex:
                e = min(n, startIndex + maxChunkLength)
                mid = startIndex + (e - startIndex) // 2
                va = vad.voiceActivity(audio[mid:e])
                sil = vad.findLongestSilence(va)
                if sil is not None:
                    silMid = sil[0] + (sil[1] - sil[0]) // 2
                    endIndex = mid + vad.voiceActivityIndexToAudioSampleIndex(silMid)
                else:
                    endIndex = e
            if not endIndex > startIndex:
                break
            out.append((startIndex, audio[startIndex:endIndex]))
            startIndex = endIndex
    return out


# ----------------------------------------------------------------------------- TranscribeTask.run
@dataclasses.dataclass
class TranscriptionResult:
    segments: List[TranscriptionSegment]
    tokens: List[int]
    language: str
    windows: int = 0
    fallbacks: int = 0
    seeks: List[int] = dataclasses.field(default_factory=list)
    temperatures: List[float] = dataclasses.field(default_factory=list)
    text: str = ""


def windowing(res: DecodingResult, options: DecodingOptions, allSegmentsCount: int, seek: int, segmentSize: int, st: SpecialTokens,
              tokenizer=None, language: str = "en"):
    """The "Windowing" block of TranscribeTask.run, Core/TranscribeTask.swift:175-241: findSeekPointAndSegments, seek never moves
    backward, optional addWordTimestamps (then zero-length segments are dropped and the seek is refined with the last word's
    end), maxWindowSeek clamp.  `tokenizer` is an oracle.tokenizer.Tokenizer (None: no text, no word timestamps).
    Returns (seek, segments or None)."""
    f32 = np.float32
    previousSeek = seek
    decode_fn = tokenizer.decode if tokenizer is not None else None
    newSeek, cur = find_seek_point_and_segments(res, options, allSegmentsCount, seek, segmentSize, st, decode_fn)
    seek = max(seek, newSeek)
    if options.wordTimestamps and res.alignment is not None and tokenizer is not None:
        cur = add_word_timestamps(cur or [], res.alignment, lambda ids: tokenizer.splitToWordTokens(ids, language), tokenizer.decode,
                                  previousSeek, float(f32(np.float64(previousSeek) / np.float64(SAMPLE_RATE))), st.specialTokenBegin)
        cur = [g for g in cur if f32(g.end) > f32(g.start)]
        if cur:
            seek = max(seek, int(f32(cur[-1].end) * f32(SAMPLE_RATE)))
    if options.maxWindowSeek is not None:
        seek = min(seek, previousSeek + options.maxWindowSeek)
    return seek, cur


def transcribe_task_run(audio: np.ndarray, options: Optional[DecodingOptions], st: SpecialTokens,
                        isModelMultilingual: bool, languageTokens: Sequence[int], logitsSize: int,
                        encode_window: Callable[[np.ndarray], object],
                        make_step: Callable[[object], StepFn],
                        seed: int = 0, get_alignment: Optional[Callable[[], np.ndarray]] = None,
                        split_fn: Optional[Callable] = None, tokenizer=None,
                        records: Optional[list] = None, make_state: Optional[Callable[[object], object]] = None) -> TranscriptionResult:
    """Core/TranscribeTask.swift:57-296 (window loop) + :316-411 (decodeWithFallback).

    encode_window(pcm[480000]) -> opaque encoder output (padOrTrim + logMel + encode, :126-151)
    make_step(encoder_output)  -> fresh StepFn with reset decoder state (decoderInputs.reset, :271,398)
    make_state(encoder_output) -> fresh decoder state object (.step, .copy_from): needed for options.beamSize > 1, whose T = 0 pass
                                  is decode_text_beam (no reference behaviour; T > 0 fallbacks sample as the reference does)
    """
    options = options or DecodingOptions()
    contentFrames = len(audio)
    allSegments: List[TranscriptionSegment] = []
    allTokens: List[int] = []
    detectedLanguage = None
    prompt = [st.startOfTranscriptToken]
    if options.usePrefillPrompt:
        prompt = prefill_prompt(options, st, isModelMultilingual)
    result = TranscriptionResult([], [], "en")
    for clipStart, clipEnd in options.prepareSeekClips(contentFrames):
        seek = clipStart
        windowPadding = int(np.float32(options.windowClipTime) * np.float32(SAMPLE_RATE))
        while seek < clipEnd - windowPadding:
            segmentSize = min(WINDOW_SAMPLES, contentFrames - seek, clipEnd - seek)
            pcm = np.zeros(WINDOW_SAMPLES, dtype=np.float32)
            pcm[:segmentSize] = audio[seek:seek + segmentSize]
            enc = encode_window(pcm)
            result.seeks.append(seek)
            # ---- decodeWithFallback
            res = None
            for i, temp in enumerate(fallback_temperatures(options)):
                sampler = GreedyTokenSampler(temp, st.endToken, options, seed=seed + 1000003 * result.windows + i)
                curOptions = options
                step = make_step(enc)
                if isModelMultilingual and options.language is None and options.detectLanguage:
                    ltok, _ = detect_language(step, sampler, st, languageTokens, logitsSize)
                    detectedLanguage = f"<lang:{ltok}>" if tokenizer is None else (tokenizer.decode([ltok]).strip("<|>") or "en")
                    if options.usePrefillPrompt:
                        prompt = prefill_prompt(curOptions, st, isModelMultilingual, languageToken=ltok)
                    step = make_step(enc)
                rec = [] if records is not None else None
                if options.beamSize > 1 and sampler.temperature == 0.0 and make_state is not None:
                    beam_samplers = []
                    res = decode_text_beam(lambda: make_state(enc), prompt, options.beamSize, options.beamPatience, curOptions, st,
                                           isModelMultilingual, languageTokens, sampler_out=beam_samplers)
                    rec = beam_samplers
                else:
                    res = decode_text(step, prompt, sampler, curOptions, st, isModelMultilingual, languageTokens,
                                      alignment=None, record_logits=rec)
                if records is not None:     # test hook: the filtered logits of every sampling step of this decode
                    records.append(dict(seek=seek, temperature=sampler.temperature, seed=sampler.seed, prompt=list(prompt),
                                        record=rec, result=res))
                if get_alignment is not None:
                    res.alignment = get_alignment()
                result.temperatures.append(res.temperature)
                if detectedLanguage is None:
                    detectedLanguage = res.language
                    if tokenizer is not None:     # decodeText: first language token of the result, decoded (TextDecoder.swift:805-822)
                        lt = next((t for t in res.tokens if t in set(languageTokens)), None)
                        detectedLanguage = (tokenizer.decode([lt]).strip("<|>") if lt is not None else "") or "en"
                if res.fallback is not None and res.fallback.needsFallback:
                    result.fallbacks += 1
                else:
                    break
            # ---- windowing
            language = "en"
            if tokenizer is not None:
                lt = next((t for t in res.tokens if t in set(languageTokens)), None)
                if lt is not None:
                    language = tokenizer.decode([lt]).strip("<|>") or "en"
            seek, cur = windowing(res, options, len(allSegments), seek, segmentSize, st, tokenizer, language)
            if cur is None:
                continue
            allSegments += cur
            for s in cur:
                allTokens += s.tokens
            result.windows += 1
    result.segments, result.tokens, result.language = allSegments, allTokens, detectedLanguage or "en"
    if tokenizer is not None:      # finalizeTranscriptionResult, TranscribeTask.swift:297-312
        result.text = trim_whitespaces(tokenizer.decode([t for t in allTokens if t < st.specialTokenBegin]))
    return result


# ----------------------------------------------------------------------------- WhisperKit.transcribe(audioArray:) with .vad chunking
def update_segment_timings(segment: TranscriptionSegment, seekTime: float) -> TranscriptionSegment:
    """Utilities/TranscriptionUtilities.swift:55-69 (Float arithmetic)."""
    seg = dataclasses.replace(segment)
    st32 = np.float32(seekTime)
    seg.seek = segment.seek + int(st32 * np.float32(SAMPLE_RATE))
    seg.start = float(np.float32(segment.start) + st32)
    seg.end = float(np.float32(segment.end) + st32)
    if getattr(segment, "words", None):
        seg.words = [dataclasses.replace(w, start=float(np.float32(w.start) + st32), end=float(np.float32(w.end) + st32))
                     for w in segment.words]
    return seg


def transcribe_vad_chunked(audio: np.ndarray, options: Optional[DecodingOptions], transcribe_one: Callable[[np.ndarray, DecodingOptions], TranscriptionResult]
                           ) -> List[Tuple[float, TranscriptionResult]]:
    """Core/WhisperKit.swift:867-931 with chunkingStrategy == .vad, and AudioChunking.updateSeekOffsetsForResults
    (Core/Audio/AudioChunker.swift:14-39): audio longer than one window is cut at the middle of the longest silence of
    each window's second half, every chunk is transcribed independently with clipTimestamps reset, and the segments are
    shifted by the chunk's seek offset.  Returns [(seekTime, result)] in chunk order."""
    options = options or DecodingOptions()
    if len(audio) <= WINDOW_SAMPLES:
        return [(0.0, transcribe_one(audio, options))]
    chunks = vad_chunk_all(audio, WINDOW_SAMPLES, options)
    chunked = dataclasses.replace(options, clipTimestamps=[])
    out = []
    for seekOffsetIndex, samples in chunks:
        res = transcribe_one(samples, chunked)
        seekTime = float(np.float32(seekOffsetIndex) / np.float32(SAMPLE_RATE))
        res = dataclasses.replace(res, segments=[update_segment_timings(g, seekTime) for g in res.segments])
        out.append((seekTime, res))
    return out


# ----------------------------------------------------------------------------- SegmentSeeker.addWordTimestamps
def add_word_timestamps(segments: List[TranscriptionSegment], alignmentWeights: np.ndarray, split_fn: Callable,
                        decode_fn: Optional[Callable[[List[int]], str]], seek: int, lastSpeechTimestamp: float,
                        specialTokenBegin: int) -> List[TranscriptionSegment]:
    """Core/Text/SegmentSeeker.swift:410-496.  `split_fn(tokenIds) -> (words, wordTokens)` is tokenizer.splitToWordTokens,
    `decode_fn` tokenizer.decode; row r of `alignmentWeights` belongs to the r-th token of the segments in order
    (filteredIndices = index + indexOffset, :427-437)."""
    wordTokenIds: List[int] = []
    logProbs: List[float] = []
    for g in segments:
        for i, t in enumerate(g.tokens):
            wordTokenIds.append(t)
            lp = g.tokenLogProbs[i]
            logProbs.append(lp[t] if isinstance(lp, dict) else lp)
    filtered = np.asarray(alignmentWeights, dtype=np.float32)[:len(wordTokenIds)]
    if len(filtered) < len(wordTokenIds):      # the appended EOT can be token 225 of a 224-row matrix: zero rows
        filtered = np.vstack([filtered, np.zeros((len(wordTokenIds) - len(filtered), filtered.shape[1]), np.float32)])
    alignment = find_alignment(wordTokenIds, filtered, logProbs, split_fn) if wordTokenIds else []
    median, maxDuration = calculate_word_duration_constraints(alignment)
    alignment = truncate_long_words_at_sentence_boundaries(alignment, maxDuration)
    if alignment:
        alignment = merge_punctuations(alignment)
    return update_segments_with_word_timings(segments, alignment, seek, lastSpeechTimestamp, median, maxDuration,
                                             specialTokenBegin, decode_fn)


def compression_ratio_text(text: str) -> float:
    """TextUtilities.compressionRatio(of: String), Utilities/TextUtilities.swift:33-52 (raw DEFLATE level 5 like the token version)."""
    if text == "":
        return float("inf")
    data = text.encode("utf-8")
    c = zlib.compressobj(5, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return float(np.float32(len(data)) / np.float32(len(comp)))


# ----------------------------------------------------------------------------- result assembly and formats
def format_time(seconds: float, alwaysIncludeHours: bool, decimalMarker: str) -> str:
    """ResultWriting.formatTime, Utilities/ResultWriter.swift:14-26 (Float arithmetic)."""
    f = np.float32
    s = f(seconds)
    hrs = int(s / f(3600))
    mins = int(f(math.fmod(float(s), 3600.0)) / f(60))
    secs = int(f(math.fmod(float(s), 60.0)))
    msec = int(f(s - f(math.floor(float(s)))) * f(1000))
    if alwaysIncludeHours or hrs > 0:
        return f"{hrs:02d}:{mins:02d}:{secs:02d}{decimalMarker}{msec:03d}"
    return f"{mins:02d}:{secs:02d}{decimalMarker}{msec:03d}"


def srt_text(segments: List[TranscriptionSegment]) -> str:
    """WriteSRT.write, Utilities/ResultWriter.swift:70-100."""
    out, index = "", 1
    for g in segments:
        cues = [(w.start, w.end, w.word) for w in g.words] if g.words else [(g.start, g.end, g.text)]
        for a, b, text in cues:
            out += f"{index}\n{format_time(a, True, ',')} --> {format_time(b, True, ',')}\n{text}\n\n"
            index += 1
    return out


def vtt_text(segments: List[TranscriptionSegment]) -> str:
    """WriteVTT.write, Utilities/ResultWriter.swift:103-134."""
    out = "WEBVTT\n\n"
    for g in segments:
        cues = [(w.start, w.end, w.word) for w in g.words] if g.words else [(g.start, g.end, g.text)]
        for a, b, text in cues:
            out += f"{format_time(a, False, '.')} --> {format_time(b, False, '.')}\n{text}\n\n"
    return out


def merge_transcription_results(results: List[Optional[dict]], confirmedWords: Optional[List[str]] = None) -> dict:
    """TranscriptionUtilities.mergeTranscriptionResults, Utilities/TranscriptionUtilities.swift:76-157.  A result is a dict
    {text, segments, language, timings: {...}, seekTime}; timings keys follow wh_timings' snake_case names."""
    text = "".join(confirmedWords) if confirmedWords is not None else " ".join((r["text"] if r else "") for r in results)
    valid = [r for r in results if r is not None]
    segments = []
    for ri, r in enumerate(valid):
        for si, g in enumerate(r["segments"]):
            segments.append(dataclasses.replace(g, id=ri + si))
    language = valid[0]["language"] if valid else "en"
    T = [r["timings"] for r in valid]
    g = lambda k: [t.get(k, 0.0) for t in T]
    earliestStart = min(g("pipeline_start")) if T else 0.0
    earliestToken = min(g("first_token_time")) if T else 0.0
    latestEnd = max((t.get("pipeline_start", 0.0) + t.get("full_pipeline", 0.0)) for t in T) if T else 0.0
    merged = {}
    for k in ("model_loading", "prewarm_load_time", "encoder_load_time", "decoder_load_time", "tokenizer_load_time"):
        merged[k] = max(g(k)) if T else 0.0
    for k in ("audio_loading", "audio_processing", "logmels", "encoding", "decoding_init", "decoding_loop", "decoding_predictions",
              "decoding_filtering", "decoding_sampling", "decoding_fallback", "decoding_windowing", "decoding_kv_caching",
              "decoding_word_timestamps", "decoding_non_prediction", "total_audio_processing_runs", "total_logmel_runs",
              "total_encoding_runs", "total_decoding_loops", "total_kv_update_runs", "total_timestamp_alignment_runs",
              "total_decoding_fallbacks", "total_decoding_windows", "input_audio_seconds"):
        merged[k] = sum(g(k))
    merged["full_pipeline"] = min(latestEnd - earliestStart, sum(g("full_pipeline")))
    merged["pipeline_start"] = earliestStart
    merged["first_token_time"] = earliestToken
    return {"text": text, "segments": segments, "language": language, "timings": merged, "seekTime": None}


# ----------------------------------------------------------------------------- audio ingest
def convert_to_mono(channels: np.ndarray, mode: str = "sumChannels", indices: Optional[Sequence[int]] = None) -> np.ndarray:
    """AudioProcessor.convertToMono, Core/Audio/AudioProcessor.swift:525-625 (vDSP float32: sum in channel order, then scale by
    maxOriginalPeak / max(monoPeak, 1e-4))."""
    x = np.asarray(channels, dtype=np.float32)
    n = x.shape[0]
    if n <= 1:
        return x[0].copy()
    if mode == "specificChannel":
        c = indices[0] if indices else 0
        return x[c if 0 <= c < n else 0].copy()
    sel = [c for c in indices if 0 <= c < n] if indices else list(range(n))
    if indices and not sel:
        return x[0].copy()
    peak = np.float32(max(np.abs(x[c]).max() if x.shape[1] else 0.0 for c in sel))
    mono = np.zeros(x.shape[1], np.float32)
    for c in sel:
        mono = (mono + x[c]).astype(np.float32)
    monoPeak = np.float32(np.abs(mono).max() if len(mono) else 0.0)
    scale = np.float32(peak / max(monoPeak, np.float32(0.0001)))
    return (mono * scale).astype(np.float32)


def load_wav_16k_mono(path: str, startTime: float = 0.0, endTime: Optional[float] = None) -> np.ndarray:
    """AudioProcessor.loadAudio for the directly readable case (16 kHz mono PCM16 WAV, Core/Audio/AudioProcessor.swift:262-274):
    frames [Int(start*rate), min(Int(end*rate), length)) scaled by 1/32768 (AVAudioFile .pcmFormatFloat32)."""
    import wave
    with wave.open(path, "rb") as w:
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        n = w.getnframes()
        pcm = np.frombuffer(w.readframes(n), dtype="<i2")
    a = int(startTime * 16000)
    b = n if endTime is None else min(int(endTime * 16000), n)
    return (pcm[a:b].astype(np.float32) / np.float32(32768.0)).astype(np.float32)

"""CPU oracle for the WhisperKit hot path  --  TEST INFRASTRUCTURE ONLY.

Nothing under `whisperkit_amd/` may import this package.  It is imported only by
`tests/`, by `__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline` leg, and there
only as the checker / timed baseline, never as the thing shipped.

What it restates (reference paths are relative to /root/reference):

* `oracle.mel`     - log-mel spectrogram.  The reference has no mel arithmetic (it calls a
                     CoreML bundle: Sources/WhisperKit/Core/FeatureExtractor.swift:40-56);
                     the algorithm is openai/whisper `audio.py:log_mel_spectrogram`
                     (= transformers 5.15.0 `feature_extraction_whisper.py:95-167`).
* `oracle.model`   - Whisper encoder / decoder-step math (openai/whisper `model.py` =
                     transformers 5.15.0 `modeling_whisper.py`); the reference calls CoreML at
                     Core/AudioEncoder.swift:60 and Core/TextDecoder.swift:406.
* `oracle.decode`  - line-by-line restatement of the reference's own Swift host logic:
                     TextDecoder.decodeText / detectLanguage, LogitsFilter.swift,
                     TokenSampler.swift, Models.swift DecodingFallback, TranscribeTask.swift,
                     SegmentSeeker.swift (incl. addWordTimestamps), EnergyVAD / VADAudioChunker,
                     TextUtilities.swift, TranscriptionUtilities.mergeTranscriptionResults,
                     ResultWriter.swift (formatTime, SRT, VTT), AudioProcessor.convertToMono / WAV read.
* `oracle.tokenizer` - the vendored swift-transformers ByteLevel decode path
                     (Sources/ArgmaxCore/External/Tokenizers/Tokenizer.swift:433-525, Decoder.swift:126-165)
                     and WhisperTokenizerWrapper (Core/Models.swift:1165-1307).

PARITY PINNING STATUS
---------------------
* decode/filter/sampler/DTW/VAD/fallback logic: pinned against the reference's own
  known-answer tests (Tests/WhisperKitTests/UnitTests.swift; ported in tests/test_oracle_kats.py).
* tokenizer text: the reference's four tokenizer KATs (UnitTests.swift:1288-1375) on a fixture vocabulary that carries
  the real id -> text pairs those tests disclose (whisperkit_amd.synth.kat_tokenizer_vocab) + the HF `tokenizers`
  library as a second implementation; no real Whisper tokenizer.json exists in the image, so the rest of the real
  vocabulary is "parity unpinned".
* mel / encoder / decoder numerics: the reference pins shapes only
  (UnitTests.swift:676-693,721-732); no value of a mel bin, activation or logit is pinned
  anywhere in /root/reference, the CoreML graphs and weights are not in the repo and cannot
  run here.  These are pinned instead against the algorithm's other public implementation,
  HF transformers 5.15.0 (`WhisperFeatureExtractor`, `WhisperForConditionalGeneration`),
  imported in this container by `tests/golden/make_golden.py`, which wrote the committed
  fixtures under tests/golden/.  Parity with the CoreML path itself is therefore
  "parity unpinned" (see DESIGN.md).
"""

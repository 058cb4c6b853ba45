//  HIP backend for WhisperKit: conformers of the three model-stage protocols over libwhisperhip.so (include/whisperhip.h).
//  Source only - see ../../README.md.  Install with
//      let m = try HIPModel(path: "large-v3.whipw"); let s = try HIPSession(model: m)      // (wh_session_create_tuned; HIPSession below)
//      let config = WhisperKitConfig(featureExtractor: HIPFeatureExtractor(model: m, session: s.handle),
//                                    audioEncoder: HIPAudioEncoder(model: m, session: s.handle),
//                                    textDecoder: HIPTextDecoder(model: m, session: s.handle))
import CoreML
import Foundation
import WhisperKit
import CWhisperHIP

/// Opaque stage outputs: WhisperKit only passes them from one stage to the next
/// (marker protocols: FeatureExtractor.swift:10-11, AudioEncoder.swift:6-7).
public final class HIPStageOutput: FeatureExtractorOutputType, AudioEncoderOutputType {
    let session: OpaquePointer; let slot: Int32
    init(_ s: OpaquePointer, _ slot: Int32) { self.session = s; self.slot = slot }
}

public final class HIPModel {
    let handle: OpaquePointer
    public init(path: String, device: Int32 = 0) throws {
        var h: OpaquePointer?
        guard wh_model_load(path, device, &h) == WH_OK.rawValue, let h else {
            throw WhisperError.modelsUnavailable(String(cString: wh_last_error()))
        }
        handle = h
    }
    deinit { wh_model_destroy(handle) }
}

/// One decode session (= the reference's per-task `DecodingInputs`, Core/Models.swift:291-323) with the creation knobs of the C ABI:
/// `wh_session_create_tuned` (cross-attention mode -1 automatic / 0 per-layer K / V rows, stored as 24-bit rows = Float16 + an 8-bit residual /
/// 1 weight-absorbed over the encoder output; key splits per slot of the absorbed form = the share of the CUs one session's cross-attention
/// takes: 4 for a lone session, 128 / slots with several sessions in flight - 1 at a 128-slot device batch, 2 at 64 slots) and the window
/// hooks of TranscribeTask for the library's own orchestrator (`wh_session_set_window_hooks`: windowPreprocess / windowPostProcess /
/// segmentDiscoveryCallback, Core/TranscribeTask.swift:42-55,130,246,260).
public final class HIPSession {
    public let handle: OpaquePointer
    let model: HIPModel
    private var hookBox: Unmanaged<HookBox>?

    public enum CrossAttentionMode: Int32 { case automatic = -1, keyValueRows = 0, absorbed = 1 }
    /// slots from which `.automatic` picks the absorbed path (wh_xabs_auto_min_slots: 28 since the K / V rows carry 24 bits; WH_XABS_MIN_SLOTS overrides)
    public static var absorbedFromSlots: Int { Int(wh_xabs_auto_min_slots()) }
    /// key splits per slot an absorbed session of `maxBatch` slots gets with `keySplits: 0` (wh_xabs_auto_splits: slots x splits within one round of the
    /// 256 CUs - 4 up to 64 slots, 3 up to 85, 2 up to 128, 1 beyond; beam-search callers pass 4, callers with several sessions in flight half of it)
    public static func automaticKeySplits(maxBatch: Int) -> Int { Int(wh_xabs_auto_splits(Int32(maxBatch))) }

    /// slotsPerWorkgroup (wh_session_options, round 6): a workgroup of the absorbed cross-attention streams this many slots one after the other, so a
    /// launch takes ceil(batch / n) x keySplits workgroups whatever the batch (256-slot device batches, 1 split, 2 slots per workgroup = half of the
    /// chip: what bench.py keeps in flight three times); 0 = automatic (1).  Results do not depend on it, bit for bit.
    public init(model: HIPModel, maxBatch: Int = 1, crossAttention: CrossAttentionMode = .automatic, keySplits: Int = 0, slotsPerWorkgroup: Int = 0) throws {
        var h: OpaquePointer?
        if slotsPerWorkgroup > 0 {
            var o = wh_session_options()
            wh_session_options_default(&o)
            o.cross_attention_mode = crossAttention.rawValue
            o.cross_attention_splits = Int32(keySplits)
            o.cross_attention_slots_per_workgroup = Int32(slotsPerWorkgroup)
            try check(wh_session_create_with_options(model.handle, Int32(maxBatch), &o, &h))
        } else {
            try check(wh_session_create_tuned(model.handle, Int32(maxBatch), crossAttention.rawValue, Int32(keySplits), &h))
        }
        handle = h!; self.model = model
    }
    deinit { wh_session_set_window_hooks(handle, nil); hookBox?.release(); wh_session_destroy(handle) }

    public var crossAttentionMode: CrossAttentionMode { CrossAttentionMode(rawValue: wh_session_cross_attention_mode(handle)) ?? .automatic }
    public var crossAttentionKeySplits: Int { Int(wh_session_cross_attention_splits(handle)) }
    public var crossAttentionSlotsPerWorkgroup: Int { Int(wh_session_cross_attention_slots_per_workgroup(handle)) }
    public var capturedStepGraphs: Int { Int(wh_session_step_graph_count(handle)) }

    /// WhisperKit.transcribeWithOptions(audioArrays:decodeOptionsArray:) (Core/WhisperKit.swift:716-812) on the library's own orchestrator:
    /// one `wh_decoding_options` per audio (nil = DecodingOptions()) and one Result per audio - a failing audio is `.failure` and does not
    /// fail its neighbours (:786-790).  The transcription handles are owned by the caller (wh_transcription_free).
    public func transcribeWithOptions(audioArrays: [[Float]], options: [wh_decoding_options?], specialTokens: wh_special_tokens) throws -> [Result<OpaquePointer, WhisperError>] {
        guard audioArrays.count == options.count else { throw WhisperError.transcriptionFailed("The number of audio arrays and decoding options must be balanced.") }
        let n = audioArrays.count
        var st = specialTokens
        var outs = [OpaquePointer?](repeating: nil, count: n)
        var statuses = [Int32](repeating: 0, count: n)
        var lens = audioArrays.map { Int32($0.count) }
        let pcm = audioArrays.map { a -> UnsafeMutablePointer<Float> in
            let p = UnsafeMutablePointer<Float>.allocate(capacity: max(a.count, 1)); p.initialize(from: a, count: a.count); return p
        }
        let optStore = UnsafeMutablePointer<wh_decoding_options>.allocate(capacity: max(n, 1))
        defer { pcm.forEach { $0.deallocate() }; optStore.deallocate() }
        var optPtrs = [UnsafePointer<wh_decoding_options>?](repeating: nil, count: n)
        for i in 0..<n { if let o = options[i] { optStore[i] = o; optPtrs[i] = UnsafePointer(optStore + i) } }
        var pcmPtrs = pcm.map { UnsafePointer<Float>?($0) }
        try check(wh_transcribe_batch_with_options(handle, &pcmPtrs, &lens, Int32(n), &optPtrs, &st, &outs, &statuses))
        return (0..<n).map { i in
            if statuses[i] == WH_OK.rawValue, let t = outs[i] { return .success(t) }
            let why = String(cString: wh_session_item_error(handle, Int32(i)))
            return .failure(wh_session_item_status(handle, Int32(i)) == WH_ERR_AUDIO_PROCESSING_FAILED.rawValue ? .audioProcessingFailed(why) : .transcriptionFailed(why))
        }
    }

    final class HookBox {
        var pre: ((Int, UnsafeBufferPointer<Float>, Int, Int) -> Void)?
        var post: ((Int, Int, Int, OpaquePointer, Int, Int) -> Int?)?      // returns how many of the window's segments to keep (nil = all)
        var discovery: ((Int, OpaquePointer, Int, Int) -> Void)?
    }
    /// All nil removes the hooks.  The closures run on the thread that called wh_transcribe*; they must not throw (a Swift error cannot
    /// cross the C frames: catch inside and keep all segments, as whisperkit_amd/api.py does for Python exceptions).
    public func setWindowHooks(windowPreprocess: ((Int, UnsafeBufferPointer<Float>, Int, Int) -> Void)? = nil,
                               windowPostProcess: ((Int, Int, Int, OpaquePointer, Int, Int) -> Int?)? = nil,
                               segmentDiscovery: ((Int, OpaquePointer, Int, Int) -> Void)? = nil) throws {
        hookBox?.release(); hookBox = nil
        guard windowPreprocess != nil || windowPostProcess != nil || segmentDiscovery != nil else {
            try check(wh_session_set_window_hooks(handle, nil)); return
        }
        let box = HookBox(); box.pre = windowPreprocess; box.post = windowPostProcess; box.discovery = segmentDiscovery
        let ref = Unmanaged.passRetained(box); hookBox = ref
        var h = wh_window_hooks()
        h.user = ref.toOpaque()
        if windowPreprocess != nil {
            h.window_preprocess = { user, ai, p, seek, size in
                Unmanaged<HookBox>.fromOpaque(user!).takeUnretainedValue().pre?(Int(ai), UnsafeBufferPointer(start: p, count: Int(size)), Int(seek), Int(size))
            }
        }
        if windowPostProcess != nil {
            h.window_postprocess = { user, ai, seek, size, t, first, n in
                Int32(Unmanaged<HookBox>.fromOpaque(user!).takeUnretainedValue().post?(Int(ai), Int(seek), Int(size), t!, Int(first), Int(n)) ?? -1)
            }
        }
        if segmentDiscovery != nil {
            h.segment_discovery = { user, ai, t, first, n in
                Unmanaged<HookBox>.fromOpaque(user!).takeUnretainedValue().discovery?(Int(ai), t!, Int(first), Int(n))
            }
        }
        try check(wh_session_set_window_hooks(handle, &h))          // the struct is copied by the library
    }
}

public final class HIPFeatureExtractor: FeatureExtracting {
    let model: HIPModel; let session: OpaquePointer
    public var melCount: Int? { Int(wh_mel_count(model.handle)) }
    public var windowSamples: Int? { Int(wh_window_samples(model.handle)) }
    init(model: HIPModel, session: OpaquePointer) { self.model = model; self.session = session }

    public func logMelSpectrogram(fromAudio input: any AudioProcessorOutputType) async throws -> (any FeatureExtractorOutputType)? {
        guard let pcm = input as? [Float] else { throw WhisperError.audioProcessingFailed("expected [Float]") }
        try pcm.withUnsafeBufferPointer { p in
            try check(wh_set_audio(session, 0, p.baseAddress, Int32(p.count)))      // padOrTrim
        }
        try check(wh_log_mel_spectrogram(session, 1))
        return HIPStageOutput(session, 0)
    }
}

public final class HIPAudioEncoder: AudioEncoding {
    let model: HIPModel; let session: OpaquePointer
    public var embedSize: Int? { Int(wh_embed_size(model.handle)) }
    init(model: HIPModel, session: OpaquePointer) { self.model = model; self.session = session }
    public func encodeFeatures(_ features: any FeatureExtractorOutputType) async throws -> (any AudioEncoderOutputType)? {
        try check(wh_encode_features(session, 1))
        try check(wh_prepare_decoder_inputs(session, 1))                              // cross K/V once per window
        return HIPStageOutput(session, 0)
    }
}

/// TextDecoding (Core/TextDecoder.swift:60-105) over the C ABI: every requirement of the protocol, with the protocol's own
/// signatures.  The KV cache, masks and alignment matrix live in the wh_session (HBM), so the MLMultiArray fields of
/// `DecodingInputs` are only the carrier of `initialPrompt` / `inputIds` / `cacheLength` that TranscribeTask reads and resets
/// (Core/TranscribeTask.swift:83,271,398); `updateKVCache` has nothing left to do.
public final class HIPTextDecoder: TextDecoding {
    let model: HIPModel; let session: OpaquePointer
    public var tokenizer: WhisperTokenizer?
    public var isModelMultilingual: Bool
    public var logitsFilters: [any LogitsFiltering]?        // custom filters (TextDecoder.swift:860-862): when set - or when the sampler is not a
                                                            // GreedyTokenSampler - decodeText leaves the fused device loop for wh_decode_text_custom
    public var supportsWordTimestamps: Bool { wh_supports_word_timestamps(model.handle) != 0 }
    public var logitsSize: Int? { Int(wh_logits_size(model.handle)) }
    public var kvCacheEmbedDim: Int? { Int(wh_kv_cache_embed_dim(model.handle)) }
    public var kvCacheMaxSequenceLength: Int? { Int(wh_kv_cache_max_sequence_length(model.handle)) }
    public var windowSize: Int? { Int(wh_window_size(model.handle)) }
    public var embedSize: Int? { Int(wh_embed_size(model.handle)) }

    init(model: HIPModel, session: OpaquePointer) {
        self.model = model; self.session = session
        self.isModelMultilingual = wh_is_model_multilingual(model.handle) != 0
    }

    /// predictLogits (TextDecoder.swift:361-418): one decoder call; the session writes this step's K/V at `cacheLength` and the
    /// alignment row at cacheLength + 1 itself (updateKVCache / updateAlignmentWeights, :218-296).
    public func predictLogits(_ inputs: any TextDecoderInputType) async throws -> TextDecoderOutputType? {
        guard let inputs = inputs as? TextDecoderMLMultiArrayInputType, let V = logitsSize else {
            throw WhisperError.decodingLogitsFailed("HIPTextDecoder expects TextDecoderMLMultiArrayInputType")
        }
        var token = Int32(truncating: inputs.inputIds[0]), pos = Int32(truncating: inputs.cacheLength[0])
        let logits = try MLMultiArray(shape: [1, 1, NSNumber(value: V)], dataType: .float32)
        try check(wh_predict_logits(session, 1, &token, &pos, logits.dataPointer.assumingMemoryBound(to: Float.self)))
        return TextDecoderMLMultiArrayOutputType(logits: logits, cache: nil)
    }

    /// prepareDecoderInputs(withPrompt:) (TextDecoder.swift:109-161): DecodingInputs.reset analogue on the device + the host carrier.
    public func prepareDecoderInputs(withPrompt initialPrompt: [Int]) throws -> any DecodingInputsType {
        try check(wh_reset_decoder_inputs(session, 1))
        let one = try MLMultiArray(shape: [1], dataType: .int32), tok = try MLMultiArray(shape: [1], dataType: .int32)
        one[0] = 0; tok[0] = NSNumber(value: initialPrompt.last ?? 0)
        let empty = try MLMultiArray(shape: [1], dataType: .float16)
        return DecodingInputs(initialPrompt: initialPrompt, inputIds: tok, cacheLength: one, keyCache: empty, valueCache: empty,
                              alignmentWeights: empty, kvCacheUpdateMask: one, decoderKeyPaddingMask: empty)
    }

    /// prefillDecoderInputs (TextDecoder.swift:163-216): the forced prompt, built by the library's restatement of the same code.
    public func prefillDecoderInputs(_ decoderInputs: any DecodingInputsType, withOptions options: DecodingOptions?) async throws -> any DecodingInputsType {
        guard let tokenizer else { throw WhisperError.tokenizerUnavailable() }
        var o = cOptions(options ?? DecodingOptions(), tokenizer: tokenizer); var st = cSpecial(tokenizer.specialTokens)
        var prompt = [Int32](repeating: 0, count: 256)
        let n = wh_prefill_prompt(model.handle, &o, &st, o.language_token, &prompt, 256)
        guard n > 0 else { throw WhisperError.prefillFailed("prefill prompt does not fit") }
        var out = decoderInputs
        out.initialPrompt = prompt.prefix(Int(n)).map(Int.init)
        return out
    }

    /// decodeText (TextDecoder.swift:541-855): the whole token loop - forced prompt, predictLogits, filters, sampler, stop rules,
    /// KV / alignment update, progress callback with early stop - runs on the device behind ONE call.
    public func decodeText(from encoderOutput: any AudioEncoderOutputType, using decoderInputs: any DecodingInputsType,
                           sampler tokenSampler: TokenSampling, options decoderOptions: DecodingOptions,
                           callback: TranscriptionCallback?) async throws -> DecodingResult {
        guard let tokenizer else { throw WhisperError.tokenizerUnavailable() }
        var o = cOptions(decoderOptions, tokenizer: tokenizer); var st = cSpecial(tokenizer.specialTokens)
        let prompt = decoderInputs.initialPrompt.map(Int32.init)
        var temps = [Float((tokenSampler as? GreedyTokenSampler)?.temperature ?? 0)]
        var res = wh_decoding_result()
        let box = callback.map { CallbackBox($0, tokenizer) }
        if let box {        // TranscriptionCallback (Core/Models.swift:728): false stops this window early (earlyStopActor, :731-755)
            wh_session_set_progress_callback(session, { user, p in
                let b = Unmanaged<CallbackBox>.fromOpaque(user!).takeUnretainedValue(); let p = p!.pointee
                let toks = (0..<Int(p.n_tokens)).map { Int(p.tokens[$0]) }
                let progress = TranscriptionProgress(timings: TranscriptionTimings(), text: p.text.map { String(cString: $0) } ?? "", tokens: toks,
                                                     avgLogprob: p.avg_logprob, compressionRatio: p.compression_ratio)
                return (b.fn(progress) ?? true) ? 1 : 0
            }, Unmanaged.passUnretained(box).toOpaque())
        }
        defer { if box != nil { wh_session_set_progress_callback(session, nil, nil) } }
        if let beam = tokenSampler as? BeamSearchTokenSampler {
            // The reference's BeamSearchTokenSampler is fatalError (TokenSampler.swift:254-290) and TokenSampling.update sees one sequence:
            // the session needs beamSize slots, the library runs openai/whisper's BeamSearchDecoder (no reference behaviour).
            try check(wh_decode_text_beam(session, 1, Int32(beam.beamSize), beam.patience, &o, &st, prompt, Int32(prompt.count), nil, &res))
            return decodingResult(res, options: decoderOptions, tokenizer: tokenizer)
        }
        let userFilters = logitsFilters ?? []
        let greedy = tokenSampler as? GreedyTokenSampler
        if !userFilters.isEmpty || greedy == nil {
            // User-pluggable LogitsFiltering / TokenSampling (Core/Text/LogitsFilter.swift:8-10, TokenSampler.swift:8-11) run on the host once
            // per token (TextDecoder.swift:641-652): the library drives the step API and calls back with the host logits.
            let plug = PluginBox(filters: userFilters, sampler: greedy == nil ? tokenSampler : nil, vocab: logitsSize ?? 0)
            let ctx = Unmanaged.passUnretained(plug).toOpaque()
            let filterFn: wh_logits_filter_fn = { user, logits, n, tokens, nTokens in
                let b = Unmanaged<PluginBox>.fromOpaque(user!).takeUnretainedValue()
                b.runFilters(logits!, Int(n), (0..<Int(nTokens)).map { Int(tokens![$0]) })
            }
            let samplerFn: wh_token_sampler_fn = { user, logits, n, tokens, logprobs, nTokens, tokOut, lpOut in
                let b = Unmanaged<PluginBox>.fromOpaque(user!).takeUnretainedValue()
                let r = b.sample(logits!, Int(n), (0..<Int(nTokens)).map { Int(tokens![$0]) }, (0..<Int(nTokens)).map { logprobs![$0] })
                tokOut!.pointee = Int32(r.token); lpOut!.pointee = r.logProb
                return r.completed ? 1 : 0
            }
            var fns: [wh_logits_filter_fn?] = userFilters.isEmpty ? [] : [filterFn]      // one trampoline runs the whole custom chain in order
            var users: [UnsafeMutableRawPointer?] = userFilters.isEmpty ? [] : [ctx]
            try check(wh_decode_text_custom(session, &o, &st, prompt, Int32(prompt.count), temps[0], 0, &fns, &users, Int32(fns.count),
                                            greedy == nil ? samplerFn : nil, ctx, &res))
            if let e = plug.error { throw e }
            return decodingResult(res, options: decoderOptions, tokenizer: tokenizer)
        }
        try check(wh_decode_text(session, 1, &o, &st, prompt, Int32(prompt.count), &temps, nil, 0, &res))
        return decodingResult(res, options: decoderOptions, tokenizer: tokenizer)
    }

    /// Carrier of the caller's filter / sampler objects across the C callbacks of wh_decode_text_custom: logits travel as Float32
    /// MLMultiArrays of shape [1, 1, V], the shape TextDecoderOutputType.logits has in the reference (Core/Models.swift:1041).
    final class PluginBox {
        let filters: [any LogitsFiltering]; let sampler: TokenSampling?; let vocab: Int
        var error: Error?
        init(filters: [any LogitsFiltering], sampler: TokenSampling?, vocab: Int) { self.filters = filters; self.sampler = sampler; self.vocab = vocab }
        func wrap(_ p: UnsafePointer<Float>, _ n: Int) -> MLMultiArray? {
            guard let a = try? MLMultiArray(shape: [1, 1, NSNumber(value: n)], dataType: .float32) else { return nil }
            a.dataPointer.assumingMemoryBound(to: Float.self).update(from: p, count: n)
            return a
        }
        func runFilters(_ logits: UnsafeMutablePointer<Float>, _ n: Int, _ tokens: [Int]) {
            guard var a = wrap(logits, n) else { return }
            for f in filters { a = f.filterLogits(a, withTokens: tokens) }
            for i in 0..<n { logits[i] = a[i].floatValue }
        }
        func sample(_ logits: UnsafePointer<Float>, _ n: Int, _ tokens: [Int], _ logProbs: [Float]) -> (token: Int, logProb: Float, completed: Bool) {
            guard let sampler, let a = wrap(logits, n) else { return (0, 0, true) }
            let r = sampler.update(tokens: tokens, logits: a, logProbs: logProbs)        // SamplingResult (TokenSampler.swift:13-27)
            return (r.tokens.last ?? 0, r.logProbs.last ?? 0, r.completed)
        }
    }

    /// detectLanguage (TextDecoder.swift:420-539): one step on <|startoftranscript|>, LanguageLogitsFilter, greedy.
    public func detectLanguage(from encoderOutput: any AudioEncoderOutputType, using decoderInputs: any DecodingInputsType,
                               sampler tokenSampler: TokenSampling, options: DecodingOptions, temperature: FloatType) async throws -> DecodingResult {
        guard let tokenizer else { throw WhisperError.tokenizerUnavailable() }
        var st = cSpecial(tokenizer.specialTokens); var lang: Int32 = -1; var lp: Float = 0
        try check(wh_detect_language(session, 1, &st, &lang, &lp))
        let code = tokenizer.decode(tokens: [Int(lang)]).trimmingSpecialTokenCharacters()
        var r = DecodingResult.emptyResults
        r.language = code; r.languageProbs = [code: exp(lp)]; r.tokens = [Int(lang)]; r.tokenLogProbs = [[Int(lang): lp]]
        r.temperature = Float(temperature)
        return r
    }

    /// The cache is written in place by the decoder kernels at `token_index`; nothing to scatter on the host.
    public static func updateKVCache(keyTensor: MLMultiArray, keySlice: MLMultiArray, valueTensor: MLMultiArray, valueSlice: MLMultiArray,
                                     insertAtIndex index: Int) {}

    // ---- value conversions
    final class CallbackBox { let fn: TranscriptionCallback; let tok: WhisperTokenizer; init(_ f: @escaping TranscriptionCallback, _ t: WhisperTokenizer) { fn = f; tok = t } }

    func cSpecial(_ s: SpecialTokens) -> wh_special_tokens {
        var st = wh_special_tokens(); wh_special_tokens_default(model.handle, &st)      // language-token range of the vocabulary size
        st.end_token = Int32(s.endToken); st.english_token = Int32(s.englishToken); st.no_speech_token = Int32(s.noSpeechToken)
        st.no_timestamps_token = Int32(s.noTimestampsToken); st.special_token_begin = Int32(s.specialTokenBegin)
        st.start_of_previous_token = Int32(s.startOfPreviousToken); st.start_of_transcript_token = Int32(s.startOfTranscriptToken)
        st.time_token_begin = Int32(s.timeTokenBegin); st.transcribe_token = Int32(s.transcribeToken)
        st.translate_token = Int32(s.translateToken); st.whitespace_token = Int32(s.whitespaceToken)
        return st
    }

    func cOptions(_ d: DecodingOptions, tokenizer: WhisperTokenizer) -> wh_decoding_options {
        var o = wh_decoding_options(); wh_decoding_options_default(&o)
        o.task = d.task == .translate ? 1 : 0
        o.language_token = d.language.flatMap { tokenizer.convertTokenToId("<|\($0)|>") }.map(Int32.init) ?? -1
        o.temperature = d.temperature; o.temperature_increment_on_fallback = d.temperatureIncrementOnFallback
        o.temperature_fallback_count = Int32(d.temperatureFallbackCount); o.sample_length = Int32(d.sampleLength); o.top_k = Int32(d.topK)
        o.use_prefill_prompt = d.usePrefillPrompt ? 1 : 0; o.detect_language = d.detectLanguage ? 1 : 0
        o.skip_special_tokens = d.skipSpecialTokens ? 1 : 0; o.without_timestamps = d.withoutTimestamps ? 1 : 0
        o.word_timestamps = d.wordTimestamps ? 1 : 0; o.suppress_blank = d.suppressBlank ? 1 : 0
        o.window_clip_time = d.windowClipTime; o.max_window_seek = d.maxWindowSeek.map(Int32.init) ?? -1
        o.compression_ratio_threshold = d.compressionRatioThreshold ?? .nan; o.log_prob_threshold = d.logProbThreshold ?? .nan
        o.first_token_log_prob_threshold = d.firstTokenLogProbThreshold ?? .nan; o.no_speech_threshold = d.noSpeechThreshold ?? .nan
        o.float16_logits = 1        // FloatType == Float16 on arm64: follow the reference's own numerics
        // promptTokens / prefixTokens / suppressTokens / clipTimestamps: pointers into arrays the caller keeps alive for the call
        return o
    }

    func decodingResult(_ r: wh_decoding_result, options: DecodingOptions, tokenizer: WhisperTokenizer) -> DecodingResult {
        let n = Int(r.n_tokens)
        let toks = withUnsafeBytes(of: r.tokens) { Array($0.bindMemory(to: Int32.self).prefix(n)).map(Int.init) }
        let lps = withUnsafeBytes(of: r.token_logprobs) { Array($0.bindMemory(to: Float.self).prefix(n)) }
        let lang = r.language_token >= 0 ? tokenizer.decode(tokens: [Int(r.language_token)]).trimmingSpecialTokenCharacters() : (options.language ?? "en")
        return DecodingResult(language: lang, languageProbs: [lang: 1.0], tokens: toks, tokenLogProbs: zip(toks, lps).map { [$0: $1] },
                              text: tokenizer.decode(tokens: toks), avgLogProb: r.avg_logprob, noSpeechProb: r.no_speech_prob,
                              temperature: r.temperature, compressionRatio: r.compression_ratio, cache: nil, timings: nil,
                              fallback: DecodingFallback(options: options, isFirstTokenLogProbTooLow: r.is_first_token_logprob_too_low != 0,
                                                         noSpeechProb: r.no_speech_prob, compressionRatio: r.compression_ratio, avgLogProb: r.avg_logprob))
    }
}

@inline(__always) func check(_ rc: Int32) throws {
    if rc != WH_OK.rawValue { throw WhisperError.transcriptionFailed(String(cString: wh_last_error())) }
}

// installation (Configurations.swift:29-31)
// let m = try HIPModel(path: "large-v3.whipw"); var s: OpaquePointer?; wh_session_create(m.handle, 1, &s)
// let config = WhisperKitConfig(featureExtractor: HIPFeatureExtractor(model: m, session: s!),
//                               audioEncoder: HIPAudioEncoder(model: m, session: s!),
//                               textDecoder: HIPTextDecoder(model: m, session: s!))

public final class HIPTokenizer /* : WhisperTokenizer */ {
    let handle: OpaquePointer
    public init(tokenizerJSON: String) throws { var h: OpaquePointer?; try check(wh_tokenizer_load(tokenizerJSON, &h)); handle = h! }
    deinit { wh_tokenizer_destroy(handle) }
    public func decode(tokens: [Int]) -> String {
        let ids = tokens.map(Int32.init)
        let n = wh_tokenizer_decode(handle, ids, Int32(ids.count), 0, nil, 0)
        var buf = [CChar](repeating: 0, count: Int(n) + 1)
        wh_tokenizer_decode(handle, ids, Int32(ids.count), 0, &buf, n + 1)
        return String(decoding: buf.prefix(Int(n)).map(UInt8.init(bitPattern:)), as: UTF8.self)
    }
    public var specialTokens: SpecialTokens { var st = wh_special_tokens(); wh_tokenizer_special_tokens(handle, &st); return SpecialTokens(hip: st) }
}
// wh_session_set_tokenizer(session, tokenizer.handle): transcribe results then carry text, words and the language code

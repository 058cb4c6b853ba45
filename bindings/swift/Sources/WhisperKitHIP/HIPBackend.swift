//  HIP backend for WhisperKit: conformers of the three model-stage protocols over libwhisperhip.so (include/whisperhip.h).
//  Source only - see ../../README.md.  Install with
//      let m = try HIPModel(path: "large-v3.whipw"); var s: OpaquePointer?; wh_session_create(m.handle, 1, &s)
//      let config = WhisperKitConfig(featureExtractor: HIPFeatureExtractor(model: m, session: s!),
//                                    audioEncoder: HIPAudioEncoder(model: m, session: s!),
//                                    textDecoder: HIPTextDecoder(model: m, session: s!))
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

public final class HIPTextDecoder /* : TextDecoding */ {
    let model: HIPModel; let session: OpaquePointer
    public var logitsSize: Int? { Int(wh_logits_size(model.handle)) }
    public var kvCacheEmbedDim: Int? { Int(wh_kv_cache_embed_dim(model.handle)) }
    public var kvCacheMaxSequenceLength: Int? { Int(wh_kv_cache_max_sequence_length(model.handle)) }
    public var windowSize: Int? { Int(wh_window_size(model.handle)) }
    public var isModelMultilingual: Bool { wh_is_model_multilingual(model.handle) != 0 }
    init(model: HIPModel, session: OpaquePointer) { self.model = model; self.session = session }

    /// decodeText(from:using:sampler:options:callback:) — the whole token loop runs on the device.
    public func decodeText(options: DecodingOptions, special: SpecialTokens, temperature: Float) throws -> DecodingResult {
        var o = wh_decoding_options(); wh_decoding_options_default(&o)
        o.temperature = temperature; o.sample_length = Int32(options.sampleLength); o.top_k = Int32(options.topK)
        o.without_timestamps = options.withoutTimestamps ? 1 : 0; o.word_timestamps = options.wordTimestamps ? 1 : 0
        o.suppress_blank = options.suppressBlank ? 1 : 0
        o.compression_ratio_threshold = options.compressionRatioThreshold ?? .nan
        o.log_prob_threshold = options.logProbThreshold ?? .nan
        o.first_token_log_prob_threshold = options.firstTokenLogProbThreshold ?? .nan
        o.no_speech_threshold = options.noSpeechThreshold ?? .nan
        var st = wh_special_tokens(end_token: Int32(special.endToken), english_token: Int32(special.englishToken), /* ... */)
        var prompt = [Int32](repeating: 0, count: 256)
        let n = wh_prefill_prompt(model.handle, &o, &st, -1, &prompt, 256)
        var temps = [temperature]; var res = wh_decoding_result()
        try check(wh_decode_text(session, 1, &o, &st, prompt, n, &temps, nil, 0, &res))
        return DecodingResult(hip: res)      // tokens SOT...EOT, tokenLogProbs, avgLogProb, compressionRatio, fallback
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

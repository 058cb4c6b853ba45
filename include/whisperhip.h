/*
 * whisperhip.h - C ABI of the MI355X (gfx950) Whisper hot path.
 *
 * This is the drop-in boundary for the three model-stage protocols of argmaxinc/WhisperKit and the
 * host logic that drives them.  Every entry point cites the reference interface it replaces
 * (paths relative to /root/reference/Sources/WhisperKit).  Plain C types only: pointers, sizes,
 * PODs.  All functions return a wh_status (0 = ok) unless stated; the message of the last failure
 * on the calling thread is returned by wh_last_error().  Nothing here ever aborts the process.
 *
 * Object model (reference analogue):
 *   wh_model    immutable weights + dims, shareable across threads   (the three loaded MLModels,
 *               Core/WhisperKit.swift:372-427)
 *   wh_session  per-task decode state for up to `max_batch` 30 s windows in flight: PCM, mel,
 *               encoder output, cross-attention K/V, self-attention KV cache, alignment matrix,
 *               token/log-prob history (DecodingInputs, Core/Models.swift:291-323;
 *               one per TranscribeTask, Core/TranscribeTask.swift:83).  Bound to one HIP stream.
 *
 * Layouts handed across the boundary are the reference's logical shapes in row-major float32:
 *   mel      [n_mels][3000]        (FeatureExtractor output  [1, n_mels, 1, 3000], Core/Models.swift:877)
 *   encoder  [1500][d]             (AudioEncoder output      [1, d, 1, 1500] transposed; a Swift shim
 *                                   transposes - see INTEGRATION.md)
 *   logits   [n_vocab]             (TextDecoder output       [1, 1, n_vocab],      Core/Models.swift:1041)
 *   alignment[224][1500]           (DecodingInputs.alignmentWeights,               Core/TextDecoder.swift:141)
 */
#ifndef WHISPERHIP_H
#define WHISPERHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wh_model wh_model;
typedef struct wh_session wh_session;
typedef struct wh_transcription wh_transcription;
typedef struct wh_tokenizer wh_tokenizer;

/* Utilities/WhisperError.swift:7-18 */
typedef enum wh_status {
    WH_OK = 0,
    WH_ERR_TOKENIZER_UNAVAILABLE = 1,
    WH_ERR_MODELS_UNAVAILABLE = 2,
    WH_ERR_PREFILL_FAILED = 3,
    WH_ERR_AUDIO_PROCESSING_FAILED = 4,
    WH_ERR_DECODING_LOGITS_FAILED = 5,
    WH_ERR_SEGMENTING_FAILED = 6,
    WH_ERR_LOAD_AUDIO_FAILED = 7,
    WH_ERR_PREPARE_DECODER_INPUTS_FAILED = 8,
    WH_ERR_TRANSCRIPTION_FAILED = 9,
    WH_ERR_DECODING_FAILED = 10,
    WH_ERR_INVALID_ARGUMENT = 100,
    WH_ERR_HIP = 101,
    WH_ERR_CANCELLED = 102,     /* Swift CancellationError thrown by Task.checkCancellation (Core/TranscribeTask.swift:135,144,165) */
    WH_ERR_OUT_OF_MEMORY = 103  /* a host allocation failed inside the library (std::bad_alloc never crosses the C ABI) */
} wh_status;

#define WH_WINDOW_SAMPLES 480000 /* Constants.defaultWindowSamples, Core/Models.swift:1457 */
#define WH_MEL_FRAMES 3000
#define WH_AUDIO_CTX 1500
#define WH_MAX_TOKEN_CONTEXT 224 /* Constants.maxTokenContext, Core/Models.swift:1334 */
#define WH_MAX_RESULT_TOKENS 232
#define WH_SAMPLE_RATE 16000     /* WhisperKit.sampleRate, Core/WhisperKit.swift:38 */

/* openai/whisper ModelDimensions; written into the weight blob header */
typedef struct wh_dims {
    int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wh_dims;

/* SpecialTokens, Core/Models.swift:1111-1149 (+ the contiguous language-token range that
 * WhisperTokenizer.allLanguageTokens enumerates, Core/Models.swift:1160) */
typedef struct wh_special_tokens {
    int32_t end_token, english_token, no_speech_token, no_timestamps_token, special_token_begin;
    int32_t start_of_previous_token, start_of_transcript_token, time_token_begin;
    int32_t transcribe_token, translate_token, whitespace_token;
    int32_t language_token_begin, n_language_tokens;
} wh_special_tokens;

/* DecodingOptions, Core/Configurations.swift:155-247.  Optionals: NAN / -1 / NULL mean `nil`. */
typedef struct wh_decoding_options {
    int32_t task;                 /* 0 transcribe, 1 translate */
    int32_t language_token;       /* id of "<|lang|>" or -1 (nil) */
    float temperature;
    float temperature_increment_on_fallback;
    int32_t temperature_fallback_count;
    int32_t sample_length;
    int32_t top_k;
    int32_t use_prefill_prompt;
    int32_t detect_language;      /* -1: default (= !use_prefill_prompt) */
    int32_t skip_special_tokens;
    int32_t without_timestamps;
    int32_t word_timestamps;
    float max_initial_timestamp;  /* NAN = nil; carried for API parity only: the rule that would use it is commented out in the
                                     reference (Core/Text/LogitsFilter.swift:112-122), so it has no effect there or here */
    int32_t max_window_seek;      /* -1 = nil */
    const float* clip_timestamps;
    int32_t n_clip_timestamps;
    float window_clip_time;
    const int32_t* prompt_tokens; /* NULL = nil */
    int32_t n_prompt_tokens;
    const int32_t* prefix_tokens; /* NULL = nil */
    int32_t n_prefix_tokens;
    int32_t suppress_blank;
    const int32_t* suppress_tokens;
    int32_t n_suppress_tokens;
    float compression_ratio_threshold; /* NAN = nil */
    float log_prob_threshold;
    float first_token_log_prob_threshold;
    float no_speech_threshold;
    uint64_t seed;                /* seeds the T>0 multinomial draw (reference: unseeded system RNG) */
    int32_t float16_logits;       /* reference-numerics switch, default 0 (fp32).  1: logits are rounded to Float16 before the filters
                                     and the sampler - the TextDecoder output is a Float16 MLMultiArray on arm64 (Core/Models.swift:1041,
                                     FloatType, ArgmaxCore/FloatType.swift:9-13) - and TimestampRulesFilter compares its two log-probabilities
                                     in Float16 (Core/Text/LogitsFilter.swift:144-242: BNNS logSoftmax / logSumExp / max on FloatType) */
    int32_t beam_size;            /* <= 1: greedy (the reference).  > 1: the temperature-0 pass of the fallback ladder is a beam search
                                     (wh_decode_text_beam: openai/whisper semantics, NO REFERENCE BEHAVIOUR - the reference's
                                     BeamSearchTokenSampler is fatalError); fallback temperatures > 0 sample as usual.  A transcribe call
                                     then batches max_batch / beam_size windows per round; not combinable with word_timestamps */
    float beam_patience;          /* maxCandidates = Int(Float(beamSize) * patience), TokenSampler.swift:269; default 1 */
    int32_t reserved_;
} wh_decoding_options;

/* DecodingFallback.fallbackReason, Core/Models.swift:357-381 */
enum { WH_FALLBACK_NONE = 0, WH_FALLBACK_FIRST_TOKEN_LOGPROB = 1, WH_FALLBACK_SILENCE = 2,
       WH_FALLBACK_COMPRESSION_RATIO = 3, WH_FALLBACK_LOGPROB = 4 };

/* DecodingResult, Core/Models.swift:383-439 (tokens are SOT..EOT inclusive) */
typedef struct wh_decoding_result {
    int32_t n_tokens;
    int32_t tokens[WH_MAX_RESULT_TOKENS];
    float token_logprobs[WH_MAX_RESULT_TOKENS];
    float avg_logprob;
    float no_speech_prob;
    float temperature;
    float compression_ratio;
    int32_t language_token;       /* first language token among the result tokens, or -1 */
    int32_t fallback_reason;      /* WH_FALLBACK_* */
    int32_t needs_fallback;
    int32_t is_first_token_logprob_too_low;
    int32_t steps;                /* decoder forward passes executed (timings.totalDecodingLoops) */
} wh_decoding_result;

/* TranscriptionSegment / WordTiming, Core/Models.swift:560-641 */
typedef struct wh_segment {
    int32_t id, seek;
    float start, end;
    int32_t token_offset, n_tokens; /* into the transcription's flat token / logprob arrays */
    float temperature, avg_logprob, compression_ratio, no_speech_prob;
    int32_t word_offset, n_words;
} wh_segment;

typedef struct wh_word_timing {
    int32_t token_offset, n_tokens; /* into the transcription's flat WORD-token array (wh_transcription_word_tokens): a merged
                                       word keeps only its text tokens, which need not be adjacent in the segment */
    float start, end, probability;
} wh_word_timing;

/* TranscriptionTimings, Core/Models.swift:730-844 (seconds; device stages measured with HIP events) */
typedef struct wh_timings {
    double audio_processing, logmels, encoding, decoding_init, decoding_predictions, decoding_filtering,
        decoding_sampling, decoding_kv_caching, decoding_word_timestamps, decoding_fallback, decoding_windowing,
        decoding_loop, full_pipeline, input_audio_seconds;
    double total_decoding_loops, total_decoding_windows, total_decoding_fallbacks, total_encoding_runs,
        total_logmel_runs;
    /* the remaining stored properties of TranscriptionTimings, so that the JSON report carries every key of the reference's:
     * pipeline_start / first_token_time are CFAbsoluteTime (seconds since 2001-01-01 UTC) */
    double pipeline_start, first_token_time, model_loading, prewarm_load_time, encoder_load_time, decoder_load_time,
        encoder_specialization_time, decoder_specialization_time, tokenizer_load_time, audio_loading, decoding_non_prediction,
        total_audio_processing_runs, total_kv_update_runs, total_timestamp_alignment_runs;
} wh_timings;

const char* wh_last_error(void);
const char* wh_version(void);

/* ---- model: WhisperKit.loadModels (Core/WhisperKit.swift:358-442) ------------------------------ */
int wh_model_create(const void* blob, size_t nbytes, int device, wh_model** out); /* WHIPW001 blob (weights.py) */
int wh_model_load(const char* path, int device, wh_model** out);
void wh_model_destroy(wh_model* m);
int wh_model_dims(const wh_model* m, wh_dims* out);
/* alignment heads for word timestamps: (layer, head) pairs; default = the blob's "dec.alignment_heads" tensor when present
 * (checkpoint conversion stores generation_config.alignment_heads there), else the upper half of the decoder layers.  May be
 * called while sessions exist but not while one of them is decoding: sessions re-size their score buffers on their next use. */
int wh_model_set_alignment_heads(wh_model* m, const int32_t* layer_head_pairs, int n_pairs);

/* dimension getters the reference introspects from the CoreML models */
int wh_mel_count(const wh_model* m);                    /* FeatureExtracting.melCount        Core/FeatureExtractor.swift:24 */
int wh_window_samples(const wh_model* m);               /* FeatureExtracting.windowSamples   Core/FeatureExtractor.swift:32 */
int wh_embed_size(const wh_model* m);                   /* AudioEncoding.embedSize           Core/AudioEncoder.swift:24 */
int wh_logits_size(const wh_model* m);                  /* TextDecoding.logitsSize           Core/TextDecoder.swift:313 */
int wh_kv_cache_embed_dim(const wh_model* m);           /* TextDecoding.kvCacheEmbedDim      Core/TextDecoder.swift:317 */
int wh_kv_cache_max_sequence_length(const wh_model* m); /* TextDecoding.kvCacheMaxSequenceLength :321 */
int wh_window_size(const wh_model* m);                  /* TextDecoding.windowSize           Core/TextDecoder.swift:325 */
int wh_is_model_multilingual(const wh_model* m);        /* TextDecoding.isModelMultilingual  Utilities/ModelUtilities.swift:124 */
int wh_supports_word_timestamps(const wh_model* m);     /* TextDecoding.supportsWordTimestamps Core/TextDecoder.swift:309 */
int wh_special_tokens_default(const wh_model* m, wh_special_tokens* out); /* ids implied by the vocabulary size */
void wh_decoding_options_default(wh_decoding_options* out);  /* DecodingOptions() defaults */

/* ---- session: prepareDecoderInputs (Core/TextDecoder.swift:109-161) ------------------------------ */
int wh_session_create(wh_model* m, int max_batch /* 1 .. 256 windows decoded in lock-step */, wh_session** out);
void wh_session_destroy(wh_session* s);
int wh_session_max_batch(const wh_session* s);
/* 1: the session's decoder attends over the encoder output directly (weight-absorbed cross-attention: encoder_output_embeds is the
   decoder input in the reference too, Core/Models.swift:986-987), 0: per-layer cross K / V rows (24 bits per element: Float16 + an 8-bit residual) are materialised by
   wh_prepare_decoder_inputs.  Fixed at creation: automatic = absorbed when the width supports it (512 / 768 / 1024 / 1280) and
   max_batch >= wh_xabs_auto_min_slots() (28; WH_XABS_MIN_SLOTS), or WH_XABS=0 / 1.  The choice is made from max_batch alone, so
   Session(m, 27) and Session(m, 28) run different kernels: both modes meet the 1e-3 relative logits contract against the fp32
   model, bit-identity across batch sizes holds within a mode.  Slots that share an encoder output (beam search: beam_size slots per
   audio) need no special request: mode 1 reads the shared tensor with cacheable loads then (the beams of an audio are dispatched back to
   back onto one XCD) and measures the same as mode 0 with its 24-bit rows shared through the L2 - 250.2 vs 249.9 audio-s/s on
   BASELINE configs[4] with beam = 5 (profiles/r06a_beam5_cross_attention_mode_ab_24bit_rows.jsonl) - so beam sessions take the
   automatic choice like every other session.  Mode 1 reads the session's encoder output LIVE at every decoder step (mode 0 snapshots it into the K / V
   rows in wh_prepare_decoder_inputs): wh_encode_features / wh_set_encoder_output between wh_prepare_decoder_inputs and the last
   decoder step of a window change the results in mode 1 - the reference passes encoder_output_embeds to every call as well. */
int wh_session_cross_attention_mode(const wh_session* s);
int wh_xabs_auto_min_slots(void);
/* the automatic key-split count of an absorbed session of max_batch slots (see wh_session_create_tuned) */
int wh_xabs_auto_splits(int max_batch);
/* key splits per slot of the absorbed cross-attention (0 in K / V-row mode): slots x splits workgroups, one per CU, stream the
   encoder output; fixed at creation (bench.py prices the kernel's algorithmic bytes with it) */
int wh_session_cross_attention_splits(const wh_session* s);
/* wh_session_create with the cross-attention mode chosen by the caller: -1 automatic (= wh_session_create), 0 K / V rows, 1 absorbed
   (WH_ERR_INVALID_ARGUMENT when the model width does not support it) */
int wh_session_create_with_mode(wh_model* m, int max_batch, int cross_attention_mode, wh_session** out);
/* ... and the key splits per slot of the absorbed form: 0 automatic, 1 .. 4.  One workgroup per (slot, split) owns a CU while it
   streams, so slots x splits is the share of the chip one session's cross-attention takes.  Automatic = wh_xabs_auto_splits(max_batch):
   as many splits as keep slots x splits within one round of the 256 CUs - 4 up to 64 slots, 3 up to 85, 2 up to 128, 1 beyond - which is
   fastest for a session running alone (profiles/r06ah_lone_session_key_splits.jsonl: 128 slots 7.64 -> 6.86 ms per decoder step against
   the 4 splits every session got before, 256 slots 12.82 -> 10.92).  A caller that keeps several sessions in flight asks for half of that
   (the other sessions' kernels keep the other half of the chip; profiles/r04ad_*); a beam-search caller (slots that share an encoder
   output stream it as L2 hits: more workgroups win) asks for 4.  The split count fixes the order of the key-split combine: results are
   bit-identical across batch sizes and sessions for EQUAL split counts - two sessions whose max_batch falls into different automatic
   classes agree to ~1e-6 relative, not bit for bit, unless the caller pins the count.  Ignored in K / V-row mode. */
int wh_session_create_tuned(wh_model* m, int max_batch, int cross_attention_mode, int cross_attention_splits, wh_session** out);
/* ... and, since round 6, every creation knob in one struct (zero-initialise or wh_session_options_default; NULL = defaults):
     cross_attention_mode                 -1 automatic, 0 K / V rows, 1 absorbed          (as wh_session_create_with_mode)
     cross_attention_splits               0 automatic, 1 .. 4 key splits per slot          (as wh_session_create_tuned)
     cross_attention_slots_per_workgroup  0 automatic (1), 1 .. 16: a workgroup of the absorbed cross-attention streams this many slots one
                                          after the other, so a launch takes ceil(batch / n) x splits workgroups = CUs whatever the batch.
                                          A process that keeps several sessions in flight gives every session's cross-attention about half of
                                          the chip (128 workgroups): with 256-slot device batches that is 1 split and 2 slots per workgroup
                                          (bench.py: 2660 -> 2749 audio-s/s against 128-slot batches, profiles/r06i_*).  A slot is processed
                                          exactly as by a workgroup of its own: results do not depend on this number, bit for bit. */
typedef struct wh_session_options {
    int32_t cross_attention_mode;
    int32_t cross_attention_splits;
    int32_t cross_attention_slots_per_workgroup;
    int32_t reserved_[5];
} wh_session_options;
void wh_session_options_default(wh_session_options* out);
int wh_session_create_with_options(wh_model* m, int max_batch, const wh_session_options* opt, wh_session** out);
int wh_session_cross_attention_slots_per_workgroup(const wh_session* s);      /* 0 in K / V-row mode */
/* development aid (kernel bring-up, tools/xabs_check.py): the first nbytes of a named decode-step device buffer ("q", "zb_hi", ...);
   nbytes beyond the buffer's size is WH_ERR_INVALID_ARGUMENT */
int wh_debug_peek(wh_session* s, const char* name, void* out_host, size_t nbytes);
/* captured step graphs the session holds (one per configuration = (batch, alignment, sampler fusion) and 8 decoder positions; the cache is
   capped at WH_GRAPH_CAP, default 112, graphs: the configuration used longest ago is dropped first) */
int wh_session_step_graph_count(const wh_session* s);
int wh_session_synchronize(wh_session* s);
void* wh_session_stream(wh_session* s);                 /* hipStream_t of the session */

/* AudioProcessing.padOrTrimAudio (Core/Audio/AudioProcessor.swift:151-174): copy <=480000 samples into slot b, zero pad */
int wh_set_audio(wh_session* s, int b, const float* pcm_host, int n_samples);
int wh_set_audio_device(wh_session* s, int b, const float* pcm_device, int n_samples);

/* FeatureExtracting.logMelSpectrogram (Core/FeatureExtractor.swift:40-56) for slots [0, batch) */
int wh_log_mel_spectrogram(wh_session* s, int batch);
int wh_get_mel(wh_session* s, int b, float* out_host /* [n_mels][3000] */);
int wh_set_mel(wh_session* s, int b, const float* mel_host /* [n_mels][3000] */);

/* AudioEncoding.encodeFeatures (Core/AudioEncoder.swift:50-63) for slots [0, batch) */
int wh_encode_features(wh_session* s, int batch);
int wh_get_encoder_output(wh_session* s, int b, float* out_host /* [1500][d] */);
int wh_set_encoder_output(wh_session* s, int b, const float* enc_host /* [1500][d] */);

/* TextDecoding.prepareDecoderInputs + DecodingInputs.reset: project the encoder output to the per-layer
 * cross-attention K/V rows (once per window; K / V-row mode only - the absorbed mode keeps reading the encoder output itself, see
 * wh_session_cross_attention_mode) and clear the decode state of slots [0, batch) */
int wh_prepare_decoder_inputs(wh_session* s, int batch);
int wh_reset_decoder_inputs(wh_session* s, int batch);  /* DecodingInputs.reset, Core/Models.swift:312-322 */

/* TextDecoding.predictLogits + updateKVCache + updateAlignmentWeights (Core/TextDecoder.swift:381-418,218-296):
 * one decoder call per slot; writes this step's K/V at `positions[b]` and the alignment row at positions[b]+1. */
int wh_predict_logits(wh_session* s, int batch, const int32_t* tokens, const int32_t* positions,
                      float* logits_out_host /* [batch][n_vocab] or NULL */);
int wh_get_alignment_weights(wh_session* s, int b, float* out_host /* [224][1500] */);

/* Optional openai/whisper-style post-processing of the alignment heads inside wh_get_alignment_weights (and therefore in the
 * word timestamps of wh_transcribe*): softmax rows -> z-normalise each (head, frame) over the decoded token rows -> median filter
 * of odd width along the frames -> mean over heads (openai/whisper timing.py find_alignment; HF generation_whisper.py:341-349,
 * median_filter_width 7).  The reference applies none of it on the host (Core/Text/SegmentSeeker.swift:195-237) and whether the
 * CoreML bundles bake it in is not published: default off (z_normalize 0, median_filter_width 0). */
int wh_session_set_alignment_postprocess(wh_session* s, int z_normalize, int median_filter_width);

/* A device-resident stage output as a plain tensor descriptor (what the MLMultiArray results of the CoreML stages,
 * Core/Models.swift:848-1107, become behind a C ABI): row-major, `device` = HIP device ordinal (the session's model device). The
 * memory stays owned by the session and is valid until the stage runs again on that session. */
enum { WH_DTYPE_F32 = 0, WH_DTYPE_F16 = 1 };
typedef struct wh_tensor {
    void* data;
    int32_t dtype;        /* WH_DTYPE_* */
    int32_t ndim;
    int64_t shape[4];
    int32_t device;
    int32_t reserved_;
} wh_tensor;
int wh_get_mel_tensor(wh_session* s, int b, wh_tensor* out);                          /* f32 [n_mels][3000], the reference layout */
int wh_get_encoder_output_tensor(wh_session* s, int b, int dtype, wh_tensor* out);    /* f32 or f16 [1500][d] */
int wh_get_logits_tensor(wh_session* s, wh_tensor* out);                              /* f32 [max_batch][n_vocab], step API / T > 0 */
/* Device-resident hand-off (MLMultiArray outputs of the CoreML stages stay on the accelerator in the reference too): pointers
 * into the session's HBM buffers of slot b, valid until the session rewrites them; consume them on wh_session_stream(s) or after
 * wh_session_synchronize.  mel [n_mels][3000] f32; encoder output [1500][d] as f32 and as the f16 copy the decoder reads
 * (either pointer argument may be NULL); logits [max_batch][n_vocab] f32 as left by the last wh_predict_logits / T > 0 step. */
int wh_get_mel_device(wh_session* s, int b, const float** mel_dev);
int wh_get_encoder_output_device(wh_session* s, int b, const float** enc_f32_dev, const void** enc_f16_dev);
int wh_get_logits_device(wh_session* s, const float** logits_dev);
/* Task.checkCancellation (Core/TranscribeTask.swift:135,144,165; TextDecoder early stop): `flag` is polled between pipeline
 * stages, between windows and every 8 decoder steps; a non-zero value makes the running call return WH_ERR_CANCELLED.  NULL
 * removes it.  The flag must outlive the calls it guards. */
int wh_session_set_cancel_flag(wh_session* s, const volatile int32_t* flag);

/* LogitsFiltering.filterLogits for the built-in chain of createLogitsFilters (Core/TextDecoder.swift:857-899,
 * Core/Text/LogitsFilter.swift) applied on device to caller-supplied logits; `tokens` = currentTokens,
 * `prefilled_index`/`initial_prompt_index` as in decodeText.  language_filter != 0 applies LanguageLogitsFilter instead. */
int wh_filter_logits(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st,
                     const int32_t* tokens, int n_tokens, int prefilled_index, int initial_prompt_index,
                     int language_filter, float* logits_inout_host, int n_logits);
/* TokenSampling.update for GreedyTokenSampler (Core/Text/TokenSampler.swift:29-252) on device */
int wh_sample_token(wh_session* s, const float* logits_host, int n_logits, float temperature, int top_k,
                    uint64_t seed, int counter, int32_t* token_out, float* logprob_out);

/* TranscriptionCallback (Core/Models.swift:728; invoked per token by decodeText, Core/TextDecoder.swift:723-741, early stop via
 * earlyStopActor :752-755).  The token loop runs on the device, so the callback fires on the calling thread each time the host
 * looks at the device state - every 8 decoder steps - once per unfinished slot, with the slot's currentTokens.  Return 0 to
 * stop that slot early (its result is finalised with the tokens decoded so far), non-zero to continue.  `text` is the current
 * transcript when a tokenizer is attached, else NULL.  NULL fn removes the callback. */
typedef struct wh_progress {
    int32_t slot, n_tokens;
    const int32_t* tokens;
    float avg_logprob, compression_ratio;
    const char* text;
} wh_progress;
typedef int (*wh_progress_fn)(void* user, const wh_progress* progress);
int wh_session_set_progress_callback(wh_session* s, wh_progress_fn fn, void* user);

/* The window-level extension points of TranscribeTask (Core/TranscribeTask.swift), for hosts that use the library's own orchestrator
 * (wh_transcribe, wh_transcribe_batch, wh_transcribe_chunked) instead of a Swift TranscribeTask of their own.  All are optional (NULL);
 * they run on the calling thread; `audio_index` is the index of the audio in the call's batch (0 for wh_transcribe).
 *   window_preprocess   TranscribeTask.windowPreprocess (:42-46, called at :130): after padOrTrim of a window, before the decoder
 *                       pipeline runs on it; `samples` are the window's segment_size samples (the zero padding to 480000 is implied)
 *   window_postprocess  TranscribeTask.windowPostProcess (:49-55, called at :246): the window's segments are segments
 *                       [first_segment, first_segment + n_segments) of `t`; the hook may change their times
 *                       (wh_transcription_set_segment_times) and returns how many of them to KEEP (0 .. n_segments, a negative value
 *                       keeps all): the others are removed before segment discovery and never reach the result
 *   segment_discovery   SegmentDiscoveryCallback (Core/Models.swift:668, called at :260) with the window's final segments
 * Windows without segments call neither of the last two (`guard let currentSegments`, :239-242). */
typedef struct wh_window_hooks {
    void (*window_preprocess)(void* user, int audio_index, const float* samples, int seek, int segment_size);
    int (*window_postprocess)(void* user, int audio_index, int seek, int segment_size, wh_transcription* t, int first_segment, int n_segments);
    void (*segment_discovery)(void* user, int audio_index, const wh_transcription* t, int first_segment, int n_segments);
    void* user;
} wh_window_hooks;
int wh_session_set_window_hooks(wh_session* s, const wh_window_hooks* hooks);   /* copied; NULL removes them */
int wh_transcription_set_segment_times(wh_transcription* t, int i, float start, float end);

/* TextDecoding.decodeText (Core/TextDecoder.swift:541-855) for slots [0, batch), whole token loop on device.
 * temperatures[b] is the sampler temperature of slot b; active[b]==0 skips the slot (may be NULL = all active). */
int wh_decode_text(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st,
                   const int32_t* prompt, int n_prompt, const float* temperatures, const int32_t* active,
                   uint64_t seed, wh_decoding_result* out /* [batch] */);
/* The same with one language token per slot (language_tokens[b] >= 0 replaces the token after <|startoftranscript|> of the
 * shared prompt for slot b): batched windows of audios in different languages, each prompted like its own TranscribeTask
 * would prompt it (Core/TextDecoder.swift:183-188).  language_tokens == NULL: identical to wh_decode_text. */
int wh_decode_text_languages(wh_session* s, int batch, const wh_decoding_options* opt, const wh_special_tokens* st,
                             const int32_t* prompt, int n_prompt, const int32_t* language_tokens, const float* temperatures,
                             const int32_t* active, uint64_t seed, wh_decoding_result* out /* [batch] */);
/* ---- user-pluggable LogitsFiltering / TokenSampling (Core/Text/LogitsFilter.swift:8-10, Core/Text/TokenSampler.swift:8-11) ----
 * The reference runs `logitsFilters` (custom filters first, Core/TextDecoder.swift:857-899) and the `TokenSampling` object on the
 * host once per token (:641-652).  The fused device loop of wh_decode_text knows the four built-in filters and the greedy /
 * top-k sampler only; a caller with its own filter or sampler decodes through this entry point instead: the reference's
 * decodeText loop (:573-757) on the host over the step API, one slot (slot 0) - per token one wh_predict_logits, the caller's
 * filters in order on the host logits (in place), the built-in chain of `opt` on the device (wh_filter_logits), then the
 * caller's sampler or, with sampler == NULL, GreedyTokenSampler(temperature, top_k, seed) exactly as the device loop samples.
 *   filter:  LogitsFiltering.filterLogits(_:withTokens:) - modify logits[0 .. n_logits) in place; tokens = currentTokens
 *   sampler: TokenSampling.update(tokens:logits:logProbs:) - write the sampled id and its log-probability; return non-zero
 *            when decoding is complete (SamplingResult.completed); the loop appends EOT like TokenSampling.finalize if the
 *            caller's last token is not EOT (GreedyTokenSampler.finalize, TokenSampler.swift:242-251).
 * Progress callback, cancel flag and alignment rows behave as in wh_decode_text (the callback fires after every token). */
typedef void (*wh_logits_filter_fn)(void* user, float* logits, int32_t n_logits, const int32_t* tokens, int32_t n_tokens);
typedef int32_t (*wh_token_sampler_fn)(void* user, const float* logits, int32_t n_logits, const int32_t* tokens, const float* logprobs,
                                       int32_t n_tokens, int32_t* token_out, float* logprob_out);
int wh_decode_text_custom(wh_session* s, const wh_decoding_options* opt, const wh_special_tokens* st, const int32_t* prompt,
                          int n_prompt, float temperature, uint64_t seed, const wh_logits_filter_fn* filters,
                          void* const* filter_users, int n_filters, wh_token_sampler_fn sampler, void* sampler_user,
                          wh_decoding_result* out /* [1] */);
/* ---- beam search: NO REFERENCE BEHAVIOUR ------------------------------------------------------------------------------
 * BeamSearchTokenSampler (Core/Text/TokenSampler.swift:254-290) exists in the reference as a class whose update / finalize are
 * fatalError("Not implemented").  These entry points keep its construction parameters (beamSize, eotToken, patience,
 * maxCandidates = Int(Float(beamSize) * patience)) and implement openai/whisper's BeamSearchDecoder (whisper/decoding.py
 * v20231117 :343-424 + MaximumLikelihoodRanker with length_penalty None) - what BASELINE configs[4] asks for, labelled as such.
 * The sampler object is plain host code (one audio): update takes the live beams (n_beams sequences of `len` tokens, their
 * per-token log-probs or NULL, their log-prob sums) and per beam the beam_size + 1 best (log-prob, token) pairs of the
 * log-softmax of its filtered logits, best first (topk_stride floats / ints per beam); it returns the next beams (<= beam_size
 * sequences of len + 1 tokens), the beam each one continues (`sources`, the cache rearrangement) and whether max_candidates
 * sequences have finished.  finalize adds the live beams (EOT appended) when fewer than beam_size finished and returns the
 * best finished sequence by sum / sampled length. */
typedef struct wh_beam_sampler wh_beam_sampler;
int wh_beam_sampler_create(int beam_size, int32_t eot_token, float patience, wh_beam_sampler** out);
void wh_beam_sampler_destroy(wh_beam_sampler* h);
void wh_beam_sampler_reset(wh_beam_sampler* h);
int wh_beam_sampler_max_candidates(const wh_beam_sampler* h);
int wh_beam_sampler_finished_count(const wh_beam_sampler* h);
int wh_beam_sampler_update(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs,
                           const float* sums, const float* topk_logprobs, const int32_t* topk_tokens, int topk_stride,
                           int32_t* new_tokens /* [beam_size][len + 1] */, float* new_token_logprobs /* same shape or NULL */,
                           float* new_sums, int32_t* sources, int32_t* n_new, int32_t* completed);
int wh_beam_sampler_finalize(wh_beam_sampler* h, int n_beams, int len, const int32_t* tokens, const float* token_logprobs,
                             const float* sums, int sample_begin, int capacity, int32_t* best_tokens, float* best_token_logprobs,
                             int32_t* best_len, float* best_sum, int32_t* n_finished);
/* decodeText at temperature 0 with that sampler, for n_audio windows whose decoder inputs were prepared in slots
 * [0, n_audio) (wh_prepare_decoder_inputs): the prompt is pre-filled exactly like wh_decode_text; audio a then owns the slots
 * a * beam_size ... (n_audio * beam_size <= max_batch; nothing is copied: its beams read the audio's one cross K/V and each
 * other's self-attention rows in place, but the slots' self-attention caches are overwritten - prepare the decoder inputs again
 * before another decode), and every position expands the beams: decoder step, the LogitsFilters of opt per beam + log-softmax +
 * top (beam_size + 1) on the device, candidate ranking on the host (the cache "rearrangement" is a row -> owner table).
 * language_tokens: per audio or NULL, as in wh_decode_text_languages.  Results follow the
 * DecodingResult conventions of wh_decode_text (temperature 0); word timestamps are not recorded along beams. */
int wh_decode_text_beam(wh_session* s, int n_audio, int beam_size, float patience, const wh_decoding_options* opt,
                        const wh_special_tokens* st, const int32_t* prompt, int n_prompt, const int32_t* language_tokens,
                        wh_decoding_result* out /* [n_audio] */);

/* TextDecoding.detectLanguage (Core/TextDecoder.swift:420-539) */
int wh_detect_language(wh_session* s, int batch, const wh_special_tokens* st, int32_t* language_tokens_out,
                       float* logprobs_out);
/* prefillDecoderInputs (Core/TextDecoder.swift:163-216): builds the forced prompt; returns its length */
int wh_prefill_prompt(const wh_model* m, const wh_decoding_options* opt, const wh_special_tokens* st,
                      int32_t language_token, int32_t* prompt_out, int capacity);

/* ---- tokenizer text: WhisperTokenizer (Core/Models.swift:1150-1307) over the vendored swift-transformers decoder
 * (ArgmaxCore/External/Tokenizers/Tokenizer.swift:510-525, Decoder.swift:126-165).  Host only - no GPU involved.
 * Strings are UTF-8; functions that write a string return its full byte length (without the NUL) and truncate to
 * capacity-1 bytes + NUL, so a first call with (NULL, 0) sizes the buffer. */
/* ModelUtilities.loadTokenizer (Utilities/ModelUtilities.swift:175-203): ByteLevel-BPE `tokenizer.json`; a
 * `tokenizer_config.json` next to it supplies clean_up_tokenization_spaces (default true). */
int wh_tokenizer_load(const char* tokenizer_json_path, wh_tokenizer** out);
void wh_tokenizer_destroy(wh_tokenizer* t);
int wh_tokenizer_vocab_size(const wh_tokenizer* t);
int wh_tokenizer_decode(const wh_tokenizer* t, const int32_t* tokens, int n, int skip_special_tokens, char* out, int capacity);
int wh_tokenizer_token_to_id(const wh_tokenizer* t, const char* token);                    /* convertTokenToId; -1 = nil */
int wh_tokenizer_id_to_token(const wh_tokenizer* t, int id, char* out, int capacity);      /* convertIdToToken; -1 = nil */
/* WhisperTokenizerWrapper.init (Core/Models.swift:1198-1224): ids looked up by token text, reference defaults otherwise */
int wh_tokenizer_special_tokens(const wh_tokenizer* t, wh_special_tokens* out);
/* splitToWordTokens (Core/Models.swift:1226-1306).  The reference picks the unicode / space splitter with Apple's
 * NLLanguageRecognizer; here the caller passes the language code (zh ja th lo my yue -> unicode splitter).  Returns the
 * number of words; word_token_counts[i] = tokens of word i (consecutive in `tokens`), word_byte_counts[i] = UTF-8 bytes of
 * word i; words_out receives the words back to back without terminators (a byte token may decode to NUL);
 * *words_bytes = bytes needed.  -2 when a buffer is too small. */
int wh_tokenizer_split_to_word_tokens(const wh_tokenizer* t, const int32_t* tokens, int n, const char* language_code,
                                      int32_t* word_token_counts, int32_t* word_byte_counts, int counts_capacity,
                                      char* words_out, int words_capacity, int* words_bytes);
/* TextDecoding.tokenizer (Core/TextDecoder.swift:61): with a tokenizer attached the transcribe entry points produce
 * segment / word / result text, group word timestamps by real words and infer the language code.  NULL detaches.
 * The tokenizer must outlive the session's use of it. */
int wh_session_set_tokenizer(wh_session* s, const wh_tokenizer* t);

/* ---- TranscribeTask.run (Core/TranscribeTask.swift:57-296) --------------------------------------- */
int wh_transcribe(wh_session* s, const float* pcm_host, int n_samples, const wh_decoding_options* opt,
                  const wh_special_tokens* st, wh_transcription** out);
/* WhisperKit.transcribe(audioArrays:) -> [[TranscriptionResult]?] (Core/WhisperKit.swift:660-688 over :716-812): independent audios /
 * chunks with ONE options value, batched on the device.  Every audio carries its own Result as in the reference (`.failure(error)`,
 * :786-790): an audio that fails leaves out[i] == NULL (the reference's nil) and does not fail its neighbours; its status and message stay
 * readable through wh_session_item_status / wh_session_item_error until the session's next batch call.  The call itself returns non-zero
 * only when nothing could run: a null argument, every audio failed (the first failing audio's status), a device error, cancellation. */
int wh_transcribe_batch(wh_session* s, const float* const* pcm_host, const int32_t* n_samples, int n_audio,
                        const wh_decoding_options* opt, const wh_special_tokens* st, wh_transcription** out /* [n_audio] */);
/* WhisperKit.transcribeWithOptions(audioArrays:decodeOptionsArray:) -> [Result<[TranscriptionResult], Error>] (Core/WhisperKit.swift:716-812):
 * opts[i] are the options of audio i (a NULL entry, or opts == NULL, = DecodingOptions()); statuses[i] (may be NULL) receives audio i's
 * wh_status, out[i] its transcription or NULL.  Audios whose options differ only in their clip timestamps share lock-stepped device
 * batches; audios with different decoding options run in groups, one group after the other (the reference runs one TranscribeTask per
 * audio: grouping changes the batching, never a result).  Returns WH_OK whenever the per-audio results are valid - also when every audio
 * failed - and non-zero for a null argument, a device error or cancellation. */
int wh_transcribe_batch_with_options(wh_session* s, const float* const* pcm_host, const int32_t* n_samples, int n_audio,
                                     const wh_decoding_options* const* opts /* [n_audio] or NULL */, const wh_special_tokens* st,
                                     wh_transcription** out /* [n_audio] */, int32_t* statuses /* [n_audio] or NULL */);
/* The Result of audio `audio_index` of the session's last wh_transcribe_batch / _with_options / wh_transcribe / wh_transcribe_chunked call:
 * its wh_status, and the message of its error ("" when it succeeded; valid until the next such call on the session). */
int wh_session_item_status(const wh_session* s, int audio_index);
const char* wh_session_item_error(const wh_session* s, int audio_index);
/* WhisperKit.transcribe(audioArray:) with chunkingStrategy .vad (Core/WhisperKit.swift:867-931): audio longer than one window is
 * split by VADAudioChunker.chunkAll, the chunks are transcribed as independent audios (batched on the device, clipTimestamps
 * reset) and every segment / word is shifted by its chunk's seek offset (AudioChunking.updateSeekOffsetsForResults,
 * Core/Audio/AudioChunker.swift:14-39; TranscriptionUtilities.updateSegmentTimings, Utilities/TranscriptionUtilities.swift:55-69).
 * Writes one transcription per chunk into out[0..*n_out) (chunk order) and the chunks' seek offsets (samples) into
 * seek_offsets_out (may be NULL); returns WH_ERR_INVALID_ARGUMENT when more than `capacity` chunks are needed.  A chunk that fails is
 * skipped as in updateSeekOffsetsForResults (`case .failure`: logged, not returned): *n_out counts the chunks that succeeded. */
int wh_transcribe_chunked(wh_session* s, const float* pcm_host, int n_samples, const wh_decoding_options* opt,
                          const wh_special_tokens* st, wh_transcription** out, int capacity, int32_t* seek_offsets_out, int* n_out);
void wh_transcription_free(wh_transcription* t);
int wh_transcription_n_segments(const wh_transcription* t);
int wh_transcription_segment(const wh_transcription* t, int i, wh_segment* out);
int wh_transcription_n_words(const wh_transcription* t);
int wh_transcription_word(const wh_transcription* t, int i, wh_word_timing* out);
int wh_transcription_tokens(const wh_transcription* t, const int32_t** tokens, const float** logprobs, int* n);
int wh_transcription_language_token(const wh_transcription* t);
int wh_transcription_timings(const wh_transcription* t, wh_timings* out);
int wh_transcription_window_seeks(const wh_transcription* t, const int32_t** seeks, int* n);

/* ---- text of a transcription (present when a tokenizer was attached to the session, or for objects built by
 * wh_transcription_create / wh_merge_transcriptions); string-returning functions follow the wh_tokenizer_decode convention */
int wh_transcription_has_text(const wh_transcription* t);
int wh_transcription_text(const wh_transcription* t, char* out, int capacity);            /* TranscriptionResult.text */
int wh_transcription_language(const wh_transcription* t, char* out, int capacity);        /* TranscriptionResult.language ("en", ...) */
int wh_transcription_segment_text(const wh_transcription* t, int i, char* out, int capacity); /* TranscriptionSegment.text */
int wh_transcription_word_text(const wh_transcription* t, int i, char* out, int capacity);    /* WordTiming.word */
int wh_transcription_word_tokens(const wh_transcription* t, const int32_t** tokens, int* n);  /* flat WordTiming.tokens */
int wh_transcription_seek_time(const wh_transcription* t, float* out);                    /* 1 + *out when seekTime != nil, else 0 */

/* SegmentSeeker.addWordTimestamps (Core/Text/SegmentSeeker.swift:410-496) for one window as a pure host function: DTW over
 * `alignment` ([alignment_rows][1500], row r = r-th token of the segments in order), word grouping (splitToWordTokens),
 * duration constraints, punctuation merge, updateSegmentsWithWordTimings.  `segments` index `tokens` / `logprobs`.
 * Returns a new transcription holding the updated segments (+ text), the words and their tokens. */
int wh_add_word_timestamps(const wh_tokenizer* tok, const char* language_code, const wh_special_tokens* st,
                           const wh_segment* segments, int n_segments, const int32_t* tokens, const float* logprobs, int n_tokens,
                           const float* alignment, int alignment_rows, int seek, float last_speech_timestamp,
                           int skip_special_tokens, wh_transcription** out);
/* SegmentSeeker.mergePunctuations(alignment:prepended:appended:) (Core/Text/SegmentSeeker.swift:280-338) on a caller-supplied
 * word list: words[i] (UTF-8) owns word_token_counts[i] consecutive ids of word_tokens.  NULL punctuation sets = the
 * reference defaults.  The merged words are returned as the words of a new transcription (wh_transcription_word*). */
int wh_merge_punctuations(const char* const* words, const int32_t* word_token_counts, const int32_t* word_tokens,
                          const float* start, const float* end, const float* probability, int n_words,
                          const char* prepended, const char* appended, wh_transcription** out);
/* The tail of addWordTimestamps after findAlignment (Core/Text/SegmentSeeker.swift:472-495) on a caller-supplied alignment:
 * calculateWordDurationConstraints (:498-507, returned through median_out / max_duration_out), truncateLongWordsAtSentenceBoundaries
 * (:509-526), mergePunctuations, updateSegmentsWithWordTimings (:528-659).  tok may be NULL unless a merged word mixes special
 * and text tokens (its text is then re-decoded). */
int wh_update_segments_with_word_timings(const wh_tokenizer* tok, int special_token_begin, const wh_segment* segments, int n_segments,
                                         const int32_t* tokens, int n_tokens, const char* const* words,
                                         const int32_t* word_token_counts, const int32_t* word_tokens, const float* start,
                                         const float* end, const float* probability, int n_words, int seek,
                                         float last_speech_timestamp, float* median_out, float* max_duration_out,
                                         wh_transcription** out);

/* ---- result assembly and on-disk formats ------------------------------------------------------------ */
/* TranscriptionResult(text:segments:language:timings:seekTime:) from parts (tok may be NULL: no text); seek_time NAN = nil */
int wh_transcription_create(const wh_tokenizer* tok, const wh_special_tokens* st, const wh_segment* segments, int n_segments,
                            const int32_t* tokens, const float* logprobs, int n_tokens, int language_token,
                            int skip_special_tokens, float seek_time, const wh_timings* timings, wh_transcription** out);
/* The "Windowing" block of TranscribeTask.run (Core/TranscribeTask.swift:175-265) for one decoded window, as a host
 * function (wh_transcribe* use it; the CPU tests drive it with synthetic decoding results): findSeekPointAndSegments, seek
 * never moves backward, optional addWordTimestamps from `alignment` ([224][1500], NULL = none; without a tokenizer every
 * text token is its own word), zero-length segments dropped, seek refined by the last word, maxWindowSeek clamp, segments
 * appended.  *seek_inout: this window's seek in, the next seek out. */
int wh_transcription_add_window(wh_transcription* t, const wh_tokenizer* tok, const wh_decoding_options* opt,
                                const wh_special_tokens* st, const wh_decoding_result* res, const float* alignment,
                                int default_language_token, int segment_size, int32_t* seek_inout);
/* finalizeTranscriptionResult (Core/TranscribeTask.swift:297-312): result text and language code (needs a tokenizer) */
int wh_transcription_finalize(wh_transcription* t, const wh_tokenizer* tok, const wh_decoding_options* opt,
                              const wh_special_tokens* st);
/* TranscriptionUtilities.mergeTranscriptionResults (Utilities/TranscriptionUtilities.swift:76-157); results[i] may be NULL
 * (a failed chunk); confirmed_words != NULL replaces the joined text by the concatenation of those words */
int wh_merge_transcriptions(const wh_transcription* const* results, int n, const char* const* confirmed_words, int n_confirmed,
                            wh_transcription** out);
/* AudioChunking.updateSeekOffsetsForResults (Core/Audio/AudioChunker.swift:14-39) for one chunk result: every segment / word
 * shifted by seek_offset_samples / 16000 (TranscriptionUtilities.updateSegmentTimings, Float), result.seekTime set */
int wh_transcription_apply_seek_offset(wh_transcription* t, int seek_offset_samples);
/* The Codable JSON document of TranscriptionResult as a string (= what wh_write_json writes; also the wire format of the
 * multi-GPU result gather, whisperkit_amd/parallel.py) and its decoder (JSONDecoder: unknown keys ignored). */
int wh_transcription_to_json(const wh_transcription* t, char* out, int capacity);
int wh_transcription_from_json(const char* json, int nbytes, wh_transcription** out);
/* ResultWriting.formatTime (Utilities/ResultWriter.swift:14-26) */
int wh_format_time(float seconds, int always_include_hours, char decimal_marker, char* out, int capacity);
/* WriteSRT / WriteVTT / WriteJSON (Utilities/ResultWriter.swift:40-134); `path` is the full file name */
int wh_write_srt(const wh_transcription* t, const char* path);
int wh_write_vtt(const wh_transcription* t, const char* path);
int wh_write_json(const wh_transcription* t, const char* path);

/* ---- audio ingest (Core/Audio/AudioProcessor.swift) ------------------------------------------------- */
enum { WH_CHANNEL_SPECIFIC = 0, WH_CHANNEL_SUM = 1 }; /* AudioInputConfig.ChannelMode :30-42 */
/* AudioProcessor.convertToMono (:525-625) on planar float channels; indices NULL / empty = all channels */
int wh_convert_to_mono(const float* const* channels, int n_channels, int n_frames, int mode, const int32_t* indices,
                       int n_indices, float* out);
/* resampleAudio (:458-519): equal rates pass through; other rates use a Kaiser-windowed sinc (AVAudioConverter's own filter
 * is unpublished - not sample-identical).  out == NULL returns the output length. */
int wh_resample(const float* in, int n_in, double in_rate, double out_rate, float* out, int capacity);
/* AudioProcessor.loadAudio(fromPath:channelMode:startTime:endTime:maxReadFrameSize:) (:229-300) for RIFF/WAVE files
 * (PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE) -> 16 kHz mono float; end_time NAN = nil,
 * max_read_frame_size 0 = Constants.defaultAudioReadFrameSize.  Free with wh_audio_free. */
int wh_load_audio(const char* path, int channel_mode, const int32_t* channel_indices, int n_channel_indices, double start_time,
                  double end_time, int max_read_frame_size, float** pcm_out, int* n_out);
void wh_audio_free(float* pcm);

/* ---- host utilities restated from the reference -------------------------------------------------- */
float wh_compression_ratio(const int32_t* tokens, int n);          /* TextUtilities.compressionRatio, Utilities/TextUtilities.swift:14-30 */
float wh_compression_ratio_text(const char* utf8, int nbytes);      /* TextUtilities.compressionRatio(of: String), :33-52 */
/* String.trimmingSpecialTokenCharacters (Constants.specialTokenCharacters, Core/Models.swift:1330): "<|en|>" -> "en" */
int wh_trimming_special_token_characters(const char* text, char* out, int capacity);
/* SegmentSeeker.dynamicTimeWarping (Core/Text/SegmentSeeker.swift:195-278); returns path length */
int wh_dynamic_time_warping(const float* matrix, int rows, int cols, int32_t* text_indices, int32_t* time_indices, int capacity);
/* DecodingFallback.init? (Core/Models.swift:357-381) -> WH_FALLBACK_*; *needs_fallback set */
int wh_decoding_fallback(const wh_decoding_options* opt, int is_first_token_logprob_too_low, float no_speech_prob,
                         float compression_ratio, float avg_logprob, int32_t* needs_fallback);
/* SegmentSeeker.findSeekPointAndSegments (Core/Text/SegmentSeeker.swift:41-189); returns number of segments or -1 (skip) */
int wh_find_seek_point_and_segments(const wh_decoding_result* res, const wh_decoding_options* opt,
                                    const wh_special_tokens* st, int all_segments_count, int current_seek,
                                    int segment_size, int32_t* new_seek, wh_segment* segments_out, int capacity);
/* DecodingOptions.prepareSeekClips(contentFrames:) (Utilities/Extensions+Internal.swift:112-130): clipTimestamps (seconds) ->
 * [start, end) sample pairs; returns the number of clips (negative: -needed when capacity is too small) */
int wh_prepare_seek_clips(const wh_decoding_options* opt, int content_frames, int32_t* clip_start, int32_t* clip_end, int capacity);
/* EnergyVAD.voiceActivity (Core/Audio/EnergyVAD.swift:41-57); returns number of frames written */
int wh_vad_voice_activity(const float* pcm, int n, int frame_length_samples, int frame_overlap_samples,
                          float energy_threshold, uint8_t* out, int capacity);
/* VADAudioChunker.chunkAll (Core/Audio/AudioChunker.swift:66-107); returns number of chunks */
int wh_vad_chunk_all(const float* pcm, int n, int max_chunk_length, const wh_decoding_options* opt,
                     int32_t* chunk_start, int32_t* chunk_end, int capacity);

/* ---- multi-GPU: chunk partition + result gather across one-process-per-GPU ranks ------------------------------------
 * The reference fans independent audio arrays out over a TaskGroup sharing the model objects (Core/WhisperKit.swift:735-812)
 * and merges per-chunk results in process (Utilities/TranscriptionUtilities.swift:76-157; chunk offsets applied by
 * AudioChunking.updateSeekOffsetsForResults, Core/Audio/AudioChunker.swift:14-39).  With one process per GPU the chunks are
 * block-partitioned over the ranks (weights replicated, no data-path collective) and the merge needs ONE all-gather of the
 * per-chunk results - the only collective of the path.  Transports:
 *   WH_COMM_RCCL  ncclAllGather over xGMI; librccl is dlopen'ed on first use (a copy already in the process is reused), `id` is
 *                 RCCL's 128-byte ncclUniqueId made by wh_comm_unique_id on rank 0 and handed to the other ranks by the caller;
 *   WH_COMM_TCP   a star over host TCP sockets (rank 0 listens at "host:port", carried in `id`): hosts without RCCL, CPU-only
 *                 tests, several ranks sharing one GPU (RCCL refuses duplicate devices).
 * Calls on one communicator are collective: every rank makes the same sequence of calls. */
typedef struct wh_comm wh_comm;
enum { WH_COMM_RCCL = 0, WH_COMM_TCP = 1 };
#define WH_COMM_ID_BYTES 128
#define WH_RECORD_TOKENS 232
/* what the merge needs of one chunk's DecodingResult: 240 x 4 bytes */
typedef struct wh_chunk_record {
    int32_t tokens[WH_RECORD_TOKENS];
    int32_t n_tokens, chunk_index /* < 0: padding */, seek /* chunk offset in samples */, steps;
    float avg_logprob, temperature, compression_ratio, no_speech_prob;
} wh_chunk_record;
int wh_comm_unique_id(int transport, const char* tcp_address /* "host:port" of rank 0, TCP only */, uint8_t* id /* [WH_COMM_ID_BYTES] */);
/* `device`: the HIP device of this rank (RCCL staging buffers live there); id may be NULL when world_size == 1 */
int wh_comm_create(int transport, const uint8_t* id, int world_size, int rank, int device, wh_comm** out);
void wh_comm_destroy(wh_comm* c);
int wh_comm_rank(const wh_comm* c);
int wh_comm_world_size(const wh_comm* c);
int wh_comm_transport(const wh_comm* c);
/* rank r owns chunks [start, end) of n_chunks: contiguous blocks, sizes differ by at most one, output order kept */
int wh_partition_chunks(int n_chunks, int world_size, int rank, int* start, int* end);
int wh_comm_barrier(wh_comm* c);
/* nbytes from every rank -> world_size * nbytes on every rank, in rank order (host buffers) */
int wh_comm_all_gather(wh_comm* c, const void* send, void* recv, size_t nbytes);
int wh_chunk_record_from_result(const wh_decoding_result* res, int chunk_index, int seek, wh_chunk_record* out);
/* every rank passes its n_local <= max_per_rank records (same max_per_rank everywhere) and receives all records by chunk index */
int wh_comm_gather_records(wh_comm* c, const wh_chunk_record* local, int n_local, int max_per_rank, wh_chunk_record* all_out,
                           int capacity, int* n_out);
/* whole TranscriptionResults (as their Codable JSON documents): every rank receives every rank's results in chunk order as new
 * handles it owns; feed them to wh_merge_transcriptions (mergeTranscriptionResults) */
int wh_comm_gather_transcriptions(wh_comm* c, const wh_transcription* const* local, const int32_t* chunk_indices, int n_local,
                                  wh_transcription** all_out, int32_t* chunk_indices_out /* may be NULL */, int capacity, int* n_out);

/* ---- measurement (bench.py roofline leg; no reference analogue) --------------------------------------
 * One eager pass of the hot path on the session stream - log-mel, encoder, cross-K/V projection, then n_steps decoder
 * steps - with a HIP event pair around every kernel launch.  avg_us / launches are indexed by kernel kind
 * [0, wh_kernel_kind_count()); wh_kernel_kind_name(k) names kind k.  Uses the prompt / sampler configuration of the last
 * wh_decode_text call on the session and the audio currently in its slots. */
int wh_kernel_kind_count(void);
const char* wh_kernel_kind_name(int kind);
int wh_measure_kernels(wh_session* s, int batch, int n_steps, double* avg_us, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* WHISPERHIP_H */

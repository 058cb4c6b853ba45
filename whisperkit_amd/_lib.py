"""ctypes declarations for libwhisperhip.so (include/whisperhip.h).  There is no CPU fallback: if the
shared library is missing this module raises, and every compute entry point needs a visible gfx950 GPU."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WHISPERHIP_LIB") or os.path.join(HERE, "libwhisperhip.so")      # WHISPERHIP_LIB: another build of the SAME ABI (A/B probes)

WH_MAX_RESULT_TOKENS = 232
WINDOW_SAMPLES = 480000
MEL_FRAMES = 3000
AUDIO_CTX = 1500
MAX_TOKEN_CONTEXT = 224


class WhDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                           "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class WhSpecialTokens(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("end_token", "english_token", "no_speech_token", "no_timestamps_token",
                                           "special_token_begin", "start_of_previous_token", "start_of_transcript_token",
                                           "time_token_begin", "transcribe_token", "translate_token", "whitespace_token",
                                           "language_token_begin", "n_language_tokens")]


class WhDecodingOptions(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("language_token", C.c_int32), ("temperature", C.c_float),
        ("temperature_increment_on_fallback", C.c_float), ("temperature_fallback_count", C.c_int32),
        ("sample_length", C.c_int32), ("top_k", C.c_int32), ("use_prefill_prompt", C.c_int32),
        ("detect_language", C.c_int32), ("skip_special_tokens", C.c_int32), ("without_timestamps", C.c_int32),
        ("word_timestamps", C.c_int32), ("max_initial_timestamp", C.c_float), ("max_window_seek", C.c_int32),
        ("clip_timestamps", C.POINTER(C.c_float)), ("n_clip_timestamps", C.c_int32), ("window_clip_time", C.c_float),
        ("prompt_tokens", C.POINTER(C.c_int32)), ("n_prompt_tokens", C.c_int32),
        ("prefix_tokens", C.POINTER(C.c_int32)), ("n_prefix_tokens", C.c_int32),
        ("suppress_blank", C.c_int32), ("suppress_tokens", C.POINTER(C.c_int32)), ("n_suppress_tokens", C.c_int32),
        ("compression_ratio_threshold", C.c_float), ("log_prob_threshold", C.c_float),
        ("first_token_log_prob_threshold", C.c_float), ("no_speech_threshold", C.c_float), ("seed", C.c_uint64),
        ("float16_logits", C.c_int32), ("beam_size", C.c_int32), ("beam_patience", C.c_float), ("reserved_", C.c_int32),
    ]


class WhDecodingResult(C.Structure):
    _fields_ = [
        ("n_tokens", C.c_int32), ("tokens", C.c_int32 * WH_MAX_RESULT_TOKENS), ("token_logprobs", C.c_float * WH_MAX_RESULT_TOKENS),
        ("avg_logprob", C.c_float), ("no_speech_prob", C.c_float), ("temperature", C.c_float), ("compression_ratio", C.c_float),
        ("language_token", C.c_int32), ("fallback_reason", C.c_int32), ("needs_fallback", C.c_int32),
        ("is_first_token_logprob_too_low", C.c_int32), ("steps", C.c_int32),
    ]


class WhSessionOptions(C.Structure):
    _fields_ = [("cross_attention_mode", C.c_int32), ("cross_attention_splits", C.c_int32), ("cross_attention_slots_per_workgroup", C.c_int32),
                ("reserved_", C.c_int32 * 5)]


class WhTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4), ("device", C.c_int32),
                ("reserved_", C.c_int32)]


class WhSegment(C.Structure):
    _fields_ = [("id", C.c_int32), ("seek", C.c_int32), ("start", C.c_float), ("end", C.c_float),
                ("token_offset", C.c_int32), ("n_tokens", C.c_int32), ("temperature", C.c_float), ("avg_logprob", C.c_float),
                ("compression_ratio", C.c_float), ("no_speech_prob", C.c_float), ("word_offset", C.c_int32), ("n_words", C.c_int32)]


class WhWordTiming(C.Structure):
    _fields_ = [("token_offset", C.c_int32), ("n_tokens", C.c_int32), ("start", C.c_float), ("end", C.c_float), ("probability", C.c_float)]


class WhProgress(C.Structure):
    _fields_ = [("slot", C.c_int32), ("n_tokens", C.c_int32), ("tokens", C.POINTER(C.c_int32)), ("avg_logprob", C.c_float),
                ("compression_ratio", C.c_float), ("text", C.c_char_p)]


PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(WhProgress))
WINDOW_PRE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int)
WINDOW_POST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int)
SEGMENT_DISCOVERY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int)


class WhWindowHooks(C.Structure):
    _fields_ = [("window_preprocess", WINDOW_PRE_FN), ("window_postprocess", WINDOW_POST_FN), ("segment_discovery", SEGMENT_DISCOVERY_FN),
                ("user", C.c_void_p)]
# wh_logits_filter_fn / wh_token_sampler_fn: LogitsFiltering.filterLogits / TokenSampling.update as C callbacks (wh_decode_text_custom)
LOGITS_FILTER_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32), C.c_int32)
TOKEN_SAMPLER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int32,
                               C.POINTER(C.c_int32), C.POINTER(C.c_float))


class WhChunkRecord(C.Structure):
    """wh_chunk_record: what the cross-rank merge needs of one chunk's DecodingResult (960 bytes)"""
    _fields_ = [("tokens", C.c_int32 * 232), ("n_tokens", C.c_int32), ("chunk_index", C.c_int32), ("seek", C.c_int32), ("steps", C.c_int32),
                ("avg_logprob", C.c_float), ("temperature", C.c_float), ("compression_ratio", C.c_float), ("no_speech_prob", C.c_float)]


COMM_RCCL, COMM_TCP, COMM_ID_BYTES = 0, 1, 128


class WhTimings(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "audio_processing", "logmels", "encoding", "decoding_init", "decoding_predictions", "decoding_filtering",
        "decoding_sampling", "decoding_kv_caching", "decoding_word_timestamps", "decoding_fallback", "decoding_windowing",
        "decoding_loop", "full_pipeline", "input_audio_seconds", "total_decoding_loops", "total_decoding_windows",
        "total_decoding_fallbacks", "total_encoding_runs", "total_logmel_runs",
        "pipeline_start", "first_token_time", "model_loading", "prewarm_load_time", "encoder_load_time", "decoder_load_time",
        "encoder_specialization_time", "decoder_specialization_time", "tokenizer_load_time", "audio_loading",
        "decoding_non_prediction", "total_audio_processing_runs", "total_kv_update_runs", "total_timestamp_alignment_runs")]


# every symbol include/whisperhip.h declares: (restype, argtypes)
VP, I, F, U64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
PF, PI32, PU8 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
PVP = C.POINTER(C.c_void_p)
POPT, PST = C.POINTER(WhDecodingOptions), C.POINTER(WhSpecialTokens)
SYMBOLS = {
    "wh_last_error": (C.c_char_p, []),
    "wh_version": (C.c_char_p, []),
    "wh_model_create": (I, [VP, C.c_size_t, I, PVP]),
    "wh_model_load": (I, [C.c_char_p, I, PVP]),
    "wh_model_destroy": (None, [VP]),
    "wh_model_dims": (I, [VP, C.POINTER(WhDims)]),
    "wh_model_set_alignment_heads": (I, [VP, PI32, I]),
    "wh_mel_count": (I, [VP]), "wh_window_samples": (I, [VP]), "wh_embed_size": (I, [VP]), "wh_logits_size": (I, [VP]),
    "wh_kv_cache_embed_dim": (I, [VP]), "wh_kv_cache_max_sequence_length": (I, [VP]), "wh_window_size": (I, [VP]),
    "wh_is_model_multilingual": (I, [VP]), "wh_supports_word_timestamps": (I, [VP]),
    "wh_special_tokens_default": (I, [VP, PST]),
    "wh_decoding_options_default": (None, [POPT]),
    "wh_session_create": (I, [VP, I, PVP]),
    "wh_session_destroy": (None, [VP]),
    "wh_session_max_batch": (I, [VP]),
    "wh_session_cross_attention_mode": (I, [VP]),
    "wh_session_cross_attention_splits": (I, [VP]),
    "wh_xabs_auto_min_slots": (I, []),
    "wh_xabs_auto_splits": (I, [I]),
    "wh_session_step_graph_count": (I, [VP]),
    "wh_session_set_window_hooks": (I, [VP, C.POINTER(WhWindowHooks)]),
    "wh_transcription_set_segment_times": (I, [VP, I, F, F]),
    "wh_session_create_with_mode": (I, [VP, I, I, PVP]),
    "wh_session_create_tuned": (I, [VP, I, I, I, PVP]),
    "wh_session_options_default": (None, [C.POINTER(WhSessionOptions)]),
    "wh_session_create_with_options": (I, [VP, I, C.POINTER(WhSessionOptions), PVP]),
    "wh_session_cross_attention_slots_per_workgroup": (I, [VP]),
    "wh_debug_peek": (I, [VP, C.c_char_p, VP, C.c_size_t]),
    "wh_session_synchronize": (I, [VP]),
    "wh_session_stream": (VP, [VP]),
    "wh_set_audio": (I, [VP, I, VP, I]),
    "wh_set_audio_device": (I, [VP, I, VP, I]),
    "wh_log_mel_spectrogram": (I, [VP, I]),
    "wh_get_mel": (I, [VP, I, VP]),
    "wh_set_mel": (I, [VP, I, VP]),
    "wh_encode_features": (I, [VP, I]),
    "wh_get_encoder_output": (I, [VP, I, VP]),
    "wh_set_encoder_output": (I, [VP, I, VP]),
    "wh_prepare_decoder_inputs": (I, [VP, I]),
    "wh_reset_decoder_inputs": (I, [VP, I]),
    "wh_predict_logits": (I, [VP, I, PI32, PI32, VP]),
    "wh_get_alignment_weights": (I, [VP, I, VP]),
    "wh_filter_logits": (I, [VP, POPT, PST, PI32, I, I, I, I, VP, I]),
    "wh_sample_token": (I, [VP, VP, I, F, I, U64, I, PI32, PF]),
    "wh_decode_text": (I, [VP, I, POPT, PST, PI32, I, PF, PI32, U64, C.POINTER(WhDecodingResult)]),
    "wh_decode_text_languages": (I, [VP, I, POPT, PST, PI32, I, PI32, PF, PI32, U64, C.POINTER(WhDecodingResult)]),
    "wh_decode_text_custom": (I, [VP, POPT, PST, PI32, I, F, U64, C.POINTER(LOGITS_FILTER_FN), PVP, I, TOKEN_SAMPLER_FN, VP,
                                  C.POINTER(WhDecodingResult)]),
    "wh_detect_language": (I, [VP, I, PST, PI32, PF]),
    "wh_decode_text_beam": (I, [VP, I, I, F, POPT, PST, PI32, I, PI32, C.POINTER(WhDecodingResult)]),
    "wh_beam_sampler_create": (I, [I, C.c_int32, F, PVP]),
    "wh_beam_sampler_destroy": (None, [VP]),
    "wh_beam_sampler_reset": (None, [VP]),
    "wh_beam_sampler_max_candidates": (I, [VP]),
    "wh_beam_sampler_finished_count": (I, [VP]),
    "wh_beam_sampler_update": (I, [VP, I, I, PI32, PF, PF, PF, PI32, I, PI32, PF, PF, PI32, PI32, PI32]),
    "wh_beam_sampler_finalize": (I, [VP, I, I, PI32, PF, PF, I, I, PI32, PF, PI32, PF, PI32]),
    "wh_get_mel_device": (I, [VP, I, PVP]),
    "wh_get_mel_tensor": (I, [VP, I, C.POINTER(WhTensor)]),
    "wh_get_encoder_output_tensor": (I, [VP, I, I, C.POINTER(WhTensor)]),
    "wh_get_logits_tensor": (I, [VP, C.POINTER(WhTensor)]),
    "wh_get_encoder_output_device": (I, [VP, I, PVP, PVP]),
    "wh_get_logits_device": (I, [VP, PVP]),
    "wh_session_set_cancel_flag": (I, [VP, VP]),
    "wh_session_set_alignment_postprocess": (I, [VP, I, I]),
    "wh_prefill_prompt": (I, [VP, POPT, PST, C.c_int32, PI32, I]),
    "wh_transcribe": (I, [VP, VP, I, POPT, PST, PVP]),
    "wh_transcribe_batch": (I, [VP, PVP, PI32, I, POPT, PST, PVP]),
    "wh_transcribe_batch_with_options": (I, [VP, PVP, PI32, I, C.POINTER(POPT), PST, PVP, PI32]),
    "wh_session_item_status": (I, [VP, I]), "wh_session_item_error": (C.c_char_p, [VP, I]),
    "wh_transcribe_chunked": (I, [VP, VP, I, POPT, PST, PVP, I, PI32, C.POINTER(I)]),
    "wh_transcription_free": (None, [VP]),
    "wh_transcription_n_segments": (I, [VP]),
    "wh_transcription_segment": (I, [VP, I, C.POINTER(WhSegment)]),
    "wh_transcription_n_words": (I, [VP]),
    "wh_transcription_word": (I, [VP, I, C.POINTER(WhWordTiming)]),
    "wh_transcription_tokens": (I, [VP, C.POINTER(PI32), C.POINTER(PF), C.POINTER(I)]),
    "wh_transcription_language_token": (I, [VP]),
    "wh_transcription_timings": (I, [VP, C.POINTER(WhTimings)]),
    "wh_transcription_window_seeks": (I, [VP, C.POINTER(PI32), C.POINTER(I)]),
    # tokenizer text, result assembly, formats, audio ingest (host only)
    "wh_tokenizer_load": (I, [C.c_char_p, PVP]),
    "wh_tokenizer_destroy": (None, [VP]),
    "wh_tokenizer_vocab_size": (I, [VP]),
    "wh_tokenizer_decode": (I, [VP, PI32, I, I, C.c_char_p, I]),
    "wh_tokenizer_token_to_id": (I, [VP, C.c_char_p]),
    "wh_tokenizer_id_to_token": (I, [VP, I, C.c_char_p, I]),
    "wh_tokenizer_special_tokens": (I, [VP, PST]),
    "wh_tokenizer_split_to_word_tokens": (I, [VP, PI32, I, C.c_char_p, PI32, PI32, I, C.c_char_p, I, C.POINTER(I)]),
    "wh_session_set_tokenizer": (I, [VP, VP]),
    "wh_session_set_progress_callback": (I, [VP, PROGRESS_FN, VP]),
    "wh_transcription_has_text": (I, [VP]),
    "wh_transcription_text": (I, [VP, C.c_char_p, I]),
    "wh_transcription_language": (I, [VP, C.c_char_p, I]),
    "wh_transcription_segment_text": (I, [VP, I, C.c_char_p, I]),
    "wh_transcription_word_text": (I, [VP, I, C.c_char_p, I]),
    "wh_transcription_word_tokens": (I, [VP, C.POINTER(PI32), C.POINTER(I)]),
    "wh_transcription_seek_time": (I, [VP, PF]),
    "wh_add_word_timestamps": (I, [VP, C.c_char_p, PST, C.POINTER(WhSegment), I, PI32, PF, I, PF, I, I, F, I, PVP]),
    "wh_merge_punctuations": (I, [C.POINTER(C.c_char_p), PI32, PI32, PF, PF, PF, I, C.c_char_p, C.c_char_p, PVP]),
    "wh_update_segments_with_word_timings": (I, [VP, I, C.POINTER(WhSegment), I, PI32, I, C.POINTER(C.c_char_p), PI32, PI32, PF, PF, PF, I, I, F,
                                                 PF, PF, PVP]),
    "wh_transcription_create": (I, [VP, PST, C.POINTER(WhSegment), I, PI32, PF, I, I, I, F, C.POINTER(WhTimings), PVP]),
    "wh_transcription_add_window": (I, [VP, VP, POPT, PST, C.POINTER(WhDecodingResult), PF, I, I, PI32]),
    "wh_transcription_finalize": (I, [VP, VP, POPT, PST]),
    "wh_transcription_apply_seek_offset": (I, [VP, I]),
    "wh_transcription_to_json": (I, [VP, C.c_char_p, I]),
    "wh_transcription_from_json": (I, [C.c_char_p, I, PVP]),
    "wh_merge_transcriptions": (I, [PVP, I, C.POINTER(C.c_char_p), I, PVP]),
    "wh_format_time": (I, [F, I, C.c_char, C.c_char_p, I]),
    "wh_write_srt": (I, [VP, C.c_char_p]),
    "wh_write_vtt": (I, [VP, C.c_char_p]),
    "wh_write_json": (I, [VP, C.c_char_p]),
    "wh_convert_to_mono": (I, [PVP, I, I, I, PI32, I, PF]),
    "wh_resample": (I, [PF, I, C.c_double, C.c_double, PF, I]),
    "wh_load_audio": (I, [C.c_char_p, I, PI32, I, C.c_double, C.c_double, I, C.POINTER(PF), C.POINTER(I)]),
    "wh_audio_free": (None, [PF]),
    "wh_compression_ratio": (F, [PI32, I]),
    "wh_compression_ratio_text": (F, [C.c_char_p, I]),
    "wh_trimming_special_token_characters": (I, [C.c_char_p, C.c_char_p, I]),
    "wh_dynamic_time_warping": (I, [PF, I, I, PI32, PI32, I]),
    "wh_decoding_fallback": (I, [POPT, I, F, F, F, PI32]),
    "wh_find_seek_point_and_segments": (I, [C.POINTER(WhDecodingResult), POPT, PST, I, I, I, PI32, C.POINTER(WhSegment), I]),
    "wh_prepare_seek_clips": (I, [POPT, I, PI32, PI32, I]),
    "wh_vad_voice_activity": (I, [PF, I, I, I, F, PU8, I]),
    "wh_vad_chunk_all": (I, [PF, I, I, POPT, PI32, PI32, I]),
    "wh_comm_unique_id": (I, [I, C.c_char_p, PU8]),
    "wh_comm_create": (I, [I, PU8, I, I, I, PVP]),
    "wh_comm_destroy": (None, [VP]),
    "wh_comm_rank": (I, [VP]), "wh_comm_world_size": (I, [VP]), "wh_comm_transport": (I, [VP]),
    "wh_partition_chunks": (I, [I, I, I, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "wh_comm_barrier": (I, [VP]),
    "wh_comm_all_gather": (I, [VP, VP, VP, C.c_size_t]),
    "wh_chunk_record_from_result": (I, [C.POINTER(WhDecodingResult), I, I, C.POINTER(WhChunkRecord)]),
    "wh_comm_gather_records": (I, [VP, C.POINTER(WhChunkRecord), I, I, C.POINTER(WhChunkRecord), I, C.POINTER(C.c_int)]),
    "wh_comm_gather_transcriptions": (I, [VP, PVP, PI32, I, PVP, PI32, I, C.POINTER(C.c_int)]),
    "wh_kernel_kind_count": (I, []),
    "wh_kernel_kind_name": (C.c_char_p, [I]),
    "wh_measure_kernels": (I, [VP, I, I, C.POINTER(C.c_double), PI32]),
}

_lib = None


def load():
    """Load libwhisperhip.so (built in-tree by `__graft_entry__.build()` / `make -C whisperkit_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C whisperkit_amd/csrc` "
                          "(the whisperhip product path has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the header and the library disagree
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib

// Result assembly, on-disk formats and audio ingest around the hot path (host only, no GPU):
//   TranscriptionUtilities.mergeTranscriptionResults   Utilities/TranscriptionUtilities.swift:76-157
//   WriteJSON / WriteSRT / WriteVTT, formatTime         Utilities/ResultWriter.swift:12-134
//   TranscriptionResult / Segment / WordTiming Codable  Core/Models.swift:447-641, TranscriptionTimings :730-844
//   AudioProcessor.loadAudio / convertToMono            Core/Audio/AudioProcessor.swift:229-300, 381-456, 525-625
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <zlib.h>

#include "json.h"
#include "text.h"

using whi::set_error;

static int copy_out(const std::string& s, char* out, int capacity) {
    if (out && capacity > 0) {
        size_t n = std::min(s.size(), (size_t)capacity - 1);
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int)s.size();
}

// ---- text accessors --------------------------------------------------------------------------------------------------------
extern "C" int wh_transcription_has_text(const wh_transcription* t) { return t && t->has_text; }
extern "C" int wh_transcription_text(const wh_transcription* t, char* out, int capacity) {
    if (!t || !t->has_text) { set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "transcription has no text (no tokenizer attached)"); return -1; }
    return copy_out(t->text, out, capacity);
}
extern "C" int wh_transcription_language(const wh_transcription* t, char* out, int capacity) {
    if (!t || !t->has_text) { set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "transcription has no text (no tokenizer attached)"); return -1; }
    return copy_out(t->language, out, capacity);
}
extern "C" int wh_transcription_segment_text(const wh_transcription* t, int i, char* out, int capacity) {
    if (!t || !t->has_text || i < 0 || i >= (int)t->segment_text.size()) { set_error(WH_ERR_INVALID_ARGUMENT, "no text for segment %d", i); return -1; }
    return copy_out(t->segment_text[i], out, capacity);
}
extern "C" int wh_transcription_word_text(const wh_transcription* t, int i, char* out, int capacity) {
    if (!t || !t->has_text || i < 0 || i >= (int)t->word_text.size()) { set_error(WH_ERR_INVALID_ARGUMENT, "no text for word %d", i); return -1; }
    return copy_out(t->word_text[i], out, capacity);
}
extern "C" int wh_transcription_word_tokens(const wh_transcription* t, const int32_t** tokens, int* n) {
    if (!t || !n) return set_error(WH_ERR_INVALID_ARGUMENT, "null transcription");
    if (tokens) *tokens = t->word_tokens.data();
    *n = (int)t->word_tokens.size();
    return WH_OK;
}
extern "C" int wh_transcription_seek_time(const wh_transcription* t, float* out) {
    if (!t) return 0;
    if (out && t->has_seek_time) *out = t->seek_time;
    return t->has_seek_time ? 1 : 0;
}

// TranscriptionResult(text:segments:language:timings:seekTime:) from parts: segment texts and the result text are decoded
// like SegmentSeeker.swift:118-121 / TranscribeTask.swift:303-305.  seek_time = NAN means nil.
extern "C" int wh_transcription_create(const wh_tokenizer* tok, const wh_special_tokens* st, const wh_segment* segments, int n_segments,
                                       const int32_t* tokens, const float* logprobs, int n_tokens, int language_token,
                                       int skip_special_tokens, float seek_time, const wh_timings* timings, wh_transcription** out) {
    if (!st || !out || n_segments < 0 || n_tokens < 0 || (n_segments && !segments) || (n_tokens && (!tokens || !logprobs)))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_create: null argument");
    for (int s = 0; s < n_segments; ++s)
        if (segments[s].token_offset < 0 || segments[s].n_tokens < 0 || segments[s].token_offset + segments[s].n_tokens > n_tokens)
            return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_create: segment %d indexes outside the token array", s);
    auto tr = new wh_transcription();
    tr->tokens.assign(tokens, tokens + n_tokens);
    tr->logprobs.assign(logprobs, logprobs + n_tokens);
    tr->segments.assign(segments, segments + n_segments);
    for (auto& g : tr->segments) { g.word_offset = 0; g.n_words = 0; }
    tr->language_token = language_token;
    if (timings) tr->timings = *timings;
    if (!std::isnan(seek_time)) { tr->seek_time = seek_time; tr->has_seek_time = true; }
    if (tok) {
        tr->has_text = true;
        std::vector<int> all;
        for (auto& g : tr->segments) {
            std::vector<int> t;
            for (int k = 0; k < g.n_tokens; ++k) {
                int id = tr->tokens[g.token_offset + k];
                if (!skip_special_tokens || id < st->special_token_begin) t.push_back(id);
                if (id < st->special_token_begin) all.push_back(id);
            }
            tr->segment_text.push_back(tok->decode(t));
        }
        tr->text = whi::trim_swift_whitespaces(tok->decode(all));
        tr->language = "en";
        if (language_token >= 0) {
            std::string c = whi::trimming_special_token_characters(tok->decode(&language_token, 1, false));
            if (!c.empty()) tr->language = c;
        }
    }
    *out = tr;
    return WH_OK;
}

// ---- TranscribeTask.run windowing ------------------------------------------------------------------------------------------
namespace whi {
int add_word_timestamps(const wh_tokenizer* tok, const char* language, int special_begin, wh_segment* segments, int n_segments,
                        const int32_t* tokens, const float* logprobs, const float* alignment, int alignment_rows, int seek,
                        float last_speech_timestamp, wh_transcription* tr);   // words.cpp
}

static std::string language_code_of(const wh_tokenizer* tok, int language_token) {
    // decodeText: language = tokenizer.decode([languageToken]).trimmingSpecialTokenCharacters() (TextDecoder.swift:814), default "en"
    if (!tok || language_token < 0) return "en";
    std::string c = whi::trimming_special_token_characters(tok->decode(&language_token, 1, false));
    return c.empty() ? std::string("en") : c;
}

// Without a tokenizer the words cannot be grouped by text: every text token becomes one word timed by DTW over its alignment row
// (findAlignment, SegmentSeeker.swift:340-408, with one token per word; no punctuation merge, no duration constraints).
static void add_token_timestamps(const float* full, const wh_decoding_result& res, const wh_special_tokens* st, wh_segment* segs, int ns,
                                 int seek, wh_transcription* win) {
    const int n = res.n_tokens, cols = WH_AUDIO_CTX;
    int cap = n + cols + 8;
    std::vector<int32_t> ti(cap), tj(cap);
    int len = wh_dynamic_time_warping(full, n, cols, ti.data(), tj.data(), cap);
    if (len <= 0) return;
    std::vector<float> startT{0.0f}, endT;
    int cur = ti[0];
    for (int k = 0; k < len; ++k)
        if (ti[k] != cur) { cur = ti[k]; float t = (float)tj[k] * 0.02f; startT.push_back(t); endT.push_back(t); }
    endT.push_back((float)tj[len - 1] * 0.02f);
    const float timeOffset = (float)seek / (float)WH_SAMPLE_RATE;
    for (int si = 0; si < ns; ++si) {
        wh_segment& g = segs[si];
        g.word_offset = (int)win->words.size();
        for (int k = 0; k < g.n_tokens; ++k) {
            int ri = g.token_offset + k;   // index into the window's token list == alignment row
            if (ri < 0 || ri >= n || ri >= (int)startT.size() || ri >= (int)endT.size()) continue;
            if (res.tokens[ri] >= st->special_token_begin) continue;
            wh_word_timing w{};
            w.token_offset = (int)win->word_tokens.size(); w.n_tokens = 1;
            win->word_tokens.push_back(res.tokens[ri]);
            w.start = whi::rounded2(timeOffset + startT[ri]);
            w.end = whi::rounded2(timeOffset + endT[ri]);
            w.probability = whi::rounded2(expf(res.token_logprobs[ri]));
            win->words.push_back(w);
            win->word_text.emplace_back();
        }
        g.n_words = (int)win->words.size() - g.word_offset;
    }
}

// The "Windowing" block of TranscribeTask.run (Core/TranscribeTask.swift:175-265) for one decoded window: findSeekPointAndSegments,
// seek never moves backward, optional addWordTimestamps (zero-length segments dropped, seek refined with the last word's end),
// maxWindowSeek clamp, segments / tokens appended to the transcription.  `alignment` = [224][1500] alignment weights of the window
// or NULL (no word timestamps).  *seek_inout: window seek in, next seek out.
extern "C" int wh_transcription_add_window(wh_transcription* tr, const wh_tokenizer* tok, const wh_decoding_options* opt,
                                           const wh_special_tokens* st, const wh_decoding_result* res, const float* alignment,
                                           int default_language_token, int segment_size, int32_t* seek_inout) {
    if (!tr || !opt || !st || !res || !seek_inout) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_add_window: null argument");
    if (res->n_tokens < 0 || res->n_tokens > WH_MAX_RESULT_TOKENS)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_add_window: n_tokens %d outside [0, %d]", res->n_tokens, WH_MAX_RESULT_TOKENS);
    if (!tr->language_set) { tr->language_token = res->language_token; tr->language_set = true; }   // "Use the predicted language if it was not detected ahead of time"
    const int prev_seek = *seek_inout;
    wh_segment segs[WH_MAX_RESULT_TOKENS];
    int32_t new_seek = prev_seek;
    int ns = wh_find_seek_point_and_segments(res, opt, st, (int)tr->segments.size(), prev_seek, segment_size, &new_seek, segs, WH_MAX_RESULT_TOKENS);
    if (ns < -1) return set_error(WH_ERR_SEGMENTING_FAILED, "findSeekPointAndSegments failed");
    int seek = std::max(prev_seek, (int)new_seek);
    wh_transcription win;   // window-local words
    const bool words = opt->word_timestamps && alignment;
    const auto tw0 = std::chrono::steady_clock::now();
    if (words) {
        if (ns < 0) ns = 0;   // `currentSegments ?? []` (:202)
        const int window_language = res->language_token >= 0 ? res->language_token : default_language_token;
        if (tok) {
            const std::string code = language_code_of(tok, window_language);
            int r = whi::add_word_timestamps(tok, code.c_str(), st->special_token_begin, segs, ns, res->tokens, res->token_logprobs, alignment,
                                             WH_MAX_TOKEN_CONTEXT, prev_seek, (float)((double)prev_seek / (double)WH_SAMPLE_RATE), &win);
            if (r) return r;
        } else {
            add_token_timestamps(alignment, *res, st, segs, ns, prev_seek, &win);
        }
        tr->timings.total_timestamp_alignment_runs += 1;
        tr->words_enabled = true;
        tr->timings.decoding_word_timestamps += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        int kept = 0;                                   // "Filter out zero length segments" (:217)
        for (int i = 0; i < ns; ++i) if (segs[i].end > segs[i].start) segs[kept++] = segs[i];
        ns = kept;
        if (ns > 0) seek = std::max(seek, (int)(segs[ns - 1].end * (float)WH_SAMPLE_RATE));   // (:220-222)
    }
    if (opt->max_window_seek >= 0) seek = std::min(seek, prev_seek + opt->max_window_seek);   // (:234-237)
    *seek_inout = seek;
    if (ns < 0) return WH_OK;                           // no segment for this window: skip to the next (:239-242)
    for (int i = 0; i < ns; ++i) {
        wh_segment g = segs[i];
        const int src = g.token_offset;
        g.token_offset = (int)tr->tokens.size();
        std::vector<int> text_ids;
        for (int k = 0; k < g.n_tokens; ++k) {
            const int id = res->tokens[src + k];
            tr->tokens.push_back(id);
            tr->logprobs.push_back(res->token_logprobs[src + k]);
            if (!opt->skip_special_tokens || id < st->special_token_begin) text_ids.push_back(id);
        }
        const int wsrc = g.word_offset;
        g.word_offset = (int)tr->words.size();
        if (!words) g.n_words = 0;
        for (int k = 0; k < g.n_words; ++k) {
            wh_word_timing w = win.words[wsrc + k];
            const int tsrc = w.token_offset;
            w.token_offset = (int)tr->word_tokens.size();
            tr->word_tokens.insert(tr->word_tokens.end(), win.word_tokens.begin() + tsrc, win.word_tokens.begin() + tsrc + w.n_tokens);
            tr->words.push_back(w);
            tr->word_text.push_back(win.word_text[wsrc + k]);
        }
        tr->segments.push_back(g);
        if (tok) tr->segment_text.push_back(tok->decode(text_ids));   // SegmentSeeker.swift:118-121,162-165
    }
    tr->timings.total_decoding_windows += 1;
    return WH_OK;
}

// Drop segments [n_keep, end) of a transcription together with their tokens, log-probs, words and texts (segments are appended in
// order, so everything that belongs to the dropped tail is a tail as well): the "replace the segments of the window" half of
// TranscribeTask.windowPostProcess (Core/TranscribeTask.swift:49-55) as far as a C hook can express it.
namespace whi {
void transcription_truncate_segments(wh_transcription* t, int n_keep) {
    if (!t || n_keep < 0 || n_keep >= (int)t->segments.size()) return;
    const wh_segment& g = t->segments[(size_t)n_keep];
    t->tokens.resize((size_t)g.token_offset);
    t->logprobs.resize((size_t)g.token_offset);
    if (g.word_offset >= 0 && g.word_offset <= (int)t->words.size()) {
        const size_t wt = g.word_offset < (int)t->words.size() ? (size_t)t->words[(size_t)g.word_offset].token_offset : t->word_tokens.size();
        t->word_tokens.resize(std::min(wt, t->word_tokens.size()));
        t->words.resize((size_t)g.word_offset);
        if (t->word_text.size() > (size_t)g.word_offset) t->word_text.resize((size_t)g.word_offset);
    }
    if (t->segment_text.size() > (size_t)n_keep) t->segment_text.resize((size_t)n_keep);
    t->segments.resize((size_t)n_keep);
}
}  // namespace whi

extern "C" int wh_transcription_set_segment_times(wh_transcription* t, int i, float start, float end) {
    if (!t || i < 0 || i >= (int)t->segments.size()) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_set_segment_times: segment %d out of range", i);
    t->segments[(size_t)i].start = start;
    t->segments[(size_t)i].end = end;
    return WH_OK;
}

// finalizeTranscriptionResult (TranscribeTask.swift:297-312): text = decode(all text tokens) trimmed, language code
extern "C" int wh_transcription_finalize(wh_transcription* tr, const wh_tokenizer* tok, const wh_decoding_options* opt, const wh_special_tokens* st) {
    if (!tr || !st) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_finalize: null argument");
    if (!tok) return WH_OK;
    std::vector<int> text_ids;
    for (int id : tr->tokens) if (id < st->special_token_begin) text_ids.push_back(id);
    tr->text = whi::trim_swift_whitespaces(tok->decode(text_ids));
    tr->language = language_code_of(tok, tr->language_token >= 0 ? tr->language_token : (opt ? opt->language_token : -1));
    tr->has_text = true;
    return WH_OK;
}

// ---- mergeTranscriptionResults ---------------------------------------------------------------------------------------------
extern "C" int wh_merge_transcriptions(const wh_transcription* const* results, int n, const char* const* confirmed_words, int n_confirmed,
                                       wh_transcription** out) {
    if (!out || n < 0 || (n && !results)) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_merge_transcriptions: null argument");
    auto m = new wh_transcription();
    m->has_text = true;
    if (confirmed_words) {
        for (int i = 0; i < n_confirmed; ++i) if (confirmed_words[i]) m->text += confirmed_words[i];
    } else {
        for (int i = 0; i < n; ++i) { if (i) m->text += " "; if (results[i]) m->text += results[i]->text; }
    }
    std::vector<const wh_transcription*> valid;
    for (int i = 0; i < n; ++i) if (results[i]) valid.push_back(results[i]);
    for (size_t ri = 0; ri < valid.size(); ++ri) {
        const wh_transcription* r = valid[ri];
        if (!r->has_text) m->has_text = false;
        if (r->words_enabled) m->words_enabled = true;
        for (size_t si = 0; si < r->segments.size(); ++si) {
            wh_segment g = r->segments[si];
            g.id = (int)(ri + si);                                  // "updatedSegment.id = resultIndex + segmentIndex" (:99)
            const int tsrc = g.token_offset, wsrc = g.word_offset;
            g.token_offset = (int)m->tokens.size();
            m->tokens.insert(m->tokens.end(), r->tokens.begin() + tsrc, r->tokens.begin() + tsrc + g.n_tokens);
            m->logprobs.insert(m->logprobs.end(), r->logprobs.begin() + tsrc, r->logprobs.begin() + tsrc + g.n_tokens);
            g.word_offset = (int)m->words.size();
            for (int k = 0; k < g.n_words; ++k) {
                wh_word_timing w = r->words[wsrc + k];
                const int wt = w.token_offset;
                w.token_offset = (int)m->word_tokens.size();
                m->word_tokens.insert(m->word_tokens.end(), r->word_tokens.begin() + wt, r->word_tokens.begin() + wt + w.n_tokens);
                m->words.push_back(w);
                m->word_text.push_back((size_t)(wsrc + k) < r->word_text.size() ? r->word_text[wsrc + k] : std::string());
            }
            m->segments.push_back(g);
            m->segment_text.push_back(si < r->segment_text.size() ? r->segment_text[si] : std::string());
        }
        for (int sk : r->seeks) m->seeks.push_back(sk);
    }
    m->language = valid.empty() ? std::string("en") : valid[0]->language;      // Constants.defaultLanguageCode
    m->language_token = valid.empty() ? -1 : valid[0]->language_token;
    // timings: loads = max, stage times and counters = sum, fullPipeline = min(wall span, sum) (:112-150)
    wh_timings& t = m->timings;
    double earliest_start = 0, earliest_token = 0, latest_end = 0, system = 0;
    for (size_t i = 0; i < valid.size(); ++i) {
        const wh_timings& a = valid[i]->timings;
        const double end = a.pipeline_start + a.full_pipeline;
        if (i == 0) { earliest_start = a.pipeline_start; earliest_token = a.first_token_time; latest_end = end; }
        earliest_start = std::min(earliest_start, a.pipeline_start);
        earliest_token = std::min(earliest_token, a.first_token_time);
        latest_end = std::max(latest_end, end);
        system += a.full_pipeline;
        t.model_loading = std::max(t.model_loading, a.model_loading);
        t.prewarm_load_time = std::max(t.prewarm_load_time, a.prewarm_load_time);
        t.encoder_load_time = std::max(t.encoder_load_time, a.encoder_load_time);
        t.decoder_load_time = std::max(t.decoder_load_time, a.decoder_load_time);
        t.tokenizer_load_time = std::max(t.tokenizer_load_time, a.tokenizer_load_time);
        t.audio_loading += a.audio_loading; t.audio_processing += a.audio_processing; t.logmels += a.logmels; t.encoding += a.encoding;
        t.decoding_init += a.decoding_init; t.decoding_loop += a.decoding_loop; t.decoding_predictions += a.decoding_predictions;
        t.decoding_filtering += a.decoding_filtering; t.decoding_sampling += a.decoding_sampling; t.decoding_fallback += a.decoding_fallback;
        t.decoding_windowing += a.decoding_windowing; t.decoding_kv_caching += a.decoding_kv_caching;
        t.decoding_word_timestamps += a.decoding_word_timestamps; t.decoding_non_prediction += a.decoding_non_prediction;
        t.total_audio_processing_runs += a.total_audio_processing_runs; t.total_logmel_runs += a.total_logmel_runs;
        t.total_encoding_runs += a.total_encoding_runs; t.total_decoding_loops += a.total_decoding_loops;
        t.total_kv_update_runs += a.total_kv_update_runs; t.total_timestamp_alignment_runs += a.total_timestamp_alignment_runs;
        t.total_decoding_fallbacks += a.total_decoding_fallbacks; t.total_decoding_windows += a.total_decoding_windows;
        t.input_audio_seconds += a.input_audio_seconds;
    }
    t.full_pipeline = std::min(latest_end - earliest_start, system);
    t.pipeline_start = earliest_start;
    t.first_token_time = earliest_token;
    *out = m;
    return WH_OK;
}

// ---- small text utilities -----------------------------------------------------------------------------------------------
// TextUtilities.compressionRatio(of: String) (Utilities/TextUtilities.swift:33-52): UTF-8 bytes / NSData.compressed(using: .zlib)
// (raw DEFLATE, level 5 - the same stream wh_compression_ratio produces for token arrays); empty text -> +inf
extern "C" float wh_compression_ratio_text(const char* utf8, int nbytes) {
    if (!utf8 || nbytes <= 0) return INFINITY;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, 5, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return INFINITY;
    std::vector<unsigned char> outb(deflateBound(&zs, (uLong)nbytes) + 16);
    zs.next_in = (Bytef*)utf8; zs.avail_in = (uInt)nbytes; zs.next_out = outb.data(); zs.avail_out = (uInt)outb.size();
    const int rc = deflate(&zs, Z_FINISH);
    const uLong clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen == 0) return INFINITY;
    return (float)nbytes / (float)clen;
}

// String.trimmingSpecialTokenCharacters (Constants.specialTokenCharacters "<|>", Core/Models.swift:1330): "<|en|>" -> "en"
extern "C" int wh_trimming_special_token_characters(const char* text, char* out, int capacity) {
    if (!text) { set_error(WH_ERR_INVALID_ARGUMENT, "null text"); return -1; }
    return copy_out(whi::trimming_special_token_characters(text), out, capacity);
}

// ---- writers ---------------------------------------------------------------------------------------------------------------
// ResultWriting.formatTime (ResultWriter.swift:14-26), Float arithmetic
extern "C" int wh_format_time(float seconds, int always_include_hours, char decimal_marker, char* out, int capacity) {
    const int hrs = (int)(seconds / 3600.0f);
    const int mins = (int)(fmodf(seconds, 3600.0f) / 60.0f);
    const int secs = (int)fmodf(seconds, 60.0f);
    const int msec = (int)((seconds - floorf(seconds)) * 1000.0f);
    char buf[64];
    if (always_include_hours || hrs > 0) snprintf(buf, sizeof buf, "%02d:%02d:%02d%c%03d", hrs, mins, secs, decimal_marker, msec);
    else snprintf(buf, sizeof buf, "%02d:%02d%c%03d", mins, secs, decimal_marker, msec);
    return copy_out(buf, out, capacity);
}

static std::string fmt_time(float s, bool hours, char marker) {
    char b[64];
    wh_format_time(s, hours, marker, b, sizeof b);
    return b;
}

static int write_file(const char* path, const std::string& content) {
    FILE* f = fopen(path, "wb");
    if (!f) return set_error(WH_ERR_TRANSCRIPTION_FAILED, "cannot open %s for writing", path);
    size_t n = fwrite(content.data(), 1, content.size(), f);
    fclose(f);
    return n == content.size() ? (int)WH_OK : set_error(WH_ERR_TRANSCRIPTION_FAILED, "short write to %s", path);
}

// WriteSRT / WriteVTT: one cue per word when a segment has word timings, else one per segment (ResultWriter.swift:70-134)
static int write_cues(const wh_transcription* t, const char* path, bool srt) {
    if (!t || !path) return set_error(WH_ERR_INVALID_ARGUMENT, "null argument");
    if (!t->has_text) return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "transcription has no text (no tokenizer attached)");
    std::string c = srt ? "" : "WEBVTT\n\n";
    int index = 1;
    auto cue = [&](float start, float end, const std::string& text) {
        if (srt) c += std::to_string(index++) + "\n" + fmt_time(start, true, ',') + " --> " + fmt_time(end, true, ',') + "\n" + text + "\n\n";
        else c += fmt_time(start, false, '.') + " --> " + fmt_time(end, false, '.') + "\n" + text + "\n\n";
    };
    for (size_t si = 0; si < t->segments.size(); ++si) {
        const wh_segment& g = t->segments[si];
        if (g.n_words > 0) for (int k = 0; k < g.n_words; ++k) cue(t->words[g.word_offset + k].start, t->words[g.word_offset + k].end, t->word_text[g.word_offset + k]);
        else cue(g.start, g.end, t->segment_text[si]);
    }
    return write_file(path, c);
}
extern "C" int wh_write_srt(const wh_transcription* t, const char* path) { return write_cues(t, path, true); }
extern "C" int wh_write_vtt(const wh_transcription* t, const char* path) { return write_cues(t, path, false); }

static void num(std::string& o, double v) {
    char b[40];
    if (std::isfinite(v)) snprintf(b, sizeof b, "%.9g", v); else snprintf(b, sizeof b, "null");
    o += b;
}
static void numd(std::string& o, double v) {
    char b[40];
    if (std::isfinite(v)) snprintf(b, sizeof b, "%.17g", v); else snprintf(b, sizeof b, "null");
    o += b;
}

// WriteJSON: the Codable encoding of TranscriptionResult (keys = property names; tokenLogProbs = [{"<id>": logprob}];
// words omitted when nil; seekTime null when nil).  Key order and float spelling are not fixed by JSONEncoder - compare as JSON.
static std::string transcription_json(const wh_transcription* t) {
    std::string o = "{\n  \"text\" : ";
    wh::json_escape(o, t->text);
    o += ",\n  \"language\" : ";
    wh::json_escape(o, t->language);
    o += ",\n  \"seekTime\" : ";
    if (t->has_seek_time) num(o, t->seek_time); else o += "null";
    o += ",\n  \"segments\" : [";
    for (size_t si = 0; si < t->segments.size(); ++si) {
        const wh_segment& g = t->segments[si];
        o += si ? ",\n    {" : "\n    {";
        o += "\"id\" : " + std::to_string(g.id) + ", \"seek\" : " + std::to_string(g.seek) + ", \"start\" : "; num(o, g.start);
        o += ", \"end\" : "; num(o, g.end);
        o += ", \"text\" : "; wh::json_escape(o, t->segment_text[si]);
        o += ", \"tokens\" : [";
        for (int k = 0; k < g.n_tokens; ++k) { if (k) o += ", "; o += std::to_string(t->tokens[g.token_offset + k]); }
        o += "], \"tokenLogProbs\" : [";
        for (int k = 0; k < g.n_tokens; ++k) { if (k) o += ", "; o += "{\"" + std::to_string(t->tokens[g.token_offset + k]) + "\" : "; num(o, t->logprobs[g.token_offset + k]); o += "}"; }
        o += "], \"temperature\" : "; num(o, g.temperature);
        o += ", \"avgLogprob\" : "; num(o, g.avg_logprob);
        o += ", \"compressionRatio\" : "; num(o, g.compression_ratio);
        o += ", \"noSpeechProb\" : "; num(o, g.no_speech_prob);
        if (g.n_words > 0 || t->words_enabled) {   // `words` is nil unless addWordTimestamps ran, then possibly []
            o += ", \"words\" : [";
            for (int k = 0; k < g.n_words; ++k) {
                const wh_word_timing& w = t->words[g.word_offset + k];
                o += k ? ", {" : "{";
                o += "\"word\" : "; wh::json_escape(o, t->word_text[g.word_offset + k]);
                o += ", \"tokens\" : [";
                for (int q = 0; q < w.n_tokens; ++q) { if (q) o += ", "; o += std::to_string(t->word_tokens[w.token_offset + q]); }
                o += "], \"start\" : "; num(o, w.start);
                o += ", \"end\" : "; num(o, w.end);
                o += ", \"probability\" : "; num(o, w.probability);
                o += "}";
            }
            o += "]";
        }
        o += "}";
    }
    o += t->segments.empty() ? "]" : "\n  ]";
    const wh_timings& m = t->timings;
    const std::pair<const char*, double> tv[] = {
        {"pipelineStart", m.pipeline_start}, {"firstTokenTime", m.first_token_time}, {"inputAudioSeconds", m.input_audio_seconds},
        {"modelLoading", m.model_loading}, {"prewarmLoadTime", m.prewarm_load_time}, {"encoderLoadTime", m.encoder_load_time},
        {"decoderLoadTime", m.decoder_load_time}, {"encoderSpecializationTime", m.encoder_specialization_time},
        {"decoderSpecializationTime", m.decoder_specialization_time}, {"tokenizerLoadTime", m.tokenizer_load_time},
        {"audioLoading", m.audio_loading}, {"audioProcessing", m.audio_processing}, {"logmels", m.logmels}, {"encoding", m.encoding},
        {"decodingInit", m.decoding_init}, {"decodingLoop", m.decoding_loop}, {"decodingPredictions", m.decoding_predictions},
        {"decodingFiltering", m.decoding_filtering}, {"decodingSampling", m.decoding_sampling}, {"decodingFallback", m.decoding_fallback},
        {"decodingWindowing", m.decoding_windowing}, {"decodingKvCaching", m.decoding_kv_caching},
        {"decodingWordTimestamps", m.decoding_word_timestamps}, {"decodingNonPrediction", m.decoding_non_prediction},
        {"totalAudioProcessingRuns", m.total_audio_processing_runs}, {"totalLogmelRuns", m.total_logmel_runs},
        {"totalEncodingRuns", m.total_encoding_runs}, {"totalDecodingLoops", m.total_decoding_loops},
        {"totalKVUpdateRuns", m.total_kv_update_runs}, {"totalTimestampAlignmentRuns", m.total_timestamp_alignment_runs},
        {"totalDecodingFallbacks", m.total_decoding_fallbacks}, {"totalDecodingWindows", m.total_decoding_windows},
        {"fullPipeline", m.full_pipeline}};
    o += ",\n  \"timings\" : {";
    bool first = true;
    for (auto& kv : tv) { o += first ? "\n    \"" : ",\n    \""; first = false; o += kv.first; o += "\" : "; numd(o, kv.second); }
    o += "\n  }\n}\n";
    return o;
}

extern "C" int wh_write_json(const wh_transcription* t, const char* path) {
    if (!t || !path) return set_error(WH_ERR_INVALID_ARGUMENT, "null argument");
    if (!t->has_text) return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "transcription has no text (no tokenizer attached)");
    return write_file(path, transcription_json(t));
}

// The same document as a string (the wire format of the multi-GPU result gather); works without text too (empty strings).
extern "C" int wh_transcription_to_json(const wh_transcription* t, char* out, int capacity) {
    if (!t) { set_error(WH_ERR_INVALID_ARGUMENT, "null transcription"); return -1; }
    wh_transcription padded;
    const wh_transcription* src = t;
    if (t->segment_text.size() < t->segments.size() || t->word_text.size() < t->words.size()) {
        padded = *t;
        padded.segment_text.resize(t->segments.size());
        padded.word_text.resize(t->words.size());
        src = &padded;
    }
    return copy_out(transcription_json(src), out, capacity);
}

// JSONDecoder on TranscriptionResult: rebuilds the flat container from the Codable document (unknown keys ignored, missing keys default)
extern "C" int wh_transcription_from_json(const char* json, int nbytes, wh_transcription** out) {
    if (!json || nbytes < 0 || !out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_transcription_from_json: null argument");
    wh::JsonValue root;
    std::string err;
    if (!wh::JsonParser(json, (size_t)nbytes).parse(root, err) || root.kind != wh::JsonValue::Object)
        return set_error(WH_ERR_TRANSCRIPTION_FAILED, "wh_transcription_from_json: %s", err.empty() ? "not a JSON object" : err.c_str());
    auto numv = [](const wh::JsonValue* v, double d) { return v && v->kind == wh::JsonValue::Number ? v->num : d; };
    auto strv = [](const wh::JsonValue* v) { return v && v->kind == wh::JsonValue::String ? v->str : std::string(); };
    auto tr = new wh_transcription();
    tr->has_text = true;
    tr->text = strv(root.get("text"));
    tr->language = strv(root.get("language"));
    if (const wh::JsonValue* sk = root.get("seekTime")) if (sk->kind == wh::JsonValue::Number) { tr->seek_time = (float)sk->num; tr->has_seek_time = true; }
    if (const wh::JsonValue* segs = root.get("segments")) {
        for (auto& g : segs->arr) {
            if (g.kind != wh::JsonValue::Object) { delete tr; return set_error(WH_ERR_TRANSCRIPTION_FAILED, "wh_transcription_from_json: segment is not an object"); }
            wh_segment c{};
            c.id = (int)numv(g.get("id"), 0); c.seek = (int)numv(g.get("seek"), 0);
            c.start = (float)numv(g.get("start"), 0); c.end = (float)numv(g.get("end"), 0);
            c.temperature = (float)numv(g.get("temperature"), 1.0); c.avg_logprob = (float)numv(g.get("avgLogprob"), 0);
            c.compression_ratio = (float)numv(g.get("compressionRatio"), 1.0); c.no_speech_prob = (float)numv(g.get("noSpeechProb"), 0);
            c.token_offset = (int)tr->tokens.size();
            const wh::JsonValue* toks = g.get("tokens");
            const wh::JsonValue* lps = g.get("tokenLogProbs");
            if (toks) for (size_t k = 0; k < toks->arr.size(); ++k) {
                tr->tokens.push_back((int)numv(&toks->arr[k], 0));
                float lp = 0;
                if (lps && k < lps->arr.size() && lps->arr[k].kind == wh::JsonValue::Object && !lps->arr[k].obj.empty()) lp = (float)numv(&lps->arr[k].obj[0].second, 0);
                tr->logprobs.push_back(lp);
            }
            c.n_tokens = (int)tr->tokens.size() - c.token_offset;
            c.word_offset = (int)tr->words.size();
            if (const wh::JsonValue* ws = g.get("words")) if (ws->kind == wh::JsonValue::Array) tr->words_enabled = true;
            if (const wh::JsonValue* ws = g.get("words")) for (auto& w : ws->arr) {
                wh_word_timing wt{};
                wt.token_offset = (int)tr->word_tokens.size();
                if (const wh::JsonValue* wtok = w.get("tokens")) for (auto& x : wtok->arr) tr->word_tokens.push_back((int)numv(&x, 0));
                wt.n_tokens = (int)tr->word_tokens.size() - wt.token_offset;
                wt.start = (float)numv(w.get("start"), 0); wt.end = (float)numv(w.get("end"), 0); wt.probability = (float)numv(w.get("probability"), 0);
                tr->words.push_back(wt);
                tr->word_text.push_back(strv(w.get("word")));
            }
            c.n_words = (int)tr->words.size() - c.word_offset;
            tr->segments.push_back(c);
            tr->segment_text.push_back(strv(g.get("text")));
        }
    }
    if (const wh::JsonValue* tm = root.get("timings")) {
        wh_timings& m = tr->timings;
        const std::pair<const char*, double*> tv[] = {
            {"pipelineStart", &m.pipeline_start}, {"firstTokenTime", &m.first_token_time}, {"inputAudioSeconds", &m.input_audio_seconds},
            {"modelLoading", &m.model_loading}, {"prewarmLoadTime", &m.prewarm_load_time}, {"encoderLoadTime", &m.encoder_load_time},
            {"decoderLoadTime", &m.decoder_load_time}, {"encoderSpecializationTime", &m.encoder_specialization_time},
            {"decoderSpecializationTime", &m.decoder_specialization_time}, {"tokenizerLoadTime", &m.tokenizer_load_time},
            {"audioLoading", &m.audio_loading}, {"audioProcessing", &m.audio_processing}, {"logmels", &m.logmels}, {"encoding", &m.encoding},
            {"decodingInit", &m.decoding_init}, {"decodingLoop", &m.decoding_loop}, {"decodingPredictions", &m.decoding_predictions},
            {"decodingFiltering", &m.decoding_filtering}, {"decodingSampling", &m.decoding_sampling}, {"decodingFallback", &m.decoding_fallback},
            {"decodingWindowing", &m.decoding_windowing}, {"decodingKvCaching", &m.decoding_kv_caching},
            {"decodingWordTimestamps", &m.decoding_word_timestamps}, {"decodingNonPrediction", &m.decoding_non_prediction},
            {"totalAudioProcessingRuns", &m.total_audio_processing_runs}, {"totalLogmelRuns", &m.total_logmel_runs},
            {"totalEncodingRuns", &m.total_encoding_runs}, {"totalDecodingLoops", &m.total_decoding_loops},
            {"totalKVUpdateRuns", &m.total_kv_update_runs}, {"totalTimestampAlignmentRuns", &m.total_timestamp_alignment_runs},
            {"totalDecodingFallbacks", &m.total_decoding_fallbacks}, {"totalDecodingWindows", &m.total_decoding_windows},
            {"fullPipeline", &m.full_pipeline}};
        for (auto& kv : tv) *kv.second = numv(tm->get(kv.first), 0.0);
    }
    *out = tr;
    return WH_OK;
}

// TranscriptionUtilities.updateSegmentTimings for every segment + result.seekTime (AudioChunking.updateSeekOffsetsForResults,
// Core/Audio/AudioChunker.swift:14-39): shifts a chunk's result to the time base of the full audio, Float arithmetic.
extern "C" int wh_transcription_apply_seek_offset(wh_transcription* t, int seek_offset_samples) {
    if (!t) return set_error(WH_ERR_INVALID_ARGUMENT, "null transcription");
    const float seekTime = (float)seek_offset_samples / (float)WH_SAMPLE_RATE;
    const int seekIdx = (int)(seekTime * (float)WH_SAMPLE_RATE);
    for (auto& g : t->segments) { g.seek += seekIdx; g.start += seekTime; g.end += seekTime; }
    for (auto& w : t->words) { w.start += seekTime; w.end += seekTime; }
    t->seek_time = seekTime;
    t->has_seek_time = true;
    return WH_OK;
}

// ---- audio ingest ----------------------------------------------------------------------------------------------------------
// AudioProcessor.convertToMono (AudioProcessor.swift:525-625) on planar float channels.  mode 0 = .specificChannel(indices[0]),
// mode 1 = .sumChannels(indices or all): sum, then rescale so the mix keeps the loudest input channel's peak.
extern "C" int wh_convert_to_mono(const float* const* channels, int n_channels, int n_frames, int mode, const int32_t* indices,
                                  int n_indices, float* out) {
    if (!channels || !out || n_channels < 1 || n_frames < 0) return set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_convert_to_mono: invalid argument");
    if (n_channels == 1) { memcpy(out, channels[0], sizeof(float) * n_frames); return WH_OK; }
    if (mode == 0) {
        int c = (indices && n_indices > 0) ? indices[0] : 0;
        if (c < 0 || c >= n_channels) c = 0;
        memcpy(out, channels[c], sizeof(float) * n_frames);
        return WH_OK;
    }
    std::vector<int> sel;
    if (indices && n_indices > 0) {
        for (int i = 0; i < n_indices; ++i) if (indices[i] >= 0 && indices[i] < n_channels) sel.push_back(indices[i]);
        if (sel.empty()) { memcpy(out, channels[0], sizeof(float) * n_frames); return WH_OK; }
    } else for (int c = 0; c < n_channels; ++c) sel.push_back(c);
    float max_peak = 0;
    for (int c : sel) { float p = 0; for (int i = 0; i < n_frames; ++i) p = std::max(p, fabsf(channels[c][i])); max_peak = std::max(max_peak, p); }
    for (int i = 0; i < n_frames; ++i) out[i] = 0;
    for (int c : sel) for (int i = 0; i < n_frames; ++i) out[i] += channels[c][i];
    float mono_peak = 0;
    for (int i = 0; i < n_frames; ++i) mono_peak = std::max(mono_peak, fabsf(out[i]));
    const float scale = max_peak / std::max(mono_peak, 0.0001f);
    for (int i = 0; i < n_frames; ++i) out[i] *= scale;
    return WH_OK;
}

// Band-limited resampling to `out_rate` (Kaiser-windowed sinc, 32 zero crossings, beta 9).  The reference delegates to
// AVAudioConverter, whose filter is not published: equal rates are passed through untouched (the one case the reference's tests pin,
// UnitTests.swift:409-461), everything else is this filter - same length rule, not sample-identical to Apple's.
extern "C" int wh_resample(const float* in, int n_in, double in_rate, double out_rate, float* out, int capacity) {
    if (!in || n_in < 0 || in_rate <= 0 || out_rate <= 0) { set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_resample: invalid argument"); return -1; }
    const long long n_out = (long long)((double)n_in / in_rate * out_rate);   // AVAudioFrameCount(inputDuration * sampleRate)
    if (n_out > 0x7fffffffLL) { set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_resample: output too long"); return -1; }
    if (!out) return (int)n_out;
    if (n_out > capacity) { set_error(WH_ERR_AUDIO_PROCESSING_FAILED, "wh_resample: %lld frames do not fit", n_out); return -1; }
    if (in_rate == out_rate) { memcpy(out, in, sizeof(float) * (size_t)n_out); return (int)n_out; }
    const double ratio = out_rate / in_rate, fc = std::min(1.0, ratio) * 0.97;   // cutoff relative to the input Nyquist
    const int zeros = 32, phases = 256;
    const double half = zeros / fc, beta = 9.0;
    auto bessel0 = [](double x) { double s = 1, t = 1; for (int k = 1; k < 60; ++k) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; if (t < 1e-14 * s) break; } return s; };
    const double ib = 1.0 / bessel0(beta);
    // h(x) = fc sinc(fc x) kaiser(x / half), tabulated at 1/phases of an input sample and interpolated linearly
    const int tn = (int)ceil(half * phases) + 2;
    std::vector<double> h((size_t)tn + 1, 0.0);
    for (int k = 0; k < tn; ++k) {
        const double x = (double)k / phases, u = x / half;
        if (u >= 1.0) break;
        const double a = M_PI * fc * x;
        h[k] = (fabs(a) < 1e-9 ? 1.0 : sin(a) / a) * fc * bessel0(beta * sqrt(1 - u * u)) * ib;
    }
    for (long long o = 0; o < n_out; ++o) {
        const double center = (double)o / ratio;
        const long long lo = std::max(0LL, (long long)ceil(center - half)), hi = std::min((long long)n_in - 1, (long long)floor(center + half));
        double acc = 0;
        for (long long i = lo; i <= hi; ++i) {
            const double t = fabs((double)i - center) * phases;
            const int k = (int)t;
            const double f = t - k;
            acc += (double)in[i] * (h[k] + (h[k + 1] - h[k]) * f);
        }
        out[o] = (float)acc;
    }
    return (int)n_out;
}

namespace {
struct Wav {
    int format = 0, channels = 0, bits = 0, block = 0;
    double rate = 0;
    const unsigned char* data = nullptr;
    size_t data_bytes = 0;
};
uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool parse_wav(const std::string& f, Wav& w, std::string& err) {
    const unsigned char* p = (const unsigned char*)f.data();
    if (f.size() < 12 || memcmp(p, "RIFF", 4) || memcmp(p + 8, "WAVE", 4)) { err = "not a RIFF/WAVE file"; return false; }
    size_t off = 12;
    bool have_fmt = false;
    while (off + 8 <= f.size()) {
        const uint32_t sz = rd32(p + off + 4);
        const unsigned char* body = p + off + 8;
        const size_t avail = f.size() - off - 8;
        if (!memcmp(p + off, "fmt ", 4) && sz >= 16 && avail >= 16) {
            w.format = rd16(body); w.channels = rd16(body + 2); w.rate = rd32(body + 4); w.block = rd16(body + 12); w.bits = rd16(body + 14);
            if (w.format == 0xFFFE && sz >= 40 && avail >= 40) w.format = rd16(body + 24);   // WAVE_FORMAT_EXTENSIBLE sub-format
            have_fmt = true;
        } else if (!memcmp(p + off, "data", 4)) {
            w.data = body; w.data_bytes = std::min((size_t)sz, avail);
            break;
        }
        off += 8 + (size_t)sz + (sz & 1);
    }
    if (!have_fmt || !w.data) { err = "missing fmt or data chunk"; return false; }
    const bool pcm = w.format == 1 && (w.bits == 8 || w.bits == 16 || w.bits == 24 || w.bits == 32);
    const bool flt = w.format == 3 && (w.bits == 32 || w.bits == 64);
    if ((!pcm && !flt) || w.channels < 1 || w.rate <= 0) { err = "unsupported WAV encoding (PCM 8/16/24/32-bit or IEEE float only)"; return false; }
    if (w.block < w.channels * w.bits / 8) w.block = w.channels * w.bits / 8;
    return true;
}

float sample_at(const Wav& w, const unsigned char* p) {   // AVAudioFile .pcmFormatFloat32 conversion: integers scaled by 2^-(bits-1)
    if (w.format == 3) { if (w.bits == 32) { float v; memcpy(&v, p, 4); return v; } double d; memcpy(&d, p, 8); return (float)d; }
    switch (w.bits) {
        case 8: return ((int)p[0] - 128) / 128.0f;
        case 16: return (float)(int16_t)rd16(p) / 32768.0f;
        case 24: { int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8; return (float)v / 8388608.0f; }
        default: return (float)((double)(int32_t)rd32(p) / 2147483648.0);
    }
}
}  // namespace

// AudioProcessor.loadAudio(fromPath:channelMode:startTime:endTime:maxReadFrameSize:) (AudioProcessor.swift:229-300) for RIFF/WAVE
// input: 16 kHz mono is returned as read; anything else is read in chunks of max_read_frame_size frames (0 = 1 323 000,
// Constants.defaultAudioReadFrameSize), each chunk mixed to mono (convertToMono - the peak renormalisation is per chunk, as in the
// reference) and resampled to 16 kHz.  end_time = NAN means nil.  The caller frees *pcm_out with wh_audio_free.
extern "C" int wh_load_audio(const char* path, int channel_mode, const int32_t* channel_indices, int n_channel_indices, double start_time,
                             double end_time, int max_read_frame_size, float** pcm_out, int* n_out) {
    if (!path || !pcm_out || !n_out) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "wh_load_audio: null argument");
    *pcm_out = nullptr; *n_out = 0;
    std::string file, err;
    if (!wh::read_file(path, file)) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "Resource path does not exist %s", path);
    Wav w;
    if (!parse_wav(file, w, err)) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "%s: %s", path, err.c_str());
    const long long length = (long long)(w.data_bytes / (size_t)w.block);
    if (std::isnan(start_time) || start_time < 0) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "start time must be a non-negative number");
    const long long start = (long long)(start_time * w.rate);
    const long long end = std::isnan(end_time) ? length : std::min((long long)(end_time * w.rate), length);
    if (start > end) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "start time %.3f s is outside the file", start_time);
    if (end - start > 0x7fffffffLL) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "audio too long for one buffer");
    const long long frames = end - start;
    const int bps = w.bits / 8;
    std::vector<float> mono;
    if (w.rate == 16000.0 && w.channels == 1) {
        mono.resize((size_t)frames);
        for (long long i = 0; i < frames; ++i) mono[(size_t)i] = sample_at(w, w.data + (size_t)(start + i) * w.block);
    } else {
        const long long chunk = max_read_frame_size > 0 ? max_read_frame_size : 1323000;
        std::vector<std::vector<float>> planes(w.channels);
        std::vector<const float*> ptrs(w.channels);
        std::vector<float> mixed, res;
        for (long long pos = 0; pos < frames; pos += chunk) {
            const int n = (int)std::min(chunk, frames - pos);
            for (int c = 0; c < w.channels; ++c) {
                planes[c].resize(n);
                for (int i = 0; i < n; ++i) planes[c][i] = sample_at(w, w.data + (size_t)(start + pos + i) * w.block + (size_t)c * bps);
                ptrs[c] = planes[c].data();
            }
            mixed.resize(n);
            int r = wh_convert_to_mono(ptrs.data(), w.channels, n, channel_mode, channel_indices, n_channel_indices, mixed.data());
            if (r) return r;
            const int no = wh_resample(mixed.data(), n, w.rate, 16000.0, nullptr, 0);
            res.resize((size_t)std::max(no, 0));
            if (no > 0 && wh_resample(mixed.data(), n, w.rate, 16000.0, res.data(), no) < 0) return WH_ERR_AUDIO_PROCESSING_FAILED;
            mono.insert(mono.end(), res.begin(), res.end());
        }
    }
    float* buf = (float*)malloc(sizeof(float) * std::max<size_t>(mono.size(), 1));
    if (!buf) return set_error(WH_ERR_LOAD_AUDIO_FAILED, "Unable to create audio buffer");
    if (!mono.empty()) memcpy(buf, mono.data(), sizeof(float) * mono.size());
    *pcm_out = buf;
    *n_out = (int)mono.size();
    return WH_OK;
}

extern "C" void wh_audio_free(float* pcm) { free(pcm); }

// Word timestamps on the host (no GPU): SegmentSeeker.addWordTimestamps and its helpers,
// WhisperKit/Core/Text/SegmentSeeker.swift:280-659.  Input is the alignment matrix the decoder kernels wrote
// (DecodingInputs.alignmentWeights rows of the window's tokens); arithmetic is Float like the reference.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "text.h"
#include "unicode_tables.h"

using whi::set_error;
using whi::Word;

namespace whi {

const char* const kDefaultPrependPunctuations = "\"'“¡¿([{-";          // Constants.defaultPrependPunctuations, Core/Models.swift:1459
const char* const kDefaultAppendPunctuations = "\"'.。,，!！?？:：”)]}、";  // Constants.defaultAppendPunctuations

// Swift String.contains(other) as the reference's tests exercise it: false for the empty string
static bool contains(const std::string& set, const std::string& s) { return !s.empty() && set.find(s) != std::string::npos; }

// mergePunctuations, SegmentSeeker.swift:280-338
std::vector<Word> merge_punctuations(const std::vector<Word>& alignment, const std::string& prepended, const std::string& appended) {
    if (alignment.empty()) return {};
    std::vector<Word> pre, app;
    if (!contains(prepended, trim_swift_whitespaces(alignment[0].word))) pre.push_back(alignment[0]);
    for (size_t i = 1; i < alignment.size(); ++i) {
        Word cur = alignment[i];
        const Word& prev = alignment[i - 1];
        auto sc = utf8_scalars(prev.word.substr(0, 4));
        if (!sc.empty() && wh::is_swift_whitespace(sc[0]) && contains(prepended, trim_swift_whitespaces(prev.word))) {
            cur.word = prev.word + cur.word;
            std::vector<int> t = prev.tokens;
            t.insert(t.end(), cur.tokens.begin(), cur.tokens.end());
            cur.tokens = std::move(t);
            if (pre.empty()) pre.push_back(cur); else pre.back() = cur;
        } else pre.push_back(cur);
    }
    if (!pre.empty()) app.push_back(pre[0]);
    for (size_t i = 1; i < pre.size(); ++i) {
        const Word& cur = pre[i];
        Word prev = pre[i - 1];
        const bool ends_with_space = !prev.word.empty() && prev.word.back() == ' ';
        if (!ends_with_space && contains(appended, trim_swift_whitespaces(cur.word))) {
            prev.word += cur.word;
            prev.tokens.insert(prev.tokens.end(), cur.tokens.begin(), cur.tokens.end());
            app.back() = prev;
        } else app.push_back(cur);
    }
    std::vector<Word> out;
    for (auto& w : app) if (!w.word.empty() && !contains(appended, w.word) && !contains(prepended, w.word)) out.push_back(w);
    return out;
}

// findAlignment, SegmentSeeker.swift:340-408
static int find_alignment(const wh_tokenizer* tok, const char* language, const std::vector<int>& word_token_ids, const float* matrix, int rows,
                          const std::vector<float>& logprobs, std::vector<Word>& out) {
    const int cols = WH_AUDIO_CTX;
    const int cap = rows + cols + 8;
    std::vector<int32_t> ti(cap), tj(cap);
    int len = wh_dynamic_time_warping(matrix, rows, cols, ti.data(), tj.data(), cap);
    if (len < 0) return set_error(WH_ERR_SEGMENTING_FAILED, "dynamicTimeWarping failed on a %d x %d matrix", rows, cols);
    std::vector<std::string> words;
    std::vector<std::vector<int>> word_tokens;
    tok->split_to_word_tokens(word_token_ids, language, words, word_tokens);
    out.clear();
    if (word_tokens.size() <= 1) return WH_OK;
    const float spt = 0.02f;   // WhisperKit.secondsPerTimeToken
    std::vector<float> start_times{0.0f}, end_times;
    int cur = len > 0 ? ti[0] : 0;
    for (int k = 0; k < len; ++k)
        if (ti[k] != cur) { cur = ti[k]; float t = (float)tj[k] * spt; start_times.push_back(t); end_times.push_back(t); }
    end_times.push_back((float)(len > 0 ? tj[len - 1] : 1500) * spt);
    size_t index = 0;
    for (size_t w = 0; w < word_tokens.size(); ++w) {
        const size_t start_index = index;
        if (index >= start_times.size()) return set_error(WH_ERR_SEGMENTING_FAILED, "alignment path is shorter than the token list");
        const float ws = start_times[index];
        index += word_tokens[w].size() - 1;
        if (index >= end_times.size()) return set_error(WH_ERR_SEGMENTING_FAILED, "alignment path is shorter than the token list");
        const float we = end_times[index];
        index += 1;
        if (index > logprobs.size()) return set_error(WH_ERR_SEGMENTING_FAILED, "fewer log-probs than word tokens");
        float sum = 0;
        for (size_t k = start_index; k < index; ++k) sum += logprobs[k];
        Word wt;
        wt.word = words[w]; wt.tokens = word_tokens[w]; wt.start = ws; wt.end = we;
        wt.probability = expf(sum / (float)(index - start_index));
        out.push_back(std::move(wt));
    }
    return WH_OK;
}

// calculateWordDurationConstraints, SegmentSeeker.swift:498-507
static void word_duration_constraints(const std::vector<Word>& alignment, float* median, float* max_duration) {
    std::vector<float> d;
    for (auto& w : alignment) if (w.duration() > 0) d.push_back(w.duration());
    std::sort(d.begin(), d.end());
    float med = d.empty() ? 0.0f : d[d.size() / 2];
    *median = std::min(0.7f, med);
    *max_duration = *median * 2;
}

// truncateLongWordsAtSentenceBoundaries, SegmentSeeker.swift:509-526
static void truncate_long_words(std::vector<Word>& a, float max_duration) {
    static const char* const marks[] = {".", "。", "!", "！", "?", "？"};
    auto is_mark = [](const std::string& w) { for (auto m : marks) if (w == m) return true; return false; };
    for (size_t i = 1; i < a.size(); ++i)
        if (a[i].duration() > max_duration) {
            if (is_mark(a[i].word)) a[i].end = a[i].start + max_duration;
            else if (is_mark(a[i - 1].word)) a[i].start = a[i].end - max_duration;
        }
}

struct SegWords { std::vector<Word> words; };

// updateSegmentsWithWordTimings, SegmentSeeker.swift:528-659
static void update_segments_with_word_timings(const wh_tokenizer* tok, int special_begin, std::vector<wh_segment>& segments,
                                              const int32_t* tokens, const std::vector<Word>& merged, int seek, float last_speech,
                                              float cmd, float max_duration, std::vector<SegWords>& out) {
    const float time_offset = (float)seek / (float)WH_SAMPLE_RATE;
    size_t word_index = 0;
    out.assign(segments.size(), SegWords());
    for (size_t si = 0; si < segments.size(); ++si) {
        const wh_segment original = segments[si];
        wh_segment& seg = segments[si];
        int text_tokens = 0, saved = 0;
        for (int k = 0; k < seg.n_tokens; ++k) if (tokens[seg.token_offset + k] < special_begin) ++text_tokens;
        std::vector<Word>& words = out[si].words;
        while (word_index < merged.size() && saved < text_tokens) {
            const Word& timing = merged[word_index++];
            std::vector<int> tt;
            for (int t : timing.tokens) if (t < special_begin) tt.push_back(t);
            if (tt.empty()) continue;
            std::string word = tt.size() < timing.tokens.size() ? tok->decode(tt) : timing.word;
            float start = rounded2(time_offset + timing.start);
            const float end = rounded2(time_offset + timing.end);
            if (end - start < cmd / 4) {
                if (!words.empty()) {
                    const float prev_end = words.back().end;
                    if (start > prev_end) start = rounded2(start - std::min(start - prev_end, cmd / 2));
                } else if (si > 0 && start > segments[si - 1].end) {
                    start = rounded2(start - std::min(start - segments[si - 1].end, cmd / 2));
                }
            }
            Word w;
            w.word = std::move(word); w.tokens = std::move(tt); w.start = start; w.end = end; w.probability = rounded2(timing.probability);
            saved += (int)w.tokens.size();
            words.push_back(std::move(w));
        }
        if (!words.empty()) {
            const Word first = words[0];
            const float pause = first.end - last_speech;
            const bool first_too_long = first.duration() > max_duration;
            const bool both_too_long = words.size() > 1 && words[1].end - first.start > max_duration * 2;
            if (pause > cmd * 4 && (first_too_long || both_too_long)) {
                if (words.size() > 1 && words[1].duration() > max_duration) {
                    const float boundary = std::max(words[1].end / 2, words[1].end - max_duration);
                    words[0].end = boundary;
                    words[1].start = boundary;
                }
                words[0].start = std::max(last_speech, words[0].end - max_duration);
            }
            if (original.start < words[0].end && original.start - 0.5f > words[0].start)
                words[0].start = std::max(0.0f, std::min(words[0].end - cmd, original.start));
            else
                seg.start = words[0].start;
            const Word last = words.back();
            if (seg.end > last.start && original.end + 0.5f < last.end)
                words.back().end = std::max(last.start + cmd, original.end);
            else
                seg.end = last.end;
            last_speech = seg.end;
        }
    }
}

// addWordTimestamps, SegmentSeeker.swift:410-496, for one window.  `segments` index `tokens` / `logprobs`; row r of `alignment`
// belongs to the r-th token of the segments in order.  Appends words (+ texts) to `tr` and rewrites segment start / end.
int add_word_timestamps(const wh_tokenizer* tok, const char* language, int special_begin, wh_segment* segments, int n_segments,
                        const int32_t* tokens, const float* logprobs, const float* alignment, int alignment_rows, int seek,
                        float last_speech_timestamp, wh_transcription* tr) {
    std::vector<int> ids;
    std::vector<float> lps;
    for (int s = 0; s < n_segments; ++s)
        for (int k = 0; k < segments[s].n_tokens; ++k) { ids.push_back(tokens[segments[s].token_offset + k]); lps.push_back(logprobs[segments[s].token_offset + k]); }
    std::vector<float> padded;          // a result may carry one token more than the 224 alignment rows (the appended EOT): zero rows
    if ((int)ids.size() > alignment_rows) {
        padded.assign(ids.size() * (size_t)WH_AUDIO_CTX, 0.0f);
        memcpy(padded.data(), alignment, sizeof(float) * (size_t)alignment_rows * WH_AUDIO_CTX);
        alignment = padded.data();
    }
    std::vector<Word> alignment_words;
    if (!ids.empty()) {
        int r = find_alignment(tok, language, ids, alignment, (int)ids.size(), lps, alignment_words);
        if (r) return r;
    }
    float median = 0, max_duration = 0;
    word_duration_constraints(alignment_words, &median, &max_duration);
    truncate_long_words(alignment_words, max_duration);
    if (!alignment_words.empty()) alignment_words = merge_punctuations(alignment_words, kDefaultPrependPunctuations, kDefaultAppendPunctuations);
    std::vector<wh_segment> segs(segments, segments + n_segments);
    std::vector<SegWords> per_segment;
    update_segments_with_word_timings(tok, special_begin, segs, tokens, alignment_words, seek, last_speech_timestamp, median, max_duration, per_segment);
    for (int s = 0; s < n_segments; ++s) {
        segs[s].word_offset = (int)tr->words.size();
        for (auto& w : per_segment[s].words) {
            wh_word_timing wt{};
            wt.token_offset = (int)tr->word_tokens.size();
            wt.n_tokens = (int)w.tokens.size();
            wt.start = w.start; wt.end = w.end; wt.probability = w.probability;
            tr->word_tokens.insert(tr->word_tokens.end(), w.tokens.begin(), w.tokens.end());
            tr->words.push_back(wt);
            tr->word_text.push_back(w.word);
        }
        segs[s].n_words = (int)tr->words.size() - segs[s].word_offset;
        segments[s] = segs[s];
    }
    return WH_OK;
}

}  // namespace whi

// C ABI: one window's addWordTimestamps as a pure host function (parity-testable without a GPU)
extern "C" int wh_add_word_timestamps(const wh_tokenizer* tok, const char* language_code, const wh_special_tokens* st,
                                      const wh_segment* segments, int n_segments, const int32_t* tokens, const float* logprobs,
                                      int n_tokens, const float* alignment, int alignment_rows, int seek, float last_speech_timestamp,
                                      int skip_special_tokens, wh_transcription** out) {
    if (!tok || !st || !out || n_segments < 0 || (n_segments && !segments) || !tokens || !logprobs || !alignment)
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_add_word_timestamps: null argument");
    for (int s = 0; s < n_segments; ++s)
        if (segments[s].token_offset < 0 || segments[s].n_tokens < 0 || segments[s].token_offset + segments[s].n_tokens > n_tokens)
            return set_error(WH_ERR_INVALID_ARGUMENT, "wh_add_word_timestamps: segment %d indexes outside the token array", s);
    auto tr = new wh_transcription();
    tr->tokens.assign(tokens, tokens + n_tokens);
    tr->logprobs.assign(logprobs, logprobs + n_tokens);
    tr->segments.assign(segments, segments + n_segments);
    int r = whi::add_word_timestamps(tok, language_code, st->special_token_begin, tr->segments.data(), n_segments, tr->tokens.data(),
                                     tr->logprobs.data(), alignment, alignment_rows, seek, last_speech_timestamp, tr);
    if (r) { delete tr; return r; }
    tr->has_text = true;
    tr->words_enabled = true;
    for (auto& g : tr->segments) {
        std::vector<int> t;
        for (int k = 0; k < g.n_tokens; ++k) {
            int id = tr->tokens[g.token_offset + k];
            if (!skip_special_tokens || id < st->special_token_begin) t.push_back(id);
        }
        tr->segment_text.push_back(tok->decode(t));
    }
    *out = tr;
    return WH_OK;
}

using namespace whi;

// ---- the reference's public SegmentSeeker helpers on caller-supplied word lists (known-answer tests, callers that bring their
// own alignment) ----------------------------------------------------------------------------------------------------------------
static bool words_from_c(const char* const* words, const int32_t* token_counts, const int32_t* tokens, const float* start,
                         const float* end, const float* probability, int n, std::vector<Word>& out) {
    if (n < 0 || (n && (!words || !token_counts || !tokens || !start || !end || !probability))) return false;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (!words[i] || token_counts[i] < 0) return false;
        Word w;
        w.word = words[i];
        w.tokens.assign(tokens + off, tokens + off + token_counts[i]);
        off += (size_t)token_counts[i];
        w.start = start[i]; w.end = end[i]; w.probability = probability[i];
        out.push_back(std::move(w));
    }
    return true;
}

static void words_into(wh_transcription* tr, const std::vector<Word>& ws) {
    for (auto& w : ws) {
        wh_word_timing wt{};
        wt.token_offset = (int)tr->word_tokens.size(); wt.n_tokens = (int)w.tokens.size();
        wt.start = w.start; wt.end = w.end; wt.probability = w.probability;
        tr->word_tokens.insert(tr->word_tokens.end(), w.tokens.begin(), w.tokens.end());
        tr->words.push_back(wt);
        tr->word_text.push_back(w.word);
    }
    tr->has_text = true;
}

// SegmentSeeker.mergePunctuations(alignment:prepended:appended:) (SegmentSeeker.swift:280-338); NULL punctuation sets = the defaults
extern "C" int wh_merge_punctuations(const char* const* words, const int32_t* word_token_counts, const int32_t* word_tokens,
                                     const float* start, const float* end, const float* probability, int n_words,
                                     const char* prepended, const char* appended, wh_transcription** out) {
    std::vector<Word> in;
    if (!out || !words_from_c(words, word_token_counts, word_tokens, start, end, probability, n_words, in))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_merge_punctuations: invalid argument");
    auto tr = new wh_transcription();
    words_into(tr, whi::merge_punctuations(in, prepended ? prepended : whi::kDefaultPrependPunctuations,
                                           appended ? appended : whi::kDefaultAppendPunctuations));
    *out = tr;
    return WH_OK;
}

// The tail of addWordTimestamps after findAlignment (SegmentSeeker.swift:472-495) on a caller-supplied alignment:
// calculateWordDurationConstraints, truncateLongWordsAtSentenceBoundaries, mergePunctuations, updateSegmentsWithWordTimings.
// `tok` is only needed when a merged word loses special tokens (its text is re-decoded) and may be NULL otherwise.
extern "C" int wh_update_segments_with_word_timings(const wh_tokenizer* tok, int special_token_begin, const wh_segment* segments, int n_segments,
                                                    const int32_t* tokens, int n_tokens, const char* const* words,
                                                    const int32_t* word_token_counts, const int32_t* word_tokens, const float* start,
                                                    const float* end, const float* probability, int n_words, int seek,
                                                    float last_speech_timestamp, float* median_out, float* max_duration_out,
                                                    wh_transcription** out) {
    std::vector<Word> alignment;
    if (!out || n_segments < 0 || (n_segments && !segments) || (n_tokens && !tokens) ||
        !words_from_c(words, word_token_counts, word_tokens, start, end, probability, n_words, alignment))
        return set_error(WH_ERR_INVALID_ARGUMENT, "wh_update_segments_with_word_timings: invalid argument");
    for (int s = 0; s < n_segments; ++s)
        if (segments[s].token_offset < 0 || segments[s].n_tokens < 0 || segments[s].token_offset + segments[s].n_tokens > n_tokens)
            return set_error(WH_ERR_INVALID_ARGUMENT, "wh_update_segments_with_word_timings: segment %d indexes outside the token array", s);
    float median = 0, max_duration = 0;
    word_duration_constraints(alignment, &median, &max_duration);
    if (median_out) *median_out = median;
    if (max_duration_out) *max_duration_out = max_duration;
    truncate_long_words(alignment, max_duration);
    if (!alignment.empty()) alignment = whi::merge_punctuations(alignment, whi::kDefaultPrependPunctuations, whi::kDefaultAppendPunctuations);
    for (auto& w : alignment) {
        bool loses = false, keeps = false;
        for (int t : w.tokens) (t < special_token_begin ? keeps : loses) = true;
        if (loses && keeps && !tok) return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "a merged word lost special tokens: a tokenizer is needed to re-decode it");
    }
    auto tr = new wh_transcription();
    tr->tokens.assign(tokens, tokens + n_tokens);
    tr->logprobs.assign((size_t)n_tokens, 0.0f);
    tr->segments.assign(segments, segments + n_segments);
    std::vector<SegWords> per_segment;
    update_segments_with_word_timings(tok, special_token_begin, tr->segments, tr->tokens.data(), alignment, seek, last_speech_timestamp, median,
                                      max_duration, per_segment);
    for (int s = 0; s < n_segments; ++s) {
        tr->segments[s].word_offset = (int)tr->words.size();
        words_into(tr, per_segment[s].words);
        tr->segments[s].n_words = (int)tr->words.size() - tr->segments[s].word_offset;
        tr->segment_text.emplace_back();
    }
    tr->has_text = true;
    *out = tr;
    return WH_OK;
}

// Host-only structures of the text side of the path: tokenizer, word timings, transcription container.
// No HIP types here - tokenizer.cpp / words.cpp / results.cpp are plain C++ and run without a GPU.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "whisperhip.h"

namespace whi {
int set_error(int code, const char* fmt, ...);
}

// WhisperTokenizerWrapper (Core/Models.swift:1165-1307) over a ByteLevel-BPE tokenizer.json
struct wh_tokenizer {
    std::vector<std::string> id_to_token;     // "" + has[id]==0 for holes
    std::vector<uint8_t> has, is_added, is_special;
    std::vector<std::string> id_bytes;        // ordinary tokens mapped back to raw bytes (ByteLevel decoder)
    std::unordered_map<std::string, int> token_to_id;
    bool clean_up = true;                     // tokenizer_config.json clean_up_tokenization_spaces (default true)
    wh_special_tokens special{};
    std::vector<int> language_tokens;         // allLanguageTokens, ascending

    std::string decode(const int32_t* tokens, int n, bool skip_special) const;
    std::string decode(const std::vector<int>& t, bool skip_special = false) const { return decode(t.data(), (int)t.size(), skip_special); }
    void split_on_unicode(const std::vector<int>& tokens, std::vector<std::string>& words, std::vector<std::vector<int>>& word_tokens) const;
    void split_on_spaces(const std::vector<int>& tokens, std::vector<std::string>& words, std::vector<std::vector<int>>& word_tokens) const;
    void split_to_word_tokens(const std::vector<int>& tokens, const char* language, std::vector<std::string>& words,
                              std::vector<std::vector<int>>& word_tokens) const;
};

namespace whi {

// WordTiming (Core/Models.swift:623-641)
struct Word {
    std::string word;
    std::vector<int> tokens;
    float start = 0, end = 0, probability = 0;
    float duration() const { return end - start; }
};

// UTF-8 helpers shared by the splitter and the punctuation merge
std::string utf8_repair(const std::string& bytes);                     // String(decoding:as: UTF8.self)
std::vector<uint32_t> utf8_scalars(const std::string& s);
std::string trim_swift_whitespaces(const std::string& s);             // trimmingCharacters(in: .whitespaces)
std::string trimming_special_token_characters(const std::string& s);  // "<|>" stripped from both ends
float rounded2(float x);                                               // Float.rounded(2)

// SegmentSeeker post-processing (Core/Text/SegmentSeeker.swift:280-338, 498-659)
std::vector<Word> merge_punctuations(const std::vector<Word>& alignment, const std::string& prepended, const std::string& appended);
extern const char* const kDefaultPrependPunctuations;
extern const char* const kDefaultAppendPunctuations;

}  // namespace whi

// TranscriptionResult (Core/Models.swift:447-466) flattened: segments index the flat token / log-prob arrays, words index the
// flat word-token array; texts exist when a tokenizer was attached.
struct wh_transcription {
    std::vector<wh_segment> segments;
    std::vector<wh_word_timing> words;
    std::vector<int32_t> tokens;
    std::vector<float> logprobs;
    std::vector<int32_t> word_tokens;
    std::vector<int32_t> seeks;
    std::vector<std::string> segment_text, word_text;
    std::string text, language;
    bool has_text = false;
    bool has_seek_time = false;
    bool words_enabled = false;      // word timestamps ran: segments carry `words` (possibly empty) instead of nil
    float seek_time = 0;
    int language_token = -1;
    bool language_set = false;       // detectedLanguage is fixed by the first detection / first decoded window (TranscribeTask.swift:352,375-377)
    wh_timings timings{};
};

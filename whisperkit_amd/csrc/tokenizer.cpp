// Tokenizer text side of the path (host only, no GPU): ids -> text and word splitting.
//
//   decode(tokens:)              ArgmaxCore/External/Tokenizers/Tokenizer.swift:510-525 (vendored swift-transformers 1.1.6)
//   ByteLevelDecoder             ArgmaxCore/External/Tokenizers/Decoder.swift:126-165 + byte table of ByteEncoder.swift
//   cleanUp(text:)               Tokenizer.swift:433-449
//   WhisperTokenizerWrapper      WhisperKit/Core/Models.swift:1165-1307 (special tokens, language tokens, splitToWordTokens)
//
// Only decoding is on the path (the reference encodes text only for CLI prompts); encode is out of scope.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "json.h"
#include "text.h"
#include "unicode_tables.h"

using whi::set_error;

namespace whi {

// ---- UTF-8 ---------------------------------------------------------------------------------------------------------------
// String(decoding: bytes, as: UTF8.self): every maximal ill-formed subpart becomes one U+FFFD (Unicode 3.9, the policy Swift,
// Rust's from_utf8_lossy and Python's errors="replace" share).
std::string utf8_repair(const std::string& in) {
    std::string out;
    out.reserve(in.size());
    const size_t n = in.size();
    size_t i = 0;
    auto cont = [&](size_t k, unsigned lo, unsigned hi) { return k < n && (unsigned char)in[k] >= lo && (unsigned char)in[k] <= hi; };
    while (i < n) {
        unsigned char b = (unsigned char)in[i];
        if (b < 0x80) { out.push_back((char)b); ++i; continue; }
        int need = 0;
        unsigned lo = 0x80, hi = 0xBF;
        if (b >= 0xC2 && b <= 0xDF) need = 1;
        else if (b == 0xE0) { need = 2; lo = 0xA0; }
        else if ((b >= 0xE1 && b <= 0xEC) || b == 0xEE || b == 0xEF) need = 2;
        else if (b == 0xED) { need = 2; hi = 0x9F; }
        else if (b == 0xF0) { need = 3; lo = 0x90; }
        else if (b >= 0xF1 && b <= 0xF3) need = 3;
        else if (b == 0xF4) { need = 3; hi = 0x8F; }
        if (!need) { out += "\xEF\xBF\xBD"; ++i; continue; }
        size_t k = i + 1;
        int got = 0;
        if (cont(k, lo, hi)) { ++k; ++got; while (got < need && cont(k, 0x80, 0xBF)) { ++k; ++got; } }
        if (got == need) out.append(in, i, k - i);
        else out += "\xEF\xBF\xBD";
        i = k;
    }
    return out;
}

std::vector<uint32_t> utf8_scalars(const std::string& s) {   // input is well formed (output of utf8_repair or JSON strings)
    std::vector<uint32_t> v;
    for (size_t i = 0; i < s.size();) {
        unsigned char b = (unsigned char)s[i];
        uint32_t cp; int len;
        if (b < 0x80) { cp = b; len = 1; }
        else if (b < 0xE0) { cp = b & 0x1F; len = 2; }
        else if (b < 0xF0) { cp = b & 0x0F; len = 3; }
        else { cp = b & 0x07; len = 4; }
        for (int k = 1; k < len && i + k < s.size(); ++k) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3F);
        v.push_back(cp);
        i += len;
    }
    return v;
}

static size_t scalar_len(unsigned char b) { return b < 0x80 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4; }

std::string trim_swift_whitespaces(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b) {
        size_t l = std::min(scalar_len((unsigned char)s[a]), b - a);
        auto sc = utf8_scalars(s.substr(a, l));
        if (sc.empty() || !wh::is_swift_whitespace(sc[0])) break;
        a += l;
    }
    while (b > a) {
        size_t k = b - 1;
        while (k > a && ((unsigned char)s[k] & 0xC0) == 0x80) --k;
        auto sc = utf8_scalars(s.substr(k, b - k));
        if (sc.empty() || !wh::is_swift_whitespace(sc[0])) break;
        b = k;
    }
    return s.substr(a, b - a);
}

std::string trimming_special_token_characters(const std::string& s) {
    size_t a = 0, b = s.size();
    auto sp = [](char c) { return c == '<' || c == '|' || c == '>'; };
    while (a < b && sp(s[a])) ++a;
    while (b > a && sp(s[b - 1])) --b;
    return s.substr(a, b - a);
}

float rounded2(float x) { return roundf(x * 100.0f) / 100.0f; }   // ArgmaxCore/FoundationExtensions.swift:10-13

}  // namespace whi

// ---- byte-level table ------------------------------------------------------------------------------------------------------
// GPT-2 byte <-> printable character: printable Latin-1 bytes map to themselves, the other 68 to U+0100..U+0143 in byte order.
static void byte_decoder_table(std::unordered_map<uint32_t, uint8_t>& dec) {
    bool direct[256] = {};
    for (int b = 33; b <= 126; ++b) direct[b] = true;
    for (int b = 161; b <= 172; ++b) direct[b] = true;
    for (int b = 174; b <= 255; ++b) direct[b] = true;
    uint32_t next = 256;
    for (int b = 0; b < 256; ++b) dec[direct[b] ? (uint32_t)b : next++] = (uint8_t)b;
}

static void replace_all(std::string& s, const char* a, const char* b) {
    const size_t la = strlen(a), lb = strlen(b);
    for (size_t pos = 0; (pos = s.find(a, pos)) != std::string::npos; pos += lb) s.replace(pos, la, b);
}

std::string wh_tokenizer::decode(const int32_t* tokens, int n, bool skip_special) const {
    std::string out, run;
    auto flush = [&]() { if (!run.empty()) { out += whi::utf8_repair(run); run.clear(); } };
    for (int i = 0; i < n; ++i) {
        int t = tokens[i];
        if (t < 0 || t >= (int)has.size() || !has[t]) continue;      // compactMap drops unknown ids
        if (skip_special && is_special[t]) continue;
        if (is_added[t]) { flush(); out += id_to_token[t]; }
        else run += id_bytes[t];
    }
    flush();
    if (clean_up) {   // Tokenizer.swift:433-449, in this order
        static const char* const rules[][2] = {{" .", "."}, {" ?", "?"}, {" !", "!"}, {" ,", ","}, {" ' ", "'"}, {" n't", "n't"},
                                               {" 'm", "'m"}, {" 's", "'s"}, {" 've", "'ve"}, {" 're", "'re"}};
        for (auto& r : rules) replace_all(out, r[0], r[1]);
    }
    return out;
}

// Core/Models.swift:1226-1254.  `decoded.range(of: "\u{fffd}")` is an index into `decoded`; used on `decodedFull` it addresses the
// same UTF-8 offset of the full string (the reference does not add `unicodeOffset`, unlike openai/whisper) - kept as written.
void wh_tokenizer::split_on_unicode(const std::vector<int>& tokens, std::vector<std::string>& words,
                                    std::vector<std::vector<int>>& word_tokens) const {
    static const std::string rep = "\xEF\xBF\xBD";
    const std::string full = decode(tokens);
    std::vector<int> cur;
    for (int t : tokens) {
        cur.push_back(t);
        std::string dec = decode(cur);
        size_t at = dec.find(rep);
        bool in_full = at != std::string::npos && at + rep.size() <= full.size() && full.compare(at, rep.size(), rep) == 0;
        if (at == std::string::npos || in_full) {
            words.push_back(std::move(dec));
            word_tokens.push_back(cur);
            cur.clear();
        }
    }
}

// Core/Models.swift:1256-1279
void wh_tokenizer::split_on_spaces(const std::vector<int>& tokens, std::vector<std::string>& words,
                                   std::vector<std::vector<int>>& word_tokens) const {
    std::vector<std::string> sub;
    std::vector<std::vector<int>> sub_tokens;
    split_on_unicode(tokens, sub, sub_tokens);
    for (size_t i = 0; i < sub.size(); ++i) {
        const std::string& s = sub[i];
        const bool is_special_word = sub_tokens[i][0] >= special.special_token_begin;
        bool with_space = !s.empty() && s[0] == ' ';
        if (with_space && s.size() > 1) {          // " " + combining mark is one grapheme: hasPrefix(" ") is false
            auto sc = whi::utf8_scalars(s.substr(1, 4));
            if (!sc.empty() && wh::is_extend_mark(sc[0])) with_space = false;
        }
        auto stripped = whi::utf8_scalars(whi::trim_swift_whitespaces(s));
        const bool punctuation = stripped.size() == 1 && wh::is_punctuation(stripped[0]);
        if (is_special_word || with_space || punctuation || words.empty()) {
            words.push_back(s);
            word_tokens.push_back(sub_tokens[i]);
        } else {
            words.back() += s;
            word_tokens.back().insert(word_tokens.back().end(), sub_tokens[i].begin(), sub_tokens[i].end());
        }
    }
}

// Core/Models.swift:1293-1306; the splitter is chosen by the caller's language code instead of NLLanguageRecognizer (not available
// off Apple platforms) - the same rule openai/whisper applies with the tokenizer's language.
void wh_tokenizer::split_to_word_tokens(const std::vector<int>& tokens, const char* language, std::vector<std::string>& words,
                                        std::vector<std::vector<int>>& word_tokens) const {
    static const char* const unicode_languages[] = {"zh", "ja", "th", "lo", "my", "yue"};
    bool unicode = false;
    if (language) for (auto l : unicode_languages) if (!strcmp(l, language)) unicode = true;
    if (unicode) split_on_unicode(tokens, words, word_tokens);
    else split_on_spaces(tokens, words, word_tokens);
}

// ---- loading -----------------------------------------------------------------------------------------------------------------
static const char* const kLanguageCodes[] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi", "he", "uk", "el", "ms",
    "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn",
    "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be",
    "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha",
    "ba", "jw", "su", "yue"};   // Constants.languages values, Core/Models.swift:1335-1449

extern "C" int wh_tokenizer_load(const char* tokenizer_json_path, wh_tokenizer** out) {
    if (!tokenizer_json_path || !out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_tokenizer_load: null argument");
    *out = nullptr;
    std::string buf, err;
    if (!wh::read_file(tokenizer_json_path, buf)) return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "cannot read %s", tokenizer_json_path);
    wh::JsonValue root;
    if (!wh::JsonParser(buf.data(), buf.size()).parse(root, err))
        return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "%s: %s", tokenizer_json_path, err.c_str());
    const wh::JsonValue* model = root.get("model");
    const wh::JsonValue* vocab = model ? model->get("vocab") : nullptr;
    const wh::JsonValue* mtype = model ? model->get("type") : nullptr;
    const wh::JsonValue* dec = root.get("decoder");
    const wh::JsonValue* dtype = dec ? dec->get("type") : nullptr;
    if (!vocab || vocab->kind != wh::JsonValue::Object || !mtype || mtype->str != "BPE" || !dtype || dtype->str != "ByteLevel")
        return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "%s: not a ByteLevel BPE tokenizer", tokenizer_json_path);
    auto t = new wh_tokenizer();
    auto put = [&](const std::string& tok, int id, bool added, bool special) {
        if (id < 0 || id > (1 << 22)) return;   // ids far beyond any Whisper vocabulary are ignored
        if (id >= (int)t->has.size()) {
            t->has.resize(id + 1, 0); t->is_added.resize(id + 1, 0); t->is_special.resize(id + 1, 0);
            t->id_to_token.resize(id + 1); t->id_bytes.resize(id + 1);
        }
        t->has[id] = 1; t->is_added[id] = added; t->is_special[id] = special;
        t->id_to_token[id] = tok;
        t->token_to_id[tok] = id;
    };
    for (auto& kv : vocab->obj) if (kv.second.kind == wh::JsonValue::Number) put(kv.first, (int)kv.second.num, false, false);
    if (const wh::JsonValue* added = root.get("added_tokens"))
        for (auto& a : added->arr) {
            const wh::JsonValue *id = a.get("id"), *content = a.get("content"), *sp = a.get("special");
            if (id && content && id->kind == wh::JsonValue::Number && content->kind == wh::JsonValue::String)
                put(content->str, (int)id->num, true, sp && sp->kind == wh::JsonValue::Bool && sp->b);
        }
    std::unordered_map<uint32_t, uint8_t> bd;
    byte_decoder_table(bd);
    for (size_t id = 0; id < t->has.size(); ++id) {
        if (!t->has[id] || t->is_added[id]) continue;
        std::string raw;
        for (uint32_t cp : whi::utf8_scalars(t->id_to_token[id])) {
            auto it = bd.find(cp);
            if (it == bd.end()) { delete t; return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "token %zu has a character outside the byte-level alphabet", id); }
            raw.push_back((char)it->second);
        }
        t->id_bytes[id] = std::move(raw);
    }
    // tokenizer_config.json next to it: clean_up_tokenization_spaces (Tokenizer.swift:407, default true)
    std::string dir(tokenizer_json_path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    std::string cbuf;
    if (wh::read_file((dir + "/tokenizer_config.json").c_str(), cbuf)) {
        wh::JsonValue cfg;
        if (wh::JsonParser(cbuf.data(), cbuf.size()).parse(cfg, err))
            if (const wh::JsonValue* c = cfg.get("clean_up_tokenization_spaces")) if (c->kind == wh::JsonValue::Bool) t->clean_up = c->b;
    }
    // WhisperTokenizerWrapper.init (Core/Models.swift:1198-1224), defaults :1309-1322
    auto id_of = [&](const char* tok, int dflt) { auto it = t->token_to_id.find(tok); return it == t->token_to_id.end() ? dflt : it->second; };
    wh_special_tokens& s = t->special;
    s.end_token = id_of("<|endoftext|>", 50257);
    s.english_token = id_of("<|en|>", 50259);
    s.no_speech_token = id_of("<|nospeech|>", 50362);
    s.no_timestamps_token = id_of("<|notimestamps|>", 50363);
    s.special_token_begin = id_of("<|endoftext|>", 50257);
    s.start_of_previous_token = id_of("<|startofprev|>", 50361);
    s.start_of_transcript_token = id_of("<|startoftranscript|>", 50258);
    s.time_token_begin = id_of("<|0.00|>", 50364);
    s.transcribe_token = id_of("<|transcribe|>", 50359);
    s.translate_token = id_of("<|translate|>", 50358);
    s.whitespace_token = id_of(" ", 220);
    for (auto code : kLanguageCodes) {
        int id = id_of((std::string("<|") + code + "|>").c_str(), -1);
        if (id > s.special_token_begin) t->language_tokens.push_back(id);
    }
    std::sort(t->language_tokens.begin(), t->language_tokens.end());
    s.language_token_begin = t->language_tokens.empty() ? -1 : t->language_tokens.front();
    s.n_language_tokens = (int)t->language_tokens.size();
    if (!t->language_tokens.empty() && t->language_tokens.back() - t->language_tokens.front() + 1 != (int)t->language_tokens.size()) {
        delete t;
        return set_error(WH_ERR_TOKENIZER_UNAVAILABLE, "language tokens are not one contiguous id range");
    }
    *out = t;
    return WH_OK;
}

extern "C" void wh_tokenizer_destroy(wh_tokenizer* t) { delete t; }
extern "C" int wh_tokenizer_vocab_size(const wh_tokenizer* t) { return t ? (int)t->has.size() : 0; }

static int copy_out(const std::string& s, char* out, int capacity) {
    if (out && capacity > 0) {
        size_t n = std::min(s.size(), (size_t)capacity - 1);
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int)s.size();
}

extern "C" int wh_tokenizer_decode(const wh_tokenizer* t, const int32_t* tokens, int n, int skip_special_tokens, char* out, int capacity) {
    if (!t || (n > 0 && !tokens) || n < 0) { set_error(WH_ERR_INVALID_ARGUMENT, "wh_tokenizer_decode: bad argument"); return -1; }
    return copy_out(t->decode(tokens, n, skip_special_tokens != 0), out, capacity);
}

extern "C" int wh_tokenizer_token_to_id(const wh_tokenizer* t, const char* token) {
    if (!t || !token) return -1;
    auto it = t->token_to_id.find(token);
    return it == t->token_to_id.end() ? -1 : it->second;
}

extern "C" int wh_tokenizer_id_to_token(const wh_tokenizer* t, int id, char* out, int capacity) {
    if (!t || id < 0 || id >= (int)t->has.size() || !t->has[id]) return -1;
    return copy_out(t->id_to_token[id], out, capacity);
}

extern "C" int wh_tokenizer_special_tokens(const wh_tokenizer* t, wh_special_tokens* out) {
    if (!t || !out) return set_error(WH_ERR_INVALID_ARGUMENT, "wh_tokenizer_special_tokens: null argument");
    *out = t->special;
    return WH_OK;
}

extern "C" int wh_tokenizer_split_to_word_tokens(const wh_tokenizer* t, const int32_t* tokens, int n, const char* language_code,
                                                 int32_t* word_token_counts, int32_t* word_byte_counts, int counts_capacity,
                                                 char* words_out, int words_capacity, int* words_bytes) {
    if (!t || (n > 0 && !tokens) || n < 0) { set_error(WH_ERR_INVALID_ARGUMENT, "wh_tokenizer_split_to_word_tokens: bad argument"); return -1; }
    std::vector<int> ids(tokens, tokens + n);
    std::vector<std::string> words;
    std::vector<std::vector<int>> wt;
    t->split_to_word_tokens(ids, language_code, words, wt);
    size_t bytes = 0;
    for (auto& w : words) bytes += w.size();
    if (words_bytes) *words_bytes = (int)bytes;
    if (((word_token_counts || word_byte_counts) && (int)words.size() > counts_capacity) || (words_out && (int)bytes > words_capacity)) {
        set_error(WH_ERR_INVALID_ARGUMENT, "wh_tokenizer_split_to_word_tokens: %zu words / %zu bytes do not fit", words.size(), bytes);
        return -2;
    }
    size_t off = 0;
    for (size_t i = 0; i < words.size(); ++i) {
        if (word_token_counts) word_token_counts[i] = (int)wt[i].size();
        if (word_byte_counts) word_byte_counts[i] = (int)words[i].size();
        if (words_out) { memcpy(words_out + off, words[i].data(), words[i].size()); off += words[i].size(); }
    }
    return (int)words.size();
}

// Minimal JSON reader / string escaper for the host side (tokenizer.json, tokenizer_config.json, result writer).
// Recursive descent over a memory buffer; objects keep insertion order; numbers are doubles.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace wh {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;

    const JsonValue* get(const char* key) const {
        if (kind != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

inline void utf8_append(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
}

class JsonParser {
public:
    JsonParser(const char* p, size_t n) : p_(p), end_(p + n) {}
    bool parse(JsonValue& out, std::string& err) {
        if (!value(out, 0)) { err = err_.empty() ? "malformed JSON" : err_; return false; }
        ws();
        if (p_ != end_) { err = "trailing characters after JSON value"; return false; }
        return true;
    }

private:
    const char *p_, *end_;
    std::string err_;
    void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_; }
    bool fail(const char* m) { if (err_.empty()) err_ = m; return false; }
    static int hex(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
    bool hex4(uint32_t& v) {
        if (end_ - p_ < 4) return false;
        v = 0;
        for (int i = 0; i < 4; ++i) { int h = hex(p_[i]); if (h < 0) return false; v = v * 16 + (uint32_t)h; }
        p_ += 4;
        return true;
    }
    bool string(std::string& s) {
        if (p_ >= end_ || *p_ != '"') return fail("expected string");
        ++p_;
        s.clear();
        while (p_ < end_) {
            char c = *p_++;
            if (c == '"') return true;
            if (c != '\\') { s.push_back(c); continue; }
            if (p_ >= end_) break;
            char e = *p_++;
            switch (e) {
                case '"': s.push_back('"'); break;
                case '\\': s.push_back('\\'); break;
                case '/': s.push_back('/'); break;
                case 'b': s.push_back('\b'); break;
                case 'f': s.push_back('\f'); break;
                case 'n': s.push_back('\n'); break;
                case 'r': s.push_back('\r'); break;
                case 't': s.push_back('\t'); break;
                case 'u': {
                    uint32_t cp;
                    if (!hex4(cp)) return fail("bad \\u escape");
                    if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        const char* save = p_;
                        p_ += 2;
                        uint32_t lo;
                        if (hex4(lo) && lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else { p_ = save; cp = 0xFFFD; }
                    } else if (cp >= 0xD800 && cp < 0xE000) cp = 0xFFFD;
                    utf8_append(s, cp);
                    break;
                }
                default: return fail("bad escape");
            }
        }
        return fail("unterminated string");
    }
    bool value(JsonValue& v, int depth) {
        if (depth > 64) return fail("JSON nested too deeply");
        ws();
        if (p_ >= end_) return fail("unexpected end of JSON");
        char c = *p_;
        if (c == '{') {
            ++p_;
            v.kind = JsonValue::Object;
            ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return true; }
            while (true) {
                ws();
                std::string k;
                if (!string(k)) return false;
                ws();
                if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(k), JsonValue());
                if (!value(v.obj.back().second, depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            v.kind = JsonValue::Array;
            ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return true; }
            while (true) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JsonValue::String; return string(v.str); }
        if (end_ - p_ >= 4 && !memcmp(p_, "true", 4)) { p_ += 4; v.kind = JsonValue::Bool; v.b = true; return true; }
        if (end_ - p_ >= 5 && !memcmp(p_, "false", 5)) { p_ += 5; v.kind = JsonValue::Bool; v.b = false; return true; }
        if (end_ - p_ >= 4 && !memcmp(p_, "null", 4)) { p_ += 4; v.kind = JsonValue::Null; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p_;
            while (p_ < end_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) ++p_;
            std::string t(s, p_);
            char* e = nullptr;
            v.num = strtod(t.c_str(), &e);
            if (!e || *e) return fail("bad number");
            v.kind = JsonValue::Number;
            return true;
        }
        return fail("unexpected character in JSON");
    }
};

inline bool read_file(const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

// JSON string literal (UTF-8 passed through, control characters and quotes escaped)
inline void json_escape(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
                else out.push_back((char)c);
        }
    }
    out.push_back('"');
}

}  // namespace wh

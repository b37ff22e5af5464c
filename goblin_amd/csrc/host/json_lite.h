// Minimal JSON DOM for Goblin scene files.  Written from scratch (the reference
// vendors nlohmann/json, which is third-party code we do not copy).  The one
// behaviour the scene loader depends on is preserved: a number literal without
// '.', 'e' or 'E' is an INTEGER and is not visible to float lookups
// (/root/reference/src/GoblinContextLoader.cpp:40-45).
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace gbl_json {

struct Value {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    long long i = 0;
    double d = 0.0;
    std::string s;
    std::vector<Value> arr;
    // a repeated key keeps its LAST value (the reference's nlohmann DOM parser does operator[](key) and assigns,
    // json.hpp:2899,2987); keys iterate sorted (std::map like the reference's json type)
    std::map<std::string, Value> obj;

    bool is_number() const { return kind == Int || kind == Float; }
    // json -> float conversion: double (or int64) narrowed with a cast
    float as_float() const { return kind == Int ? static_cast<float>(i) : static_cast<float>(d); }
    const Value* find(const std::string& k) const {
        if (kind != Object) return nullptr;
        auto it = obj.find(k);
        return it == obj.end() ? nullptr : &it->second;
    }
};

class Parser {
public:
    Parser(const char* text, size_t n) : p_(text), end_(text + n) {}
    bool parse(Value* out, std::string* err) {
        skip();
        if (!value(out)) {
            *err = err_.empty() ? "json syntax error" : err_;
            return false;
        }
        skip();
        if (p_ != end_) {
            *err = "trailing characters after json document";
            return false;
        }
        return true;
    }

private:
    const char* p_;
    const char* end_;
    std::string err_;

    void skip() {
        while (p_ < end_) {
            char c = *p_;
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                ++p_;
            } else if (c == '/' && p_ + 1 < end_ && p_[1] == '/') {  // tolerate // comments
                while (p_ < end_ && *p_ != '\n') ++p_;
            } else {
                break;
            }
        }
    }
    bool fail(const char* m) {
        if (err_.empty()) err_ = m;
        return false;
    }
    bool literal(const char* lit) {
        size_t n = strlen(lit);
        if (static_cast<size_t>(end_ - p_) < n || strncmp(p_, lit, n) != 0) return fail("bad literal");
        p_ += n;
        return true;
    }
    bool string(std::string* out) {
        if (p_ >= end_ || *p_ != '"') return fail("expected string");
        ++p_;
        out->clear();
        while (p_ < end_ && *p_ != '"') {
            char c = *p_++;
            if (c == '\\') {
                if (p_ >= end_) return fail("bad escape");
                char e = *p_++;
                switch (e) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'u': {
                        if (end_ - p_ < 4) return fail("bad \\u escape");
                        unsigned cp = static_cast<unsigned>(strtoul(std::string(p_, 4).c_str(), nullptr, 16));
                        p_ += 4;
                        if (cp < 0x80) {
                            out->push_back(static_cast<char>(cp));
                        } else if (cp < 0x800) {
                            out->push_back(static_cast<char>(0xC0 | (cp >> 6)));
                            out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        } else {
                            out->push_back(static_cast<char>(0xE0 | (cp >> 12)));
                            out->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
                            out->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                        }
                        break;
                    }
                    default: out->push_back(e); break;  // \" \\ \/
                }
            } else {
                out->push_back(c);
            }
        }
        if (p_ >= end_) return fail("unterminated string");
        ++p_;
        return true;
    }
    bool number(Value* out) {
        const char* s = p_;
        bool is_float = false;
        if (p_ < end_ && (*p_ == '-' || *p_ == '+')) ++p_;
        while (p_ < end_) {
            char c = *p_;
            if (c >= '0' && c <= '9') {
                ++p_;
            } else if (c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-') {
                is_float = true;
                ++p_;
            } else {
                break;
            }
        }
        if (p_ == s) return fail("expected number");
        std::string tok(s, p_ - s);
        if (is_float) {
            out->kind = Value::Float;
            out->d = strtod(tok.c_str(), nullptr);
        } else {
            out->kind = Value::Int;
            out->i = strtoll(tok.c_str(), nullptr, 10);
        }
        return true;
    }
    // containers nest at most kMaxDepth deep: a hostile file cannot run the recursive descent out of stack
    static const int kMaxDepth = 256;
    int depth_ = 0;
    struct DepthGuard {
        int& d;
        explicit DepthGuard(int& dd) : d(dd) { ++d; }
        ~DepthGuard() { --d; }
    };
    bool value(Value* out) {
        DepthGuard guard(depth_);
        if (depth_ > kMaxDepth) return fail("json nested too deeply");
        skip();
        if (p_ >= end_) return fail("unexpected end of json");
        char c = *p_;
        if (c == '{') {
            ++p_;
            out->kind = Value::Object;
            skip();
            if (p_ < end_ && *p_ == '}') {
                ++p_;
                return true;
            }
            while (true) {
                skip();
                std::string key;
                if (!string(&key)) return false;
                skip();
                if (p_ >= end_ || *p_ != ':') return fail("expected ':'");
                ++p_;
                Value v;
                if (!value(&v)) return false;
                out->obj[key] = std::move(v);
                skip();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == '}') {
                    ++p_;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            out->kind = Value::Array;
            skip();
            if (p_ < end_ && *p_ == ']') {
                ++p_;
                return true;
            }
            while (true) {
                Value v;
                if (!value(&v)) return false;
                out->arr.push_back(std::move(v));
                skip();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == ']') {
                    ++p_;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            out->kind = Value::String;
            return string(&out->s);
        }
        if (c == 't') {
            out->kind = Value::Bool;
            out->b = true;
            return literal("true");
        }
        if (c == 'f') {
            out->kind = Value::Bool;
            out->b = false;
            return literal("false");
        }
        if (c == 'n') {
            out->kind = Value::Null;
            return literal("null");
        }
        return number(out);
    }
};

}  // namespace gbl_json

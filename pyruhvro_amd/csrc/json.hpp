// Minimal JSON DOM parser for Avro schema documents (host side).
// Enough of RFC 8259 for schemas: objects keep member order, strings are
// unescaped to UTF-8, numbers are kept as double + original text.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rh {
namespace json {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0;
  std::string str;                                   // String payload / Number text
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  bool is_string() const { return type == String; }
  bool is_number() const { return type == Number; }
  bool is_object() const { return type == Object; }
  bool is_array() const { return type == Array; }
  const Value* get(const char* key) const {
    if (type != Object) return nullptr;
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

class Parser {
 public:
  Parser(const char* s, size_t n) : p_(s), e_(s + n) {}
  Value parse() {
    Value v = value(0);
    ws();
    if (p_ != e_) fail("trailing characters");
    return v;
  }

 private:
  const char* p_;
  const char* e_;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("Failed to parse schema from JSON: ") + m); }
  void ws() {
    while (p_ < e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
  }
  bool lit(const char* w) {
    size_t n = 0;
    while (w[n]) ++n;
    if ((size_t)(e_ - p_) < n) return false;
    for (size_t i = 0; i < n; i++)
      if (p_[i] != w[i]) return false;
    p_ += n;
    return true;
  }
  static void utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) {
      out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
    } else {
      out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F));
      out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F));
    }
  }
  uint32_t hex4() {
    if (e_ - p_ < 4) fail("bad \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string string() {
    std::string out;
    ++p_;  // opening quote
    for (;;) {
      if (p_ >= e_) fail("unterminated string");
      char c = *p_++;
      if (c == '"') return out;
      if (c != '\\') { out += c; continue; }
      if (p_ >= e_) fail("unterminated escape");
      char x = *p_++;
      switch (x) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            uint32_t lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
  }
  Value value(int depth) {
    if (depth > 256) fail("nesting too deep");
    ws();
    if (p_ >= e_) fail("unexpected end of input");
    Value v;
    char c = *p_;
    if (c == '{') {
      v.type = Value::Object;
      ++p_;
      ws();
      if (p_ < e_ && *p_ == '}') { ++p_; return v; }
      for (;;) {
        ws();
        if (p_ >= e_ || *p_ != '"') fail("expected object key");
        std::string k = string();
        ws();
        if (p_ >= e_ || *p_ != ':') fail("expected ':'");
        ++p_;
        v.obj.emplace_back(std::move(k), value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == '}') { ++p_; return v; }
        fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.type = Value::Array;
      ++p_;
      ws();
      if (p_ < e_ && *p_ == ']') { ++p_; return v; }
      for (;;) {
        v.arr.push_back(value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == ']') { ++p_; return v; }
        fail("expected ',' or ']'");
      }
    }
    if (c == '"') { v.type = Value::String; v.str = string(); return v; }
    if (lit("true")) { v.type = Value::Bool; v.b = true; return v; }
    if (lit("false")) { v.type = Value::Bool; v.b = false; return v; }
    if (lit("null")) { v.type = Value::Null; return v; }
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char* s = p_;
      while (p_ < e_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) ++p_;
      v.type = Value::Number;
      v.str.assign(s, p_);
      v.num = std::strtod(v.str.c_str(), nullptr);
      return v;
    }
    fail("unexpected character");
  }
};

inline Value parse(const char* s, size_t n) { return Parser(s, n).parse(); }

}  // namespace json
}  // namespace rh

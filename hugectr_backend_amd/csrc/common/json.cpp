#include "json.h"

#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace hps {

class JsonParser {
 public:
  JsonParser(const std::string& s) : p_(s.data()), end_(s.data() + s.size()), begin_(s.data()) {}

  bool ParseDocument(Json* out, std::string* err) {
    SkipWs();
    if (!ParseValue(out, 0)) { if (err) *err = err_; return false; }
    SkipWs();
    if (p_ != end_) { Fail("trailing characters"); if (err) *err = err_; return false; }
    return true;
  }

 private:
  const char* p_;
  const char* end_;
  const char* begin_;
  std::string err_;

  bool Fail(const char* what) {
    if (err_.empty()) {
      err_ = std::string("JSON parse error at offset ") + std::to_string(p_ - begin_) + ": " + what;
    }
    return false;
  }
  void SkipWs() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
  }
  bool Lit(const char* s) {
    const size_t n = strlen(s);
    if ((size_t)(end_ - p_) >= n && memcmp(p_, s, n) == 0) { p_ += n; return true; }
    return false;
  }

  bool ParseValue(Json* v, int depth) {
    if (depth > 128) return Fail("nesting too deep");
    SkipWs();
    if (p_ >= end_) return Fail("unexpected end of input");
    switch (*p_) {
      case '{': return ParseObject(v, depth);
      case '[': return ParseArray(v, depth);
      case '"': v->type_ = Json::Type::String; return ParseString(&v->str_);
      case 't': if (Lit("true")) { v->type_ = Json::Type::Bool; v->bool_ = true; return true; } return Fail("bad literal");
      case 'f': if (Lit("false")) { v->type_ = Json::Type::Bool; v->bool_ = false; return true; } return Fail("bad literal");
      case 'n': if (Lit("null")) { v->type_ = Json::Type::Null; return true; } return Fail("bad literal");
      default: return ParseNumber(v);
    }
  }

  bool ParseObject(Json* v, int depth) {
    v->type_ = Json::Type::Object;
    ++p_;
    SkipWs();
    if (p_ < end_ && *p_ == '}') { ++p_; return true; }
    for (;;) {
      SkipWs();
      if (p_ >= end_ || *p_ != '"') return Fail("expected object key");
      std::string key;
      if (!ParseString(&key)) return false;
      SkipWs();
      if (p_ >= end_ || *p_ != ':') return Fail("expected ':'");
      ++p_;
      Json child;
      if (!ParseValue(&child, depth + 1)) return false;
      v->Set(key, std::move(child));  // duplicate key: last wins
      SkipWs();
      if (p_ < end_ && *p_ == ',') {
        ++p_;
        SkipWs();
        if (p_ < end_ && *p_ == '}') { ++p_; return true; }  // tolerate trailing comma
        continue;
      }
      if (p_ < end_ && *p_ == '}') { ++p_; return true; }
      return Fail("expected ',' or '}'");
    }
  }

  bool ParseArray(Json* v, int depth) {
    v->type_ = Json::Type::Array;
    ++p_;
    SkipWs();
    if (p_ < end_ && *p_ == ']') { ++p_; return true; }
    for (;;) {
      Json child;
      if (!ParseValue(&child, depth + 1)) return false;
      v->arr_.push_back(std::move(child));
      SkipWs();
      if (p_ < end_ && *p_ == ',') {
        ++p_;
        SkipWs();
        if (p_ < end_ && *p_ == ']') { ++p_; return true; }
        continue;
      }
      if (p_ < end_ && *p_ == ']') { ++p_; return true; }
      return Fail("expected ',' or ']'");
    }
  }

  static void AppendUtf8(std::string* s, uint32_t cp) {
    if (cp < 0x80) s->push_back((char)cp);
    else if (cp < 0x800) { s->push_back((char)(0xC0 | (cp >> 6))); s->push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s->push_back((char)(0xE0 | (cp >> 12))); s->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s->push_back((char)(0x80 | (cp & 0x3F))); }
    else { s->push_back((char)(0xF0 | (cp >> 18))); s->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s->push_back((char)(0x80 | (cp & 0x3F))); }
  }
  bool Hex4(uint32_t* out) {
    if (end_ - p_ < 4) return Fail("bad \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else return Fail("bad \\u escape");
    }
    *out = v;
    return true;
  }

  bool ParseString(std::string* s) {
    ++p_;  // opening quote
    s->clear();
    while (p_ < end_) {
      const char c = *p_++;
      if (c == '"') return true;
      if (c != '\\') { s->push_back(c); continue; }
      if (p_ >= end_) break;
      const char e = *p_++;
      switch (e) {
        case '"': s->push_back('"'); break;
        case '\\': s->push_back('\\'); break;
        case '/': s->push_back('/'); break;
        case 'b': s->push_back('\b'); break;
        case 'f': s->push_back('\f'); break;
        case 'n': s->push_back('\n'); break;
        case 'r': s->push_back('\r'); break;
        case 't': s->push_back('\t'); break;
        case 'u': {
          uint32_t cp;
          if (!Hex4(&cp)) return false;
          if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            uint32_t lo;
            if (!Hex4(&lo)) return false;
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          AppendUtf8(s, cp);
          break;
        }
        default: return Fail("bad escape");
      }
    }
    return Fail("unterminated string");
  }

  bool ParseNumber(Json* v) {
    const char* s = p_;
    if (p_ < end_ && (*p_ == '-' || *p_ == '+')) ++p_;
    bool is_int = true, any = false;
    while (p_ < end_ && isdigit((unsigned char)*p_)) { ++p_; any = true; }
    if (p_ < end_ && *p_ == '.') { is_int = false; ++p_; while (p_ < end_ && isdigit((unsigned char)*p_)) { ++p_; any = true; } }
    if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
      is_int = false; ++p_;
      if (p_ < end_ && (*p_ == '-' || *p_ == '+')) ++p_;
      while (p_ < end_ && isdigit((unsigned char)*p_)) ++p_;
    }
    if (!any) return Fail("invalid value");
    const std::string tok(s, p_ - s);
    if (is_int) {
      errno = 0;
      char* e = nullptr;
      const long long iv = strtoll(tok.c_str(), &e, 10);
      if (errno == 0 && e && *e == 0) {
        v->type_ = Json::Type::Int; v->int_ = iv; v->dbl_ = (double)iv;
        return true;
      }
    }
    v->type_ = Json::Type::Double;
    v->dbl_ = strtod(tok.c_str(), nullptr);
    v->int_ = (int64_t)v->dbl_;
    return true;
  }
};

bool Json::Parse(const std::string& text, Json* out, std::string* err) {
  *out = Json();
  JsonParser p(text);
  return p.ParseDocument(out, err);
}

bool Json::ParseFile(const std::string& path, Json* out, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f.is_open()) { if (err) *err = "cannot open '" + path + "'"; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  return Parse(ss.str(), out, err);
}

const Json* Json::Find(const std::string& key) const {
  if (type_ != Type::Object) return nullptr;
  for (const auto& kv : obj_) if (kv.first == key) return &kv.second;
  return nullptr;
}

void Json::Set(const std::string& key, Json v) {
  if (type_ != Type::Object) { *this = MakeObject(); }
  for (auto& kv : obj_) if (kv.first == key) { kv.second = std::move(v); return; }
  obj_.emplace_back(key, std::move(v));
}

static std::string Lower(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  return s;
}

// triton_helpers.cpp:42-67: real bool, else string "true"/"false"/integer.
bool Json::AsBool(bool* v) const {
  switch (type_) {
    case Type::Bool: *v = bool_; return true;
    case Type::Int: *v = int_ != 0; return true;
    case Type::Double: *v = dbl_ != 0.0; return true;
    case Type::String: {
      const std::string t = Lower(str_);
      if (t == "true") { *v = true; return true; }
      if (t == "false") { *v = false; return true; }
      char* e = nullptr;
      errno = 0;
      const long long iv = strtoll(t.c_str(), &e, 10);
      if (e == t.c_str() || errno != 0) return false;
      *v = iv != 0;
      return true;
    }
    default: return false;
  }
}

// triton_helpers.cpp:127-144
bool Json::AsInt(int64_t* v) const {
  switch (type_) {
    case Type::Int: *v = int_; return true;
    case Type::Double: if (std::floor(dbl_) == dbl_) { *v = (int64_t)dbl_; return true; } return false;
    case Type::String: {
      char* e = nullptr;
      errno = 0;
      const long long iv = strtoll(str_.c_str(), &e, 10);
      if (e == str_.c_str() || errno != 0) return false;
      *v = iv;
      return true;
    }
    default: return false;
  }
}

// triton_helpers.cpp:146-163
bool Json::AsUInt(uint64_t* v) const {
  switch (type_) {
    case Type::Int: if (int_ < 0) return false; *v = (uint64_t)int_; return true;
    case Type::Double: if (dbl_ < 0 || std::floor(dbl_) != dbl_) return false; *v = (uint64_t)dbl_; return true;
    case Type::String: {
      char* e = nullptr;
      errno = 0;
      const unsigned long long uv = strtoull(str_.c_str(), &e, 10);
      if (e == str_.c_str() || errno != 0 || str_.find('-') != std::string::npos) return false;
      *v = uv;
      return true;
    }
    default: return false;
  }
}

// triton_helpers.cpp:69-86
bool Json::AsDouble(double* v) const {
  switch (type_) {
    case Type::Int: *v = (double)int_; return true;
    case Type::Double: *v = dbl_; return true;
    case Type::String: {
      char* e = nullptr;
      errno = 0;
      const double d = strtod(str_.c_str(), &e);
      if (e == str_.c_str()) return false;
      *v = d;
      return true;
    }
    default: return false;
  }
}

bool Json::AsString(std::string* v) const {
  if (type_ != Type::String) return false;
  *v = str_;
  return true;
}

static void DumpString(const std::string& s, std::string* o) {
  o->push_back('"');
  for (const char ch : s) {
    const unsigned char c = (unsigned char)ch;
    switch (c) {
      case '"': *o += "\\\""; break;
      case '\\': *o += "\\\\"; break;
      case '\n': *o += "\\n"; break;
      case '\r': *o += "\\r"; break;
      case '\t': *o += "\\t"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); *o += b; }
        else o->push_back(ch);
    }
  }
  o->push_back('"');
}

std::string Json::Dump() const {
  std::string o;
  switch (type_) {
    case Type::Null: o = "null"; break;
    case Type::Bool: o = bool_ ? "true" : "false"; break;
    case Type::Int: o = std::to_string(int_); break;
    case Type::Double: { char b[40]; snprintf(b, sizeof b, "%.17g", dbl_); o = b; break; }
    case Type::String: DumpString(str_, &o); break;
    case Type::Array:
      o.push_back('[');
      for (size_t i = 0; i < arr_.size(); ++i) { if (i) o.push_back(','); o += arr_[i].Dump(); }
      o.push_back(']');
      break;
    case Type::Object:
      o.push_back('{');
      for (size_t i = 0; i < obj_.size(); ++i) {
        if (i) o.push_back(',');
        DumpString(obj_[i].first, &o);
        o.push_back(':');
        o += obj_[i].second.Dump();
      }
      o.push_back('}');
      break;
  }
  return o;
}

}  // namespace hps

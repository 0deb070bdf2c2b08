// Minimal JSON DOM for ps.json / Triton model-config / backend-config messages.
// The reference leans on TritonJson (rapidjson) + boost, neither of which exists in this image
// (SURVEY.md §2.1 #6); this is a from-scratch recursive-descent parser with the same tolerant
// accessors the reference's TritonJsonHelper offers (string-encoded numbers and bools:
// /root/reference/hps_backend/src/triton_helpers.cpp:42-178).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace hps {

class Json {
 public:
  enum class Type { Null, Bool, Int, Double, String, Array, Object };

  Json() = default;
  static bool Parse(const std::string& text, Json* out, std::string* err);
  static bool ParseFile(const std::string& path, Json* out, std::string* err);

  Type type() const { return type_; }
  bool is_null() const { return type_ == Type::Null; }
  bool is_object() const { return type_ == Type::Object; }
  bool is_array() const { return type_ == Type::Array; }
  bool is_string() const { return type_ == Type::String; }
  bool is_number() const { return type_ == Type::Int || type_ == Type::Double; }

  // object access
  const Json* Find(const std::string& key) const;
  const std::vector<std::pair<std::string, Json>>& members() const { return obj_; }
  // array access
  size_t size() const { return type_ == Type::Array ? arr_.size() : obj_.size(); }
  const Json& at(size_t i) const { return arr_[i]; }

  // tolerant scalar accessors: a JSON string holding a number/bool converts.
  bool AsBool(bool* v) const;
  bool AsInt(int64_t* v) const;
  bool AsUInt(uint64_t* v) const;
  bool AsDouble(double* v) const;
  bool AsString(std::string* v) const;  // only real JSON strings

  std::string Dump() const;  // compact serialisation (used by the mock Triton core and logs)

  // builders (mock core / tests)
  static Json MakeObject() { Json j; j.type_ = Type::Object; return j; }
  static Json MakeArray() { Json j; j.type_ = Type::Array; return j; }
  static Json MakeString(const std::string& s) { Json j; j.type_ = Type::String; j.str_ = s; return j; }
  static Json MakeInt(int64_t v) { Json j; j.type_ = Type::Int; j.int_ = v; j.dbl_ = (double)v; return j; }
  static Json MakeDouble(double v) { Json j; j.type_ = Type::Double; j.dbl_ = v; return j; }
  static Json MakeBool(bool v) { Json j; j.type_ = Type::Bool; j.bool_ = v; return j; }
  void Set(const std::string& key, Json v);
  void Append(Json v) { arr_.push_back(std::move(v)); }

 private:
  friend class JsonParser;
  Type type_ = Type::Null;
  bool bool_ = false;
  int64_t int_ = 0;
  double dbl_ = 0.0;
  std::string str_;
  std::vector<Json> arr_;
  std::vector<std::pair<std::string, Json>> obj_;
};

}  // namespace hps

// json.hpp — a small JSON reader for the scene files (objects, arrays, numbers,
// strings, true/false/null).  The reference reads its scenes with play-json
// (jvm/src/main/scala/main.scala:56); this is the host-side counterpart for the
// C++ front-end.
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {

class Json {
 public:
  enum Kind { Null, Bool, Number, String, Array, Object };
  Kind kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // insertion order kept (materials map order)

  static Json parse(const std::string& text);

  bool has(const std::string& key) const {
    for (auto& kv : obj) if (kv.first == key) return true;
    return false;
  }
  const Json& at(const std::string& key) const {
    if (kind != Object) throw std::runtime_error("JSON: expected an object when reading '" + key + "'");
    for (auto& kv : obj) if (kv.first == key) return kv.second;
    throw std::runtime_error("JSON: missing key '" + key + "'");
  }
  const Json& at(size_t i) const {
    if (kind != Array || i >= arr.size()) throw std::runtime_error("JSON: array index out of range");
    return arr[i];
  }
  double number() const {
    if (kind != Number) throw std::runtime_error("JSON: expected a number");
    return num;
  }
  const std::string& string() const {
    if (kind != String) throw std::runtime_error("JSON: expected a string");
    return str;
  }
};

}  // namespace rt

#include "json.hpp"

#include <cctype>
#include <cstdlib>

namespace rt {
namespace {

struct Parser {
  const std::string& s;
  size_t i = 0;
  explicit Parser(const std::string& t) : s(t) {}

  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error(std::string("JSON: ") + what + " at offset " + std::to_string(i));
  }
  void ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
  }
  bool eat(char c) {
    ws();
    if (i < s.size() && s[i] == c) { ++i; return true; }
    return false;
  }
  void expect(char c) {
    if (!eat(c)) fail((std::string("expected '") + c + "'").c_str());
  }
  Json value() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    char c = s[i];
    Json j;
    if (c == '{') {
      ++i;
      j.kind = Json::Object;
      if (eat('}')) return j;
      do {
        ws();
        Json k = value();
        if (k.kind != Json::String) fail("object key is not a string");
        expect(':');
        j.obj.emplace_back(k.str, value());
      } while (eat(','));
      expect('}');
      return j;
    }
    if (c == '[') {
      ++i;
      j.kind = Json::Array;
      if (eat(']')) return j;
      do { j.arr.push_back(value()); } while (eat(','));
      expect(']');
      return j;
    }
    if (c == '"') {
      ++i;
      j.kind = Json::String;
      while (i < s.size() && s[i] != '"') {
        if (s[i] == '\\') {
          if (++i >= s.size()) fail("bad escape");
          switch (s[i]) {
            case 'n': j.str += '\n'; break;
            case 't': j.str += '\t'; break;
            case 'r': j.str += '\r'; break;
            case 'b': j.str += '\b'; break;
            case 'f': j.str += '\f'; break;
            case 'u': {
              if (i + 4 >= s.size()) fail("bad \\u escape");
              unsigned cp = (unsigned)std::strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16);
              i += 4;
              if (cp < 0x80) j.str += (char)cp;
              else if (cp < 0x800) { j.str += (char)(0xC0 | (cp >> 6)); j.str += (char)(0x80 | (cp & 0x3F)); }
              else { j.str += (char)(0xE0 | (cp >> 12)); j.str += (char)(0x80 | ((cp >> 6) & 0x3F)); j.str += (char)(0x80 | (cp & 0x3F)); }
              break;
            }
            default: j.str += s[i];
          }
          ++i;
        } else {
          j.str += s[i++];
        }
      }
      if (i >= s.size()) fail("unterminated string");
      ++i;
      return j;
    }
    if (s.compare(i, 4, "true") == 0) { i += 4; j.kind = Json::Bool; j.b = true; return j; }
    if (s.compare(i, 5, "false") == 0) { i += 5; j.kind = Json::Bool; return j; }
    if (s.compare(i, 4, "null") == 0) { i += 4; return j; }
    if (c == '-' || std::isdigit((unsigned char)c)) {
      char* end = nullptr;
      j.num = std::strtod(s.c_str() + i, &end);
      if (end == s.c_str() + i) fail("bad number");
      i = (size_t)(end - s.c_str());
      j.kind = Json::Number;
      return j;
    }
    fail("unexpected character");
  }
};

}  // namespace

Json Json::parse(const std::string& text) {
  Parser p(text);
  Json j = p.value();
  p.ws();
  if (p.i != text.size()) p.fail("trailing characters");
  return j;
}

}  // namespace rt

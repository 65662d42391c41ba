// json_lite.h -- minimal JSON object reader with the call shape of the reference's
// utils::JsonParser (util/utils.h: Parse / GetInt / GetDouble / GetString / GetObject return 0
// on success, non-zero when the key is absent or has another type), enough for the
// retrieval_param / retrieval_params strings a RetrievalModel receives.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>

namespace utils {

class JsonParser {
 public:
  int Parse(const char *str) {
    vals_.clear();
    if (!str) return -1;
    const char *p = str;
    skip(p);
    if (*p != '{') return -1;
    return parse_object(p) ? 0 : -1;
  }
  int GetInt(const std::string &key, int &value) const {
    auto it = vals_.find(key);
    if (it == vals_.end() || it->second.kind != NUM) return -1;
    value = (int)it->second.num;
    return 0;
  }
  int GetDouble(const std::string &key, double &value) const {
    auto it = vals_.find(key);
    if (it == vals_.end() || it->second.kind != NUM) return -1;
    value = it->second.num;
    return 0;
  }
  int GetString(const std::string &key, std::string &value) const {
    auto it = vals_.find(key);
    if (it == vals_.end() || it->second.kind != STR) return -1;
    value = it->second.str;
    return 0;
  }
  int GetObject(const std::string &key, JsonParser &value) const {
    auto it = vals_.find(key);
    if (it == vals_.end() || it->second.kind != OBJ) return -1;
    return value.Parse(it->second.str.c_str());
  }
  bool Contains(const std::string &key) const { return vals_.count(key) != 0; }

 private:
  enum Kind { NUM, STR, OBJ, OTHER };
  struct Val {
    Kind kind;
    double num;
    std::string str;
  };
  std::map<std::string, Val> vals_;

  static void skip(const char *&p) {
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
  }
  static bool parse_string(const char *&p, std::string &out) {
    if (*p != '"') return false;
    p++;
    out.clear();
    while (*p && *p != '"') {
      if (*p == '\\' && p[1]) {
        p++;
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          default: out += *p;
        }
      } else {
        out += *p;
      }
      p++;
    }
    if (*p != '"') return false;
    p++;
    return true;
  }
  // copies a balanced {...} or [...] span verbatim
  static bool span(const char *&p, std::string &out) {
    const char open = *p, close = open == '{' ? '}' : ']';
    int depth = 0;
    const char *s = p;
    bool in_str = false;
    for (; *p; p++) {
      if (in_str) {
        if (*p == '\\' && p[1]) p++;
        else if (*p == '"') in_str = false;
        continue;
      }
      if (*p == '"') in_str = true;
      else if (*p == open) depth++;
      else if (*p == close && --depth == 0) {
        p++;
        out.assign(s, p - s);
        return true;
      }
    }
    return false;
  }
  bool parse_object(const char *&p) {
    p++;  // {
    skip(p);
    if (*p == '}') return true;
    for (;;) {
      skip(p);
      std::string key;
      if (!parse_string(p, key)) return false;
      skip(p);
      if (*p != ':') return false;
      p++;
      skip(p);
      Val v;
      v.num = 0;
      if (*p == '"') {
        v.kind = STR;
        if (!parse_string(p, v.str)) return false;
      } else if (*p == '{') {
        v.kind = OBJ;
        if (!span(p, v.str)) return false;
      } else if (*p == '[') {
        v.kind = OTHER;
        if (!span(p, v.str)) return false;
      } else if (!strncmp(p, "true", 4)) {
        v.kind = NUM; v.num = 1; p += 4;
      } else if (!strncmp(p, "false", 5)) {
        v.kind = NUM; v.num = 0; p += 5;
      } else if (!strncmp(p, "null", 4)) {
        v.kind = OTHER; p += 4;
      } else {
        char *end = nullptr;
        v.num = strtod(p, &end);
        if (end == p) return false;
        v.kind = NUM;
        p = end;
      }
      vals_[key] = v;
      skip(p);
      if (*p == ',') { p++; continue; }
      if (*p == '}') { p++; return true; }
      return false;
    }
  }
};

}  // namespace utils

// ORACLE / TEST INFRASTRUCTURE ONLY — a minimal stand-in for <pangolin/pangolin.h>.
//
// On the reference's per-frame path Pangolin is used for one thing: MonoSLAM::Init reads
// its configuration with pangolin::ParseVarsFile + pangolin::Var<T>(name, default)
// (monoslam.cpp:1578-1846).  File format: `name = value;` lines, `#` starts a comment,
// later assignments win, unknown names keep the default.
#ifndef SL2_REF_SHIM_PANGOLIN
#define SL2_REF_SHIM_PANGOLIN

#include <cstdlib>
#include <fstream>
#include <map>
#include <string>

namespace pangolin {

inline std::map<std::string, std::string>& shim_vars() {
  static std::map<std::string, std::string> v;
  return v;
}

inline std::string shim_trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

inline void ParseVarsFile(const std::string& path) {
  std::ifstream in(path.c_str());
  std::string line;
  while (std::getline(in, line)) {
    size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string key = shim_trim(line.substr(0, eq));
    std::string val = shim_trim(line.substr(eq + 1));
    if (!val.empty() && val[val.size() - 1] == ';') val = shim_trim(val.substr(0, val.size() - 1));
    if (!key.empty()) shim_vars()[key] = val;
  }
}

template <class T>
struct shim_convert;
template <>
struct shim_convert<double> {
  static double from(const std::string& s) { return std::strtod(s.c_str(), nullptr); }
};
template <>
struct shim_convert<int> {
  // a Var<int> given "195.5" reads 195: stream extraction stops at the '.'
  static int from(const std::string& s) { return (int)std::strtol(s.c_str(), nullptr, 10); }
};
template <>
struct shim_convert<bool> {
  static bool from(const std::string& s) { return !(s == "0" || s == "false" || s.empty()); }
};
template <>
struct shim_convert<std::string> {
  static std::string from(const std::string& s) { return s; }
};

template <class T>
class Var {
 public:
  Var(const std::string& name, const T& def) : v_(def) {
    std::map<std::string, std::string>::const_iterator it = shim_vars().find(name);
    if (it != shim_vars().end()) v_ = shim_convert<T>::from(it->second);
  }
  operator const T&() const { return v_; }
  const T& Get() const { return v_; }
 private:
  T v_;
};

}  // namespace pangolin

#endif  // SL2_REF_SHIM_PANGOLIN

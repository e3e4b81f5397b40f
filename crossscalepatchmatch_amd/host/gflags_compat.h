// gflags_compat.h -- the gflags surface main.cc uses (main.cc:23-34,59): DEFINE_string/int32/bool/double,
// FLAGS_*, gflags::ParseCommandLineFlags with --k=v, --k v, -k=v, --flag / --noflag for booleans.
// gflags is not installed in this image; with -DCSPM_USE_GFLAGS the real library is used.
#pragma once
#ifdef CSPM_USE_GFLAGS
#include <gflags/gflags.h>
#else
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>

namespace gflags {
struct FlagInfo {
  enum Kind { kString, kInt32, kBool, kDouble } kind;
  void *ptr;
  const char *help;
};
inline std::map<std::string, FlagInfo> &registry() {
  static std::map<std::string, FlagInfo> r;
  return r;
}
struct Registrar {
  Registrar(const char *name, FlagInfo::Kind k, void *p, const char *help) { registry()[name] = FlagInfo{k, p, help}; }
};
inline bool set_flag(const FlagInfo &f, const std::string &v) {
  switch (f.kind) {
    case FlagInfo::kString: *static_cast<std::string *>(f.ptr) = v; return true;
    case FlagInfo::kInt32: { char *e; long x = std::strtol(v.c_str(), &e, 0); if (*e || v.empty()) return false; *static_cast<int32_t *>(f.ptr) = (int32_t)x; return true; }
    case FlagInfo::kDouble: { char *e; double x = std::strtod(v.c_str(), &e); if (*e || v.empty()) return false; *static_cast<double *>(f.ptr) = x; return true; }
    case FlagInfo::kBool: {
      if (v == "true" || v == "1" || v == "t" || v == "yes" || v == "y") { *static_cast<bool *>(f.ptr) = true; return true; }
      if (v == "false" || v == "0" || v == "f" || v == "no" || v == "n") { *static_cast<bool *>(f.ptr) = false; return true; }
      return false;
    }
  }
  return false;
}
// returns the index of the first non-flag argument; unknown flags / bad values abort like gflags does
inline uint32_t ParseCommandLineFlags(int *argc, char ***argv, bool /*remove_flags*/) {
  int i = 1;
  for (; i < *argc; ++i) {
    std::string a = (*argv)[i];
    if (a == "--") { ++i; break; }
    if (a.size() < 2 || a[0] != '-') break;
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string name = a, val;
    bool has_val = false;
    size_t eq = a.find('=');
    if (eq != std::string::npos) { name = a.substr(0, eq); val = a.substr(eq + 1); has_val = true; }
    if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
    auto it = registry().find(name);
    if (it == registry().end() && name.compare(0, 2, "no") == 0) {
      auto jt = registry().find(name.substr(2));
      if (jt != registry().end() && jt->second.kind == FlagInfo::kBool && !has_val) { *static_cast<bool *>(jt->second.ptr) = false; continue; }
    }
    if (it == registry().end()) { std::fprintf(stderr, "ERROR: unknown command line flag '%s'\n", name.c_str()); std::exit(1); }
    if (!has_val) {
      if (it->second.kind == FlagInfo::kBool) { *static_cast<bool *>(it->second.ptr) = true; continue; }
      if (i + 1 >= *argc) { std::fprintf(stderr, "ERROR: flag '%s' is missing its argument\n", name.c_str()); std::exit(1); }
      val = (*argv)[++i];
    }
    if (!set_flag(it->second, val)) { std::fprintf(stderr, "ERROR: illegal value '%s' specified for flag '%s'\n", val.c_str(), name.c_str()); std::exit(1); }
  }
  return (uint32_t)i;
}
}  // namespace gflags

#define CSPM_DEFINE_FLAG(type, kind, name, def, help) \
  type FLAGS_##name = def;                             \
  static gflags::Registrar cspm_flag_reg_##name(#name, gflags::FlagInfo::kind, &FLAGS_##name, help)
#define DEFINE_string(name, def, help) CSPM_DEFINE_FLAG(std::string, kString, name, def, help)
#define DEFINE_int32(name, def, help) CSPM_DEFINE_FLAG(int32_t, kInt32, name, def, help)
#define DEFINE_bool(name, def, help) CSPM_DEFINE_FLAG(bool, kBool, name, def, help)
#define DEFINE_double(name, def, help) CSPM_DEFINE_FLAG(double, kDouble, name, def, help)
#endif

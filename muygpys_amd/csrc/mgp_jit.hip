// Run-time specialisation of the wave kernel (hiprtc).
//
// The library carries precompiled instantiations of fused_wave_kernel for the BASELINE shapes and the
// run-time-shape forms; every other (k, R, d) that fits a static instantiation is compiled on first use
// from the same header the library was built from (mgp_fused_wave_kernel.h, shipped in-tree next to the
// .so), about one second per shape, and kept
//   * in the process: one hipModule per (device, shape), under a mutex;
//   * on disk: <lib dir>/jit/*.hsaco (MUYGPYS_HIP_JIT_CACHE overrides), keyed by the shape AND a hash of
//     the kernel sources + compile options, so a rebuilt library never loads a stale object.
//     `python -m muygpys_amd.build` pre-populates it for a family of common shapes (no GPU needed).
// hiprtc is opened with dlopen: without it (or without the sources) the caller falls back to the
// run-time-shape kernels -- slower, never wrong.
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "mgp_args.h"

namespace mgp {

namespace {

struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcAddNameExpression) add_name = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetLoweredName) lowered = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  bool ok = false;
  Rtc() {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return;
#define MGP_RTC_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, #sym))
    MGP_RTC_SYM(create, hiprtcCreateProgram);
    MGP_RTC_SYM(add_name, hiprtcAddNameExpression);
    MGP_RTC_SYM(compile, hiprtcCompileProgram);
    MGP_RTC_SYM(log_size, hiprtcGetProgramLogSize);
    MGP_RTC_SYM(log, hiprtcGetProgramLog);
    MGP_RTC_SYM(lowered, hiprtcGetLoweredName);
    MGP_RTC_SYM(code_size, hiprtcGetCodeSize);
    MGP_RTC_SYM(code, hiprtcGetCode);
    MGP_RTC_SYM(destroy, hiprtcDestroyProgram);
    MGP_RTC_SYM(version, hiprtcVersion);
#undef MGP_RTC_SYM
    ok = create && add_name && compile && log_size && log && lowered && code_size && code && destroy;
  }
};

std::string dir_of_this_library() {
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void*>(&dir_of_this_library), &info) || !info.dli_fname) return "";
  std::string p(info.dli_fname);
  const size_t s = p.rfind('/');
  return s == std::string::npos ? "." : p.substr(0, s);
}

bool read_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  out->clear();
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h;
}

// the options every kernel file of the library is built with (muygpys_amd/build.py) -- part of the key
const char* const kOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
                                "-pragma-unroll-threshold=1000000", "-Wno-pass-failed"};
const char* const kSources[] = {"mgp_fused_wave_kernel.h", "mgp_wave_common.h", "mgp_args.h", "mgp_device.h", "mgp_loocv_tree.h"};

struct Env {
  Rtc rtc;
  std::string src_dir, cache_dir;
  uint64_t src_hash = 0;
  bool sources_ok = false;
  bool trace = false;
  Env() {
    const std::string lib = dir_of_this_library();
    const char* s = getenv("MUYGPYS_HIP_SRC");
    src_dir = s && *s ? s : lib + "/../csrc";
    const char* c = getenv("MUYGPYS_HIP_JIT_CACHE");
    cache_dir = c && *c ? c : lib + "/jit";
    trace = getenv("MGP_TRACE") != nullptr;
    uint64_t h = 1469598103934665603ull;
    sources_ok = true;
    for (const char* name : kSources) {
      std::string text;
      if (!read_file(src_dir + "/" + name, &text)) {
        sources_ok = false;
        break;
      }
      h = fnv1a(text, h);
    }
    for (const char* o : kOptions) h = fnv1a(o, h);
    // ... and the compiler: an object built by another hiprtc is not reused after an upgrade
    int major = 0, minor = 0;
    if (rtc.ok && rtc.version && rtc.version(&major, &minor) == HIPRTC_SUCCESS)
      h = fnv1a("hiprtc " + std::to_string(major) + "." + std::to_string(minor), h);
    src_hash = h;
  }
};

Env& env() {
  static Env e;
  return e;
}

using Key = std::tuple<int, int, int, int, int, int, int, int>;  // es, np, k, R, d, packed, gram, flags (1: gen64, 2: backward)

std::string instantiation(const Key& key) {
  char buf[176];
  snprintf(buf, sizeof buf, "mgp::fused_wave_kernel<%s, %d, %d, %d, %d, true, false, %s, %s, %s, %s>",
           std::get<0>(key) == 4 ? "float" : "double", std::get<1>(key), std::get<2>(key), std::get<3>(key), std::get<4>(key),
           std::get<5>(key) ? "true" : "false", std::get<6>(key) ? "true" : "false", (std::get<7>(key) & 1) ? "true" : "false",
           (std::get<7>(key) & 2) ? "true" : "false");
  return buf;
}

std::string cache_path(const Key& key) {
  char buf[128];
  snprintf(buf, sizeof buf, "/wave_f%d_np%d_k%d_r%d_d%d_p%d_g%d%s%s_%016llx.hsaco", std::get<0>(key) * 8, std::get<1>(key),
           std::get<2>(key), std::get<3>(key), std::get<4>(key), std::get<5>(key), std::get<6>(key), (std::get<7>(key) & 1) ? "_n1" : "",
           (std::get<7>(key) & 2) ? "_b1" : "", (unsigned long long)env().src_hash);
  return env().cache_dir + buf;
}

// file = "MGPJIT1\n<lowered name>\n" + code object
bool load_cached(const Key& key, std::string* name, std::string* code) {
  std::string blob;
  if (!read_file(cache_path(key), &blob)) return false;
  if (blob.compare(0, 8, "MGPJIT1\n") != 0) return false;
  const size_t nl = blob.find('\n', 8);
  if (nl == std::string::npos) return false;
  *name = blob.substr(8, nl - 8);
  *code = blob.substr(nl + 1);
  return !name->empty() && !code->empty();
}

void store_cached(const Key& key, const std::string& name, const std::string& code) {
  mkdir(env().cache_dir.c_str(), 0755);  // code objects: writable by their owner only
  const std::string path = cache_path(key), tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;  // a read-only tree: the object lives in this process only
  const bool ok = fwrite("MGPJIT1\n", 1, 8, f) == 8 && fwrite(name.data(), 1, name.size(), f) == name.size() &&
                  fwrite("\n", 1, 1, f) == 1 && fwrite(code.data(), 1, code.size(), f) == code.size();
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

// VGPRs the register allocator spilled, from the code object's msgpack metadata (".vgpr_spill_count" is a
// fixstr key followed by a small unsigned integer); -1 when the entry is not found
int spilled_vgprs(const std::string& code) {
  static const char key[] = "\xb1.vgpr_spill_count";
  const size_t i = code.find(key, 0, sizeof key - 1);
  if (i == std::string::npos || i + sizeof key - 1 + 3 > code.size()) return -1;
  const unsigned char* p = reinterpret_cast<const unsigned char*>(code.data()) + i + sizeof key - 1;
  if (p[0] < 0x80) return p[0];
  if (p[0] == 0xcc) return p[1];
  if (p[0] == 0xcd) return (p[1] << 8) | p[2];
  return -1;
}

int compile_once(const Key& key, const char* extra_option, std::string* name, std::string* code);

// compile (or fetch from disk) the code object of one instantiation; no GPU needed.  The folded elimination
// (mgp_fused_wave_kernel.h, phase 4F) parks 48-96 registers across a task: where that makes the allocator
// spill, the unfolded kernel is the faster one and is what gets cached.
int ensure_code(const Key& key, std::string* name, std::string* code) {
  if (load_cached(key, name, code)) return MGP_OK;
  int rc = compile_once(key, nullptr, name, code);
  // (the backward instantiations run at two waves per SIMD and may keep a handful of loop invariants in scratch: as built)
  if (rc == MGP_OK && spilled_vgprs(*code) > 0 && !(std::get<7>(key) & 2)) {
    // register-hungry options an instantiation may not afford: the folded elimination (fp32), three waves per SIMD
    // for the dealt-lower-triangle kernels (fp64) -- the first build without spills is what gets cached
    for (const char* opt : {"-DMGP_FOLD=0", "-DMGP_C4_W3=0"}) {
      std::string name2, code2;
      if (compile_once(key, opt, &name2, &code2) == MGP_OK && spilled_vgprs(code2) == 0) {
        if (env().trace) fprintf(stderr, "mgp: %s spills: built with %s\n", instantiation(key).c_str(), opt);
        *name = name2;
        *code = code2;
        break;
      }
    }
  }
  if (rc == MGP_OK) store_cached(key, *name, *code);
  return rc;
}

int compile_once(const Key& key, const char* extra_option, std::string* name, std::string* code) {
  Env& e = env();
  if (!e.rtc.ok || !e.sources_ok) return MGP_EUNSUPPORTED;
  const std::string inst = instantiation(key);
  const std::string src = "#include \"mgp_fused_wave_kernel.h\"\ntemplate __global__ void " + inst +
                          "(mgp::FusedArgs, mgp::WaveGeom);\n";
  hiprtcProgram prog;
  if (e.rtc.create(&prog, src.c_str(), "mgp_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return MGP_EUNSUPPORTED;
  std::vector<const char*> opts(kOptions, kOptions + sizeof kOptions / sizeof *kOptions);
  const std::string inc = "-I" + e.src_dir;
  opts.push_back(inc.c_str());
  if (extra_option) opts.push_back(extra_option);
  int rc = MGP_EUNSUPPORTED;
  if (e.rtc.add_name(prog, inst.c_str()) == HIPRTC_SUCCESS &&
      e.rtc.compile(prog, (int)opts.size(), opts.data()) == HIPRTC_SUCCESS) {
    const char* lowered = nullptr;
    size_t n = 0;
    if (e.rtc.lowered(prog, inst.c_str(), &lowered) == HIPRTC_SUCCESS && lowered &&
        e.rtc.code_size(prog, &n) == HIPRTC_SUCCESS && n > 0) {
      code->resize(n);
      if (e.rtc.code(prog, &(*code)[0]) == HIPRTC_SUCCESS) {
        *name = lowered;
        rc = MGP_OK;
      }
    }
  } else if (e.trace) {
    size_t n = 0;
    e.rtc.log_size(prog, &n);
    std::string log(n, '\0');
    if (n) e.rtc.log(prog, &log[0]);
    fprintf(stderr, "mgp: run-time compile of %s failed:\n%s\n", inst.c_str(), log.c_str());
  }
  e.rtc.destroy(&prog);
  if (e.trace) fprintf(stderr, "mgp: run-time compile %s -> %s\n", inst.c_str(), rc == MGP_OK ? "ok" : "failed");
  return rc;
}

struct Loaded {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  int status = MGP_EUNSUPPORTED;  // remembered: a shape that failed once is not retried on every call
};
std::mutex g_mu;
std::map<std::pair<int, Key>, Loaded> g_loaded;

}  // namespace

// MGP_OK and the kernel of one static instantiation on the current device, or MGP_EUNSUPPORTED.
// allow_compile = false: only what is loaded already or lies in the disk cache (the shapes compiled at build time,
// or by an earlier run) -- a call too short to pay for a compile still gets its specialised kernel then.
int jit_wave_function(int es, int np, int k, int R, int d, bool packed, bool gram, hipFunction_t* fn, bool allow_compile, bool gen64,
                      bool bwd) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MGP_EHIP;
  const Key key{es, np, k, R, d, packed ? 1 : 0, gram ? 1 : 0, (gen64 ? 1 : 0) | (bwd ? 2 : 0)};
  std::lock_guard<std::mutex> lock(g_mu);
  Loaded& l = g_loaded[{dev, key}];
  if (l.fn == nullptr && l.module == nullptr && (l.status == MGP_EUNSUPPORTED || (l.status == -4 && allow_compile))) {
    if (!allow_compile) {
      struct stat st;
      if (stat(cache_path(key).c_str(), &st) != 0) {
        l.status = -4;  // not cached (asked without the right to compile): a later large call may still build it
        return MGP_EUNSUPPORTED;
      }
    }
    l.status = -3;  // tried
    std::string name, code;
    if (ensure_code(key, &name, &code) == MGP_OK && hipModuleLoadData(&l.module, code.data()) == hipSuccess &&
        hipModuleGetFunction(&l.fn, l.module, name.c_str()) == hipSuccess)
      l.status = MGP_OK;
    else
      l.fn = nullptr;
  }
  if (l.status != MGP_OK) return MGP_EUNSUPPORTED;
  *fn = l.fn;
  return MGP_OK;
}

// the hash every cache file name of this build ends in (sources + options + compiler): build.py drops the others
uint64_t jit_source_hash() { return env().src_hash; }

// run-time compiled kernels loaded in this process so far (all devices)
int jit_loaded_count() {
  std::lock_guard<std::mutex> lock(g_mu);
  int n = 0;
  for (const auto& kv : g_loaded) n += kv.second.status == MGP_OK;
  return n;
}

// compile into the disk cache only (build time; no GPU): MGP_OK / MGP_EUNSUPPORTED
int jit_wave_prepare(int es, int np, int k, int R, int d, bool packed, bool gram, bool gen64, bool bwd) {
  std::string name, code;
  return ensure_code(Key{es, np, k, R, d, packed ? 1 : 0, gram ? 1 : 0, (gen64 ? 1 : 0) | (bwd ? 2 : 0)}, &name, &code);
}

// MUYGPYS_HIP_JIT: "0" never; "force" every eligible shape; otherwise (default) eligible shapes from
// MUYGPYS_HIP_JIT_MIN_BATCH neighbourhoods per call on (default 65536: below that a call is too short
// for the specialisation to matter, and the test-sized problems never wait for a compile)
int jit_mode() {
  static const int mode = [] {
    const char* s = getenv("MUYGPYS_HIP_JIT");
    if (s && (!strcmp(s, "0") || !strcmp(s, "off"))) return 0;
    if (s && !strcmp(s, "force")) return 2;
    return 1;
  }();
  return mode;
}
// ... and from MUYGPYS_HIP_JIT_CACHED_MIN_BATCH neighbourhoods on (default 4096) when the shape's kernel is
// already loaded or on disk
int64_t jit_cached_min_batch() {
  static const int64_t n = [] {
    const char* s = getenv("MUYGPYS_HIP_JIT_CACHED_MIN_BATCH");
    return s && *s ? (int64_t)atoll(s) : (int64_t)4096;
  }();
  return n;
}
int64_t jit_min_batch() {
  static const int64_t n = [] {
    const char* s = getenv("MUYGPYS_HIP_JIT_MIN_BATCH");
    return s && *s ? (int64_t)atoll(s) : (int64_t)65536;
  }();
  return n;
}

}  // namespace mgp

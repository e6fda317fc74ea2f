// hamk_build.cpp -- generated source -> gfx950 code objects (host side of libhamk.so, see hamk_host.h): hiprtc
// specialisation of the hand-written device library, the on-disk cache of compiled code objects, reading kernel sizes and
// spilled SGPRs out of the ELF, the two builds per module.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "hamk_host.h"

using namespace hamk_host;

namespace hamk_host {

static const char kDeviceHeader[] =
#include "hamk_device_src.inc"
    ;
static const char kWaveHeader[] =
#include "hamk_wave_src.inc"
    ;
static const char kQuadHeader[] =
#include "hamk_quad_src.inc"
    ;

static const char kSampleSource[] =
#include "hamk_sample_src.inc"
    ;

// ---------------------------------------------------------------------------
// specialisation
// ---------------------------------------------------------------------------
// On-disk cache of compiled code objects, keyed by everything that determines them (generated
// source, both device headers, the option list, the hiprtc version): the same System built twice
// -- another process, another rank of the same job -- costs one compile.
// Location: HAMK_CACHE_DIR, else $XDG_CACHE_HOME/hamk, else $HOME/.cache/hamk; HAMK_CACHE=0 disables
// it.  What is loaded from it runs on the GPU inside this process, so the directory must be a real
// directory (lstat: not a symlink) OWNED BY THE CALLER with no group/other permission bits -- a
// pre-created or world-writable directory disables the cache instead of being trusted -- and every
// entry carries a trailer (magic, payload size, SHA-256 of the full key material, SHA-256 of the
// payload) that is verified before use: a 64-bit file-name collision, a truncated write or a stale
// file yields a recompile, never someone else's kernels.
namespace {
constexpr char kCacheMagic[8] = {'H', 'A', 'M', 'K', 'C', 'O', '0', '2'};
struct CacheTrailer {          // appended to the code object
  char magic[8];
  uint64_t payload_bytes;
  unsigned char key_sha[32];   // source + headers + options + hiprtc version
  unsigned char blob_sha[32];  // the code object itself
};

bool private_dir(const std::string& d) {      // a real directory of ours that nobody else can touch
  struct stat st;
  if (::lstat(d.c_str(), &st) != 0) return false;
  return S_ISDIR(st.st_mode) && st.st_uid == ::getuid() && (st.st_mode & 077) == 0;
}
}  // namespace

static uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
  const unsigned char* p = (const unsigned char*)data;
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

static std::string cache_dir() {
  if (const char* e = std::getenv("HAMK_CACHE")) if (e[0] == '0') return std::string();
  std::string d;
  if (const char* e = std::getenv("HAMK_CACHE_DIR")) d = e;
  else if (const char* x = std::getenv("XDG_CACHE_HOME")) { if (x[0]) { ::mkdir(x, 0700); d = std::string(x) + "/hamk"; } }
  if (d.empty()) {
    const char* home = std::getenv("HOME");
    if (!home || !home[0]) return std::string();           // nowhere private to put it: no cache
    const std::string c = std::string(home) + "/.cache";
    ::mkdir(c.c_str(), 0700);
    d = c + "/hamk";
  }
  ::mkdir(d.c_str(), 0700);                                // EEXIST is fine: what exists is checked next
  if (!private_dir(d)) return std::string();
  return d;
}

struct CacheKey { std::string path; unsigned char sha[32]; };

static CacheKey cache_key(const Variant* s, const std::vector<const char*>& opts, bool cache_on) {
  CacheKey k;
  std::memset(k.sha, 0, sizeof k.sha);
  const std::string dir = cache_on ? cache_dir() : std::string();
  if (dir.empty()) return k;
  int major = 0, minor = 0;
  hiprtcVersion(&major, &minor);
  uint64_t h = 1469598103934665603ull;
  Sha256 sha;
  auto feed = [&](const void* p, size_t n) { h = fnv1a(p, n, h); const uint64_t len = n; sha.update(&len, sizeof len); sha.update(p, n); };
  feed(s->source.data(), s->source.size());
  feed(kDeviceHeader, sizeof kDeviceHeader);                // the headers this variant's source includes
  if (s->mapping == HAMK_MAP_WAVE) feed(kWaveHeader, sizeof kWaveHeader);
  if (s->mapping == HAMK_MAP_QUAD) feed(kQuadHeader, sizeof kQuadHeader);
  for (const char* o : opts) feed(o, std::strlen(o) + 1);
  feed(&major, sizeof major);
  feed(&minor, sizeof minor);
  sha.finish(k.sha);
  char name[64];
  std::snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
  k.path = dir + name;
  return k;
}

// a cache entry is used only if its trailer matches the key and the payload it describes
static bool cache_load(const CacheKey& k, std::vector<char>& code) {
  std::ifstream in(k.path, std::ios::binary);
  if (!in) return false;
  std::vector<char> blob((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (blob.size() < sizeof(CacheTrailer) + 64) return false;
  CacheTrailer t;
  std::memcpy(&t, blob.data() + blob.size() - sizeof t, sizeof t);
  if (std::memcmp(t.magic, kCacheMagic, sizeof t.magic) != 0) return false;
  if (t.payload_bytes != blob.size() - sizeof t) return false;
  if (std::memcmp(t.key_sha, k.sha, 32) != 0) return false;
  unsigned char got[32];
  Sha256 sha; sha.update(blob.data(), (size_t)t.payload_bytes); sha.finish(got);
  if (std::memcmp(got, t.blob_sha, 32) != 0) return false;
  if (std::memcmp(blob.data(), "\177ELF", 4) != 0) return false;
  blob.resize((size_t)t.payload_bytes);
  code.swap(blob);
  return true;
}

static void cache_store(const CacheKey& k, const std::vector<char>& code) {   // publish atomically: write aside, rename
  CacheTrailer t;
  std::memcpy(t.magic, kCacheMagic, sizeof t.magic);
  t.payload_bytes = code.size();
  std::memcpy(t.key_sha, k.sha, 32);
  Sha256 sha; sha.update(code.data(), code.size()); sha.finish(t.blob_sha);
  const std::string tmp = k.path + ".tmp." + std::to_string((long)getpid());
  std::ofstream out(tmp, std::ios::binary);
  if (!out) return;
  out.write(code.data(), (std::streamsize)code.size());
  out.write((const char*)&t, sizeof t);
  out.close();
  if (!out || std::rename(tmp.c_str(), k.path.c_str()) != 0) std::remove(tmp.c_str());
}

int compile_module(Variant* s, bool cache_on, bool no_machine_licm, std::vector<char>& code) {
  hiprtcProgram prog = nullptr;
  const char* hdr_src[] = {kDeviceHeader, kWaveHeader, kQuadHeader};
  const char* hdr_name[] = {"hamk_device.hpp", "hamk_wave.hpp", "hamk_quad.hpp"};
  hiprtcResult r = hiprtcCreateProgram(&prog, s->source.c_str(), "hamk_system.hip", 3, hdr_src, hdr_name);
  if (r != HIPRTC_SUCCESS) return fail(HAMK_ERR_COMPILE, std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r));
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast",
                                   "-fno-honor-nans", "-fno-signed-zeros"};
  if (s->desc.wave || s->desc.mapping == HAMK_MAP_QUAD) {
    // CodeGenPrepare's address sinking is quadratic in the thousands of LDS accesses of the
    // straight-line wave kernels (chain32: 170 s of a 200 s build); it is an optimisation pass only
    opts.push_back("-mllvm");
    opts.push_back("-disable-cgp");
  }
  if (s->desc.mapping == HAMK_MAP_QUAD) {
    // the factorisation is written as loops over literal bounds that MUST unroll completely (rows of K are registers, not
    // memory): ~10^4 FMAs at n = 32, beyond the default cap on `#pragma unroll` (16 K instructions of estimated size) --
    // with the cap in force the inner loops stay rolled and K is indexed dynamically, i.e. lives in scratch
    opts.push_back("-mllvm");
    opts.push_back("-pragma-unroll-threshold=4194304");
  }
  if (no_machine_licm) {
    opts.push_back("-mllvm");
    opts.push_back("-disable-machine-licm");
  }
  std::string extra;                                   // experiments: HAMK_HIPRTC_FLAGS="-mllvm -foo ..."
  std::vector<std::string> extra_tok;
  if (const char* e = test_env("HAMK_HIPRTC_FLAGS")) {
    extra = e;
    size_t pos = 0;
    while (pos < extra.size()) {
      size_t sp = extra.find(' ', pos);
      if (sp == std::string::npos) sp = extra.size();
      if (sp > pos) extra_tok.push_back(extra.substr(pos, sp - pos));
      pos = sp + 1;
    }
    for (auto& t : extra_tok) opts.push_back(t.c_str());
  }
  const CacheKey ckey = cache_key(s, opts, cache_on);
  if (!ckey.path.empty() && cache_load(ckey, code)) {
    s->build_log = "cache hit: " + ckey.path;
    hiprtcDestroyProgram(&prog);
    return HAMK_OK;
  }
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  size_t logsz = 0;
  hiprtcGetProgramLogSize(prog, &logsz);
  if (logsz > 1) {
    s->build_log.resize(logsz);
    hiprtcGetProgramLog(prog, &s->build_log[0]);
  }
  if (r != HIPRTC_SUCCESS) {
    std::string msg = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + s->build_log;
    hiprtcDestroyProgram(&prog);
    return fail(HAMK_ERR_COMPILE, msg);
  }
  size_t sz = 0;
  hiprtcGetCodeSize(prog, &sz);
  code.resize(sz);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (!ckey.path.empty()) cache_store(ckey, code);
  return HAMK_OK;
}

// Size in bytes of one kernel's machine code, read from the code object's ELF symbol table
// (0 if not found).  Used to keep every kernel well inside the +-128 KiB reach of a SOPP
// branch: beyond it the compiler must relax branches through s_setpc with spare SGPRs, and
// the fully unrolled adaptive stepper of a large system was observed to misbehave there
// (MI355X, ROCm 7.2: wrong sub-step counts on the 27-opcode test system).
// name == nullptr: returns the NUMBER of function symbols instead (8 kernels; more means a
// device function was not inlined and is reached through a real call).
size_t kernel_code_bytes(const std::vector<char>& elf, const char* name) {
  size_t nfunc = 0;
  struct Ehdr { unsigned char ident[16]; uint16_t type, machine; uint32_t version; uint64_t entry, phoff, shoff;
                uint32_t flags; uint16_t ehsize, phentsize, phnum, shentsize, shnum, shstrndx; };
  struct Shdr { uint32_t name, type; uint64_t flags, addr, offset, size; uint32_t link, info; uint64_t addralign, entsize; };
  struct Sym { uint32_t name; unsigned char info, other; uint16_t shndx; uint64_t value, size; };
  if (elf.size() < sizeof(Ehdr) || std::memcmp(elf.data(), "\177ELF", 4) != 0) return 0;
  Ehdr eh; std::memcpy(&eh, elf.data(), sizeof eh);
  if (eh.shoff == 0 || eh.shentsize != sizeof(Shdr)) return 0;
  for (unsigned i = 0; i < eh.shnum; ++i) {
    Shdr sh; std::memcpy(&sh, elf.data() + eh.shoff + (size_t)i * sizeof(Shdr), sizeof sh);
    if (sh.type != 2 /* SHT_SYMTAB */ || sh.entsize != sizeof(Sym)) continue;
    Shdr str; std::memcpy(&str, elf.data() + eh.shoff + (size_t)sh.link * sizeof(Shdr), sizeof str);
    for (uint64_t k = 0; k < sh.size / sizeof(Sym); ++k) {
      Sym sy; std::memcpy(&sy, elf.data() + sh.offset + k * sizeof(Sym), sizeof sy);
      if ((sy.info & 0xf) != 2 /* STT_FUNC */) continue;
      ++nfunc;
      const char* nm = elf.data() + str.offset + sy.name;
      if (name && std::strcmp(nm, name) == 0) return (size_t)sy.size;
    }
  }
  return name ? 0 : nfunc;
}


// An unsigned entry of one kernel's metadata map in the code object's note (msgpack; within a kernel's map the keys are
// sorted, so ".name" precedes ".sgpr_spill_count" / ".vgpr_spill_count" and the argument maps' ".name" entries come before
// them).  key: the msgpack fixstr / str8 encoding of the key, e.g. "\xb1.sgpr_spill_count".  -1 if not found.
static int metadata_count(const std::vector<char>& elf, const char* kernel, const char* key, size_t lk) {
  static const char kName[] = "\xa5.name";
  const size_t ln = sizeof kName - 1;
  std::string last;
  for (size_t i = 0; i + lk + 5 < elf.size(); ++i) {
    if (elf[i] == kName[0] && std::memcmp(&elf[i], kName, ln) == 0) {
      const unsigned char b = (unsigned char)elf[i + ln];
      size_t len = 0, at = 0;
      if ((b & 0xe0) == 0xa0) { len = b & 0x1f; at = i + ln + 1; }
      else if (b == 0xd9) { len = (unsigned char)elf[i + ln + 1]; at = i + ln + 2; }
      else continue;
      if (at + len <= elf.size()) last.assign(&elf[at], len);
    } else if (elf[i] == key[0] && std::memcmp(&elf[i], key, lk) == 0) {
      const unsigned char* v = (const unsigned char*)&elf[i + lk];
      long n = -1;
      if (v[0] < 0x80) n = v[0];
      else if (v[0] == 0xcc) n = v[1];
      else if (v[0] == 0xcd) n = (v[1] << 8) | v[2];
      else if (v[0] == 0xce) n = ((long)v[1] << 24) | (v[2] << 16) | (v[3] << 8) | v[4];
      if (last == kernel) return (int)n;
    }
  }
  return -1;
}
// SGPRs / VGPRs a kernel spills
int sgpr_spill_count(const std::vector<char>& elf, const char* kernel) {
  static const char k[] = "\xb1.sgpr_spill_count";
  return metadata_count(elf, kernel, k, sizeof k - 1);
}
int vgpr_spill_count(const std::vector<char>& elf, const char* kernel) {
  static const char k[] = "\xb1.vgpr_spill_count";
  return metadata_count(elf, kernel, k, sizeof k - 1);
}

static void describe_build(Variant* s);

// Build the code object(s) of s->source.  Kernels that spill SGPRs under the default options are
// taken from a second build without MachineLICM when that build spills fewer: the hoisting of the
// 64-bit literal constants out of the stepping loops is what overflows the 102 SGPRs (each fp64
// literal is an SGPR pair on gfx9), re-materialising them in place costs a few SALU moves, and the
// one kernel found giving run-to-run different results is correct again
// without its 101 spilled SGPRs.  Spill-free kernels keep the default build (the headline RK4
// kernel is 3 % faster with the hoisting).  force (hamk_options::build / HAMK_NOLICM): 0 the default build only,
// 1 the build without MachineLICM only, -1 per kernel.
int build_code(Variant* s, bool cache_on, int force) {
  s->code2.clear();
  for (bool& u : s->use2) u = false;
  int rc = compile_module(s, cache_on, false, s->code);
  if (rc != HAMK_OK) return rc;
  if (force == 0) { describe_build(s); return HAMK_OK; }
  int spills[K__COUNT];
  bool any = false;
  for (int k = 0; k < K__COUNT; ++k) { spills[k] = sgpr_spill_count(s->code, kKernelNames[k]); any = any || spills[k] > 0; }
  if (!any && force != 1) { describe_build(s); return HAMK_OK; }
  std::vector<char> alt;
  rc = compile_module(s, cache_on, true, alt);
  if (rc != HAMK_OK) return rc;
  bool used = false;
  for (int k = 0; k < K__COUNT; ++k) {
    const int sp2 = sgpr_spill_count(alt, kKernelNames[k]);
    s->use2[k] = force == 1 || (spills[k] > 0 && sp2 >= 0 && sp2 < spills[k]);
    used = used || s->use2[k];
  }
  if (used) s->code2.swap(alt);
  describe_build(s);
  return HAMK_OK;
}

static void describe_build(Variant* s) {
  std::string t;
  for (int k = 0; k < K__COUNT; ++k) {
    const std::vector<char>& c = s->use2[k] ? s->code2 : s->code;
    char line[160];
    std::snprintf(line, sizeof line, "%s build=%s bytes=%zu sgpr_spills=%d vgpr_spills=%d\n", kKernelNames[k],
                  s->use2[k] ? "no-machine-licm" : "default", kernel_code_bytes(c, kKernelNames[k]),
                  sgpr_spill_count(c, kKernelNames[k]), vgpr_spill_count(c, kKernelNames[k]));
    t += line;
  }
  s->build_info = t;
}

size_t chosen_kernel_bytes(const Variant* s, int k) {
  return kernel_code_bytes(s->use2[k] ? s->code2 : s->code, kKernelNames[k]);
}



// ---- initial conditions on the device ----------------------------------------------------------------
namespace { struct HamkBoxes { double q_lo[64], q_hi[64], qd_lo[64], qd_hi[64]; }; }     // = hamk_sample.hpp

int sample_code(bool cache_on, const std::vector<char>** out) {     // the sampler's code object: compiled once per process
  static std::mutex mu;
  static std::vector<char> code;
  std::lock_guard<std::mutex> lock(mu);
  if (code.empty()) {
    Variant v;
    v.mapping = HAMK_MAP_LANE;
    v.source = kSampleSource;
    TRY0(compile_module(&v, cache_on, false, code));
  }
  *out = &code;
  return HAMK_OK;
}

}  // namespace hamk_host

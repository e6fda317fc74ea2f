// hamk_api.cpp -- C ABI of libhamk.so (include/hamk.h): system construction
// (tape -> hiprtc-specialised gfx950 module) and ensemble launches.
//
// Host-side counterpart of the reference's `System` record and its callers
// (Hamilton.hs:160-169, :201-254, :262-462): there a `System` is a bundle of
// Haskell closures re-entering `ad` on every call; here it is an opaque handle
// owning one compiled code object whose kernels evaluate the whole ensemble.
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

#include "hamk_internal.h"

using namespace hamk_host;

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

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(HAMK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

#define TRY0(expr) do { int rc0_ = (expr); if (rc0_ != HAMK_OK) return rc0_; } while (0)

enum KernelId { K_RK4, K_HAMEQS, K_COORDS, K_TO_PHASE, K_FROM_PHASE, K_OBSERVE, K_OBSERVE_CFG, K_RKF45, K_SCRIBBLE, K__COUNT };
static const char* kKernelNames[K__COUNT] = {"hamk_rk4_steps_k", "hamk_hameqs_k",  "hamk_coords_k",         "hamk_to_phase_k",
                                             "hamk_from_phase_k", "hamk_observe_k", "hamk_observe_config_k", "hamk_rkf45_k",
                                             "hamk_scribble_k"};

// One specialisation of the device library for a system: which lanes serve a trajectory (hamk.h HAMK_MAP_*) and the
// choices that go with it.  A handle builds the one its options name -- or, with mapping = HAMK_AUTO, the one a large
// ensemble uses -- when it is created, and the others the first time a launch asks for them.
struct Variant {
  int mapping = HAMK_MAP_LANE;
  SystemDesc desc;
  std::string source;
  std::vector<char> code;      // gfx950 code object (default options)
  std::vector<char> code2;     // the same source built without MachineLICM; empty unless some kernel is taken from it
  bool use2[K__COUNT] = {};    // kernel k comes from code2 (it spills no / fewer SGPRs there)
  std::string build_log;
  std::string build_info;      // per kernel: which build it comes from, bytes, spilled SGPRs
  int generation = 0;          // bumped whenever `code` is rebuilt (the self-check's recovery path)
  int self_check_rebuilds = 0;
  bool forced_rk4_body = false, forced_rkf_body = false;
  bool has[K__COUNT] = {};     // the kernels this module provides (the quad module: four of the eight)
};
constexpr int kMaxMap = 4;     // HAMK_MAP_* ids are 1..3

// The modules of one Variant on one device.
struct DevModule {
  hipModule_t module = nullptr;
  hipModule_t module2 = nullptr;
  hipFunction_t fn[K__COUNT] = {};
  bool self_checked = false;
  int code_generation = -1;     // Variant::generation the loaded modules were built from
  void unload() {
    if (module) { hipModuleUnload(module); module = nullptr; }
    if (module2) { hipModuleUnload(module2); module2 = nullptr; }
  }
};

// What a handle owns on ONE device.  A handle used from several devices (one process driving every
// GPU of a node, or a torch program whose tensors live on cuda:1 while cuda:0 is current) keeps one
// of these per device: modules stay loaded and staging buffers stay allocated when the calls
// alternate between devices.
struct DevState {
  int device = -1;
  DevModule mod[kMaxMap];
  // the stream this handle launches on ON THIS DEVICE (hamk_set_stream binds it to the device that is current
  // at the time: a stream belongs to one device, and a handle may be used from several)
  hipStream_t stream = nullptr;
  // grow-only device staging for HAMK_MEM_HOST calls (slot i serves the i-th staged array of a
  // call): the reference's own usage pattern is one small stepHam per frame (Examples.hs:429),
  // where a hipMalloc/hipFree pair per array per call would dominate
  std::vector<void*> stage_buf;
  std::vector<size_t> stage_cap;
  // pinned, device-mapped arena for SMALL host-pointer calls (the reference's one-trajectory
  // stepHam per frame): the kernel reads and writes host memory directly over PCIe -- a launch
  // and a stream synchronisation per call, no hipMemcpy at all.  The block is allocated COHERENT
  // (fine-grained, uncached on the device): the CPU rewrites it before every call, and
  // hipHostMallocMapped alone gives non-coherent memory whose lines the device may keep in L2
  // across launches.
  char* pin = nullptr;          // host address
  char* pin_dev = nullptr;      // the same block as the device sees it
  bool pin_failed = false;
  // hamk_sample_batch's kernel (hamk_sample.hpp; one module per process, loaded per device on first use)
  hipModule_t sample_module = nullptr;
  hipFunction_t sample_fn = nullptr;
  // small device scratch for evolveHam's time grid
  double* d_ts = nullptr;
  size_t d_ts_cap = 0;
  std::vector<double> h_ts;
  void release() {
    for (DevModule& m : mod) m.unload();
    if (sample_module) { hipModuleUnload(sample_module); sample_module = nullptr; sample_fn = nullptr; }
    if (d_ts) { hipFree(d_ts); d_ts = nullptr; d_ts_cap = 0; }
    for (void*& b : stage_buf) if (b) { hipFree(b); b = nullptr; }
    stage_cap.assign(stage_cap.size(), 0);
    if (pin) { hipHostFree(pin); pin = pin_dev = nullptr; }
  }
};

struct hamk_system {
  SystemDesc base;             // m, n, inertia, tapes: what every specialisation shares
  hamk_options opt;            // as the caller gave them (HAMK_AUTO where the choice is the library's)
  Variant* var[kMaxMap] = {};  // by mapping id; built on demand
  Variant* curv = nullptr;     // the specialisation the call in progress uses
  Variant* info = nullptr;     // the one the introspection entry points describe (hamk_system_describe_batch)
  // lazily bound to the calling thread's current device, one DevState per device ever used
  std::vector<DevState*> devs;
  DevState* cur = nullptr;
  int gsl_api = 2;             // which binding of hmatrix-gsl's gsl-ode.c stepHam/evolveHam follow (hamk.h)
  int max_substeps = 1 << 24;
  bool self_check_on = true, cache_on = true;
  int quad_eligible = -1;      // -1: not analysed yet
  int64_t ensemble_size = 0;   // hamk_options::ensemble_size: AUTO picks the mapping for THIS size instead of a launch's own B
  DevModule& mod() { return cur->mod[curv->mapping]; }
};

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
struct Sha256 {
  uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  unsigned char buf[64];
  size_t fill = 0;
  uint64_t total = 0;
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const unsigned char* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
        0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
        0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
        0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
        0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
        0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
      const uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void* data, size_t n) {
    const unsigned char* p = (const unsigned char*)data;
    total += n;
    while (n) {
      const size_t k = std::min(n, sizeof buf - fill);
      std::memcpy(buf + fill, p, k);
      fill += k; p += k; n -= k;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(unsigned char out[32]) {
    const uint64_t bits = total * 8;
    const unsigned char one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    unsigned char len[8];
    for (int i = 0; i < 8; ++i) len[i] = (unsigned char)(bits >> (56 - 8 * i));
    update(len, 8);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (unsigned char)(h[i] >> 24); out[4 * i + 1] = (unsigned char)(h[i] >> 16); out[4 * i + 2] = (unsigned char)(h[i] >> 8); out[4 * i + 3] = (unsigned char)h[i]; }
  }
};

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

static int compile_module(Variant* s, bool cache_on, bool no_machine_licm, std::vector<char>& code) {
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
  if (const char* e = std::getenv("HAMK_HIPRTC_FLAGS")) {
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
static size_t kernel_code_bytes(const std::vector<char>& elf, const char* name) {
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


// SGPRs a kernel spills, from the code object's metadata note (msgpack; within a kernel's map the
// keys are sorted, so ".name" precedes ".sgpr_spill_count" and the argument maps' ".name" entries
// come before both).  -1 if not found.
static int sgpr_spill_count(const std::vector<char>& elf, const char* kernel) {
  static const char kName[] = "\xa5.name", kSpill[] = "\xb1.sgpr_spill_count";
  const size_t ln = sizeof kName - 1, ls = sizeof kSpill - 1;
  std::string last;
  for (size_t i = 0; i + ls + 5 < elf.size(); ++i) {
    if (elf[i] == kName[0] && std::memcmp(&elf[i], kName, ln) == 0) {
      const unsigned char b = (unsigned char)elf[i + ln];
      size_t len = 0, at = 0;
      if ((b & 0xe0) == 0xa0) { len = b & 0x1f; at = i + ln + 1; }
      else if (b == 0xd9) { len = (unsigned char)elf[i + ln + 1]; at = i + ln + 2; }
      else continue;
      if (at + len <= elf.size()) last.assign(&elf[at], len);
    } else if (elf[i] == kSpill[0] && std::memcmp(&elf[i], kSpill, ls) == 0) {
      const unsigned char* v = (const unsigned char*)&elf[i + ls];
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

static void describe_build(Variant* s);

// Build the code object(s) of s->source.  Kernels that spill SGPRs under the default options are
// taken from a second build without MachineLICM when that build spills fewer: the hoisting of the
// 64-bit literal constants out of the stepping loops is what overflows the 102 SGPRs (each fp64
// literal is an SGPR pair on gfx9), re-materialising them in place costs a few SALU moves, and the
// one kernel found giving run-to-run different results is correct again
// without its 101 spilled SGPRs.  Spill-free kernels keep the default build (the headline RK4
// kernel is 3 % faster with the hoisting).  force (hamk_options::build / HAMK_NOLICM): 0 the default build only,
// 1 the build without MachineLICM only, -1 per kernel.
static int build_code(Variant* s, bool cache_on, int force) {
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
    std::snprintf(line, sizeof line, "%s build=%s bytes=%zu sgpr_spills=%d\n", kKernelNames[k],
                  s->use2[k] ? "no-machine-licm" : "default", kernel_code_bytes(c, kKernelNames[k]),
                  sgpr_spill_count(c, kKernelNames[k]));
    t += line;
  }
  s->build_info = t;
}

static size_t chosen_kernel_bytes(const Variant* s, int k) {
  return kernel_code_bytes(s->use2[k] ? s->code2 : s->code, kKernelNames[k]);
}

static int build_force(const hamk_system* s) {                      // hamk_options::build, else HAMK_NOLICM (tests), else per kernel
  if (s->opt.build == HAMK_BUILD_DEFAULT) return 0;
  if (s->opt.build == HAMK_BUILD_NOLICM) return 1;
  if (const char* e = std::getenv("HAMK_NOLICM")) return (e[0] == '1') ? 1 : (e[0] == '0' ? 0 : -1);
  return -1;
}

static int launch(hamk_system* s, KernelId k, int64_t B, void** args);
// the flags argument of hamk_rkf45_k (hamk_device.hpp rkf45_body)
static int rkf_flags(int row0, int inplace, int gsl_api) { return (row0 & 1) | ((inplace & 3) << 8) | ((gsl_api & 3) << 16); }
static int load_modules(hamk_system* s);

// ---------------------------------------------------------------------------
// First-use self-check of the stepping kernels against the (small, separately compiled) hamEqs
// kernel: one RK4 step of the fused kernel must equal four hamEqs launches combined on the host,
// one accepted RKF45 sub-step must equal its six stage evaluations combined on the host.  A JIT
// product cannot take the code generator's word for it: on this toolchain one large unrolled
// stepping kernel was observed to be silently wrong.  On a mismatch the
// module is rebuilt once with the stage-loop bodies; if that does not help, the system is refused.
// HAMK_SELFCHECK=0 skips it.
// ---------------------------------------------------------------------------
static const double kRefEpsilon = 1.49012e-08;   // Hamilton.hs:448
static thread_local std::string g_selfcheck_detail;
static int self_check_once(hamk_system* s, bool* rk4_ok, bool* rkf_ok) {
  g_selfcheck_detail.clear();
  const int n = s->base.n;
  const int64_t B = 64;
  const size_t cnt = (size_t)n * B;
  std::vector<double> q(cnt), p(cnt), k(2 * cnt), acc(2 * cnt), yt(2 * cnt);
  for (int j = 0; j < n; ++j)
    for (int64_t i = 0; i < B; ++i) {
      q[(size_t)j * B + i] = 0.31 + 0.07 * j + 0.011 * (double)i;
      p[(size_t)j * B + i] = 0.23 - 0.05 * j + 0.007 * (double)i;
    }
  // device scratch of the check, released on every path out of this function
  struct Scratch {
    void* p[5] = {};
    ~Scratch() { for (void* x : p) if (x) hipFree(x); }
  } scratch;
  HIP_TRY(hipMalloc(&scratch.p[0], cnt * 8)); HIP_TRY(hipMalloc(&scratch.p[1], cnt * 8));
  HIP_TRY(hipMalloc(&scratch.p[2], cnt * 8)); HIP_TRY(hipMalloc(&scratch.p[3], cnt * 8));
  HIP_TRY(hipMalloc(&scratch.p[4], B * 4));
  double *d_q = (double*)scratch.p[0], *d_p = (double*)scratch.p[1], *d_dq = (double*)scratch.p[2], *d_dp = (double*)scratch.p[3];
  int32_t* d_st = (int32_t*)scratch.p[4];
  long long b = B;
  bool flagged = false;
  auto rhs = [&](const std::vector<double>& y, std::vector<double>& out) -> int {   // out = hamEqs(y), y = [q; p]
    HIP_TRY(hipMemcpy(d_q, y.data(), cnt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p, y.data() + cnt, cnt * 8, hipMemcpyHostToDevice));
    const double *cq = d_q, *cp = d_p; int32_t* st = d_st;
    void* args[] = {&cq, &cp, &d_dq, &d_dp, &b, &st};
    int rc = launch(s, K_HAMEQS, B, args);
    if (rc != HAMK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->cur->stream));
    HIP_TRY(hipMemcpy(out.data(), d_dq, cnt * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out.data() + cnt, d_dp, cnt * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> hst((size_t)B);
    HIP_TRY(hipMemcpy(hst.data(), d_st, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int32_t v : hst) if (v != 0) flagged = true;    // singular / non-finite at the test points: cannot judge
    return HAMK_OK;
  };
  auto close_enough = [&](const std::vector<double>& a, const std::vector<double>& ref, bool* usable, const char* what) {
    double worst = 0.0; *usable = true;
    for (size_t i = 0; i < 2 * cnt; ++i) {
      if (!std::isfinite(ref[i])) { *usable = false; return true; }     // test points outside the system's domain: cannot judge
      const double e = std::fabs(a[i] - ref[i]) / std::fmax(1.0, std::fabs(ref[i]));
      if (!(e <= worst)) worst = e;
    }
    if (!(worst <= 1e-9)) {
      char msg[160];
      std::snprintf(msg, sizeof msg, "%s: worst deviation %.3g from the same step built from hamEqs launches", what, worst);
      g_selfcheck_detail = msg;
    }
    return worst <= 1e-9;
  };
  std::vector<double> y0(2 * cnt), got(2 * cnt), ref(2 * cnt);
  std::copy(q.begin(), q.end(), y0.begin()); std::copy(p.begin(), p.end(), y0.begin() + cnt);
  int rc = HAMK_OK;
  bool usable = true;
  // ---- RK4: one step of dt -------------------------------------------------------------------
  // dt = 1e-3 unless the right-hand side at the test points is large (a 64-link chain with these
  // momenta turns at 1e4 rad/s): the two ways of computing one step agree to roundoff only while the
  // step is a small perturbation of the state, so dt keeps the largest increment at 0.05.
  double dt = 1e-3;
  {
    rc = rhs(y0, k);
    double kmax = 0.0;
    for (size_t i = 0; i < 2 * cnt; ++i) if (std::isfinite(k[i]) && std::fabs(k[i]) > kmax) kmax = std::fabs(k[i]);
    if (kmax * dt > 0.05) dt = 0.05 / kmax;
  }
  if (rc == HAMK_OK) {
    const double a[4] = {0.0, 0.5 * dt, 0.5 * dt, dt}, w[4] = {dt / 6, dt / 3, dt / 3, dt / 6};
    ref = y0;
    std::fill(k.begin(), k.end(), 0.0);
    for (int sg = 0; sg < 4 && rc == HAMK_OK; ++sg) {
      for (size_t i = 0; i < 2 * cnt; ++i) yt[i] = y0[i] + a[sg] * k[i];
      rc = rhs(yt, k);
      for (size_t i = 0; i < 2 * cnt; ++i) ref[i] += w[sg] * k[i];
    }
    if (rc == HAMK_OK) {
      hipMemcpy(d_q, q.data(), cnt * 8, hipMemcpyHostToDevice); hipMemcpy(d_p, p.data(), cnt * 8, hipMemcpyHostToDevice);
      double ddt = dt, no_drift = 0.0; int ns = 1; int32_t* st = d_st;
      void* args[] = {&d_q, &d_p, &b, &ddt, &ns, &no_drift, &st};
      rc = launch(s, K_RK4, B, args);
      if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RK4 kernel failed");
      hipMemcpy(got.data(), d_q, cnt * 8, hipMemcpyDeviceToHost); hipMemcpy(got.data() + cnt, d_p, cnt * 8, hipMemcpyDeviceToHost);
      *rk4_ok = close_enough(got, ref, &usable, "one RK4 step");
    }
  }
  // ---- RKF45: one accepted sub-step (h = dt, huge tolerances, t: 0 -> dt) ---------------------
  const bool has_rkf = s->curv->has[K_RKF45];              // (the quad module leaves the adaptive stepper to the wave module)
  if (rc == HAMK_OK && usable && has_rkf) {
    static const double A[5][5] = {{1.0 / 4, 0, 0, 0, 0},
                                   {3.0 / 32, 9.0 / 32, 0, 0, 0},
                                   {1932.0 / 2197, -7200.0 / 2197, 7296.0 / 2197, 0, 0},
                                   {8341.0 / 4104, -32832.0 / 4104, 29440.0 / 4104, -845.0 / 4104, 0},
                                   {-6080.0 / 20520, 41040.0 / 20520, -28352.0 / 20520, 9295.0 / 20520, -5643.0 / 20520}};
    static const double C[6] = {902880.0 / 7618050, 0, 3953664.0 / 7618050, 3855735.0 / 7618050, -1371249.0 / 7618050, 277020.0 / 7618050};
    std::vector<std::vector<double>> ks(6, std::vector<double>(2 * cnt));
    rc = rhs(y0, ks[0]);
    for (int sg = 1; sg < 6 && rc == HAMK_OK; ++sg) {
      for (size_t i = 0; i < 2 * cnt; ++i) {
        double t = 0.0;
        for (int j2 = 0; j2 < sg; ++j2) t += A[sg - 1][j2] * ks[j2][i];
        yt[i] = y0[i] + dt * t;
      }
      rc = rhs(yt, ks[sg]);
    }
    if (rc == HAMK_OK) {
      for (size_t i = 0; i < 2 * cnt; ++i) {
        double t = 0.0;
        for (int j2 = 0; j2 < 6; ++j2) t += C[j2] * ks[j2][i];
        ref[i] = y0[i] + dt * t;
      }
      hipMemcpy(d_q, q.data(), cnt * 8, hipMemcpyHostToDevice); hipMemcpy(d_p, p.data(), cnt * 8, hipMemcpyHostToDevice);
      double h0 = dt, ea = 1e30, er = 1e30, t0 = 0.0, t1 = dt;
      int nt = 2, flags = rkf_flags(1, 1, s->gsl_api), max_sub = 8;
      const double *cq = d_q, *cp = d_p, *cts = nullptr; int32_t* st = d_st; int32_t* ns = nullptr;
      int ncalls = 1, it_every = 0;
      void* args[] = {&cq, &cp, &d_q, &d_p, &b, &nt, &cts, &t0, &t1, &h0, &ea, &er, &flags, &max_sub, &st, &ns, &ncalls, &it_every};
      rc = launch(s, K_RKF45, B, args);
      if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RKF45 kernel failed");
      hipMemcpy(got.data(), d_q, cnt * 8, hipMemcpyDeviceToHost); hipMemcpy(got.data() + cnt, d_p, cnt * 8, hipMemcpyDeviceToHost);
      *rkf_ok = close_enough(got, ref, &usable, "one accepted RKF45 sub-step");
    }
  }
  if (!usable || flagged) { *rk4_ok = true; *rkf_ok = true; }
  // ---- adaptive stepper end to end ---------------------------------------------------------------
  // stepHam(T) with the reference's tolerances and step-size control on 256 trajectories, TWICE:
  // the two runs must agree bit for bit (lanes are independent: a difference is a broken kernel, and
  // exactly that was seen once -- an unrolled RKF45 body whose results changed from run to run),
  // and both must agree with 64 fixed RK4 steps of T/64 (the kernel checked above) to well within
  // what the controller's tolerance allows.
  if (rc == HAMK_OK && *rk4_ok && *rkf_ok && usable && !flagged) {
    const int64_t B3 = 4096;                                   // 64 wavefronts: many divergence patterns
    const size_t c3 = (size_t)n * B3;
    const double T = 0.02;
    std::vector<double> q3(c3), p3(c3), ref3(2 * c3), run[2] = {std::vector<double>(2 * c3), std::vector<double>(2 * c3)};
    std::vector<int32_t> st_ref((size_t)B3), st_run[2] = {std::vector<int32_t>((size_t)B3), std::vector<int32_t>((size_t)B3)},
        ns_run[2] = {std::vector<int32_t>((size_t)B3), std::vector<int32_t>((size_t)B3)};
    for (int j = 0; j < n; ++j)
      for (int64_t i = 0; i < B3; ++i) {
        const double u = std::fmod(0.6180339887498949 * (double)(i + 1) + 0.37 * j, 1.0);     // low-discrepancy in [0, 1)
        const double w = std::fmod(0.7548776662466927 * (double)(i + 1) + 0.19 * j, 1.0);
        q3[(size_t)j * B3 + i] = 0.31 + 0.07 * j + 0.29 * u;
        p3[(size_t)j * B3 + i] = 0.23 - 0.05 * j + 0.29 * w;
      }
    double *e_q = nullptr, *e_p = nullptr; int32_t *e_st = nullptr, *e_ns = nullptr;
    bool alloc_ok = hipMalloc((void**)&e_q, c3 * 8) == hipSuccess && hipMalloc((void**)&e_p, c3 * 8) == hipSuccess &&
                    hipMalloc((void**)&e_st, B3 * 4) == hipSuccess && hipMalloc((void**)&e_ns, B3 * 4) == hipSuccess;
    long long b3 = B3;
    auto upload = [&]() { hipMemcpy(e_q, q3.data(), c3 * 8, hipMemcpyHostToDevice); hipMemcpy(e_p, p3.data(), c3 * 8, hipMemcpyHostToDevice); };
    auto download = [&](std::vector<double>& y) { hipMemcpy(y.data(), e_q, c3 * 8, hipMemcpyDeviceToHost); hipMemcpy(y.data() + c3, e_p, c3 * 8, hipMemcpyDeviceToHost); };
    if (alloc_ok) {
      std::vector<double> ref3b(2 * c3);
      // Between repeated launches a kernel of the module fills every VGPR of every SIMD with launch-dependent
      // values: a stepping kernel that reads registers it did not write (scripts/probes/sgpr_spill_repro)
      // agrees with itself back to back and differs once something else has used the registers.
      auto scribble = [&](unsigned seed) {
        void* as[] = {&seed};
        if (hipModuleLaunchKernel(s->mod().fn[K_SCRIBBLE], 2048, 1, 1, 256, 1, 1, 0, s->cur->stream, as, nullptr) != hipSuccess) (void)hipGetLastError();
      };
      for (int r = 0; r < 2 && rc == HAMK_OK; ++r) {         // the fixed-step kernel, twice as well
        scribble(0x9e3779b9u * (unsigned)(r + 1));
        upload();
        double ddt = T / 64, no_drift = 0.0; int ns = 64;
        void* a4[] = {&e_q, &e_p, &b3, &ddt, &ns, &no_drift, &e_st};
        rc = launch(s, K_RK4, B3, a4);
        if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RK4 kernel failed");
        download(r == 0 ? ref3 : ref3b);
      }
      hipMemcpy(st_ref.data(), e_st, B3 * 4, hipMemcpyDeviceToHost);
      if (rc == HAMK_OK && std::memcmp(ref3.data(), ref3b.data(), 2 * c3 * 8) != 0) {
        *rk4_ok = false;
        g_selfcheck_detail = "two runs of the RK4 kernel on the same input DIFFER";
      }
      for (int r = 0; r < 2 && rc == HAMK_OK && has_rkf; ++r) {
        scribble(0x85ebca6bu * (unsigned)(r + 3));
        upload();
        double h0 = T / 100, ea = kRefEpsilon, er = kRefEpsilon, t0 = 0.0, t1 = T;
        int nt = 2, flags = rkf_flags(1, 1, s->gsl_api), max_sub = 4096;
        const double *cq = e_q, *cp = e_p, *cts = nullptr;
        int ncalls = 1, it_every = 0;
        void* a5[] = {&cq, &cp, &e_q, &e_p, &b3, &nt, &cts, &t0, &t1, &h0, &ea, &er, &flags, &max_sub, &e_st, &e_ns, &ncalls, &it_every};
        rc = launch(s, K_RKF45, B3, a5);
        if (rc == HAMK_OK && hipStreamSynchronize(s->cur->stream) != hipSuccess) rc = fail(HAMK_ERR_HIP, "self-check: RKF45 kernel failed");
        download(run[r]);
        hipMemcpy(st_run[r].data(), e_st, B3 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(ns_run[r].data(), e_ns, B3 * 4, hipMemcpyDeviceToHost);
      }
      if (rc == HAMK_OK && has_rkf) {
        bool same = std::memcmp(run[0].data(), run[1].data(), 2 * c3 * 8) == 0 && ns_run[0] == ns_run[1] && st_run[0] == st_run[1];
        double worst = 0.0;
        for (int64_t i = 0; i < B3; ++i) {
          if (st_ref[(size_t)i] != 0 || st_run[0][(size_t)i] != 0) continue;       // outside the system's domain: cannot judge
          if (ns_run[0][(size_t)i] > 16) continue;       // a hard stretch: 64 RK4 steps are no yardstick there
          for (int j = 0; j < 2 * n; ++j) {
            const double a = run[0][(size_t)j * B3 + i], f = ref3[(size_t)j * B3 + i];
            const double e = std::fabs(a - f) / std::fmax(1.0, std::fabs(f));
            if (!(e <= worst)) worst = e;
          }
        }
        if (!same || !(worst <= 1e-4)) {
          *rkf_ok = false;
          char msg[160];
          std::snprintf(msg, sizeof msg, "adaptive end-to-end check: two runs %s, worst deviation from 64 RK4 steps %.3g",
                        same ? "agree" : "DIFFER", worst);
          g_selfcheck_detail = msg;
          if (std::getenv("HAMK_SELFCHECK_VERBOSE")) std::fprintf(stderr, "hamk self-check: %s\n", msg);
        }
      }
    }
    hipFree(e_q); hipFree(e_p); hipFree(e_st); hipFree(e_ns);
    (void)hipGetLastError();
  }
  if (const char* e = std::getenv("HAMK_SELFCHECK_FAULT")) {        // test hook: pretend the unrolled body is wrong
    if (std::strstr(e, "rk4") && !s->curv->desc.rk4_stage_loop) *rk4_ok = false;
    if (std::strstr(e, "rkf") && !s->curv->desc.rkf_stage_loop) *rkf_ok = false;
  }
  return rc;
}

static int self_check(hamk_system* s) {
  if (!s->self_check_on) return HAMK_OK;
  Variant* v = s->curv;
  if (s->mod().self_checked) return HAMK_OK;
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool rk4_ok = true, rkf_ok = true;
    int rc = self_check_once(s, &rk4_ok, &rkf_ok);
    if (rc != HAMK_OK) return rc;
    if (rk4_ok && rkf_ok) { s->mod().self_checked = true; return HAMK_OK; }
    const bool can_retry = attempt == 0 && v->mapping == HAMK_MAP_LANE &&
                           ((!rk4_ok && !v->desc.rk4_stage_loop) || (!rkf_ok && !v->desc.rkf_stage_loop));
    if (!can_retry)
      return fail(HAMK_ERR_COMPILE, std::string("self-check failed: the fused ") + (!rk4_ok ? "RK4" : "RKF45") +
                                        " kernel disagrees with the hamEqs kernel (miscompiled module?)" +
                                        (g_selfcheck_detail.empty() ? "" : " [" + g_selfcheck_detail + "]"));
    if (!rk4_ok) v->desc.rk4_stage_loop = true;            // rebuild with the stage-loop bodies
    if (!rkf_ok) v->desc.rkf_stage_loop = true;
    v->source = generate_source(v->desc);
    rc = build_code(v, s->cache_on, build_force(s));
    if (rc != HAMK_OK) return rc;
    v->generation++;                                       // other devices reload (and re-check) lazily
    for (DevState* d : s->devs) if (d != s->cur) d->mod[v->mapping].self_checked = false;
    rc = load_modules(s);
    if (rc != HAMK_OK) return rc;
    v->self_check_rebuilds++;
  }
  return fail(HAMK_ERR_COMPILE, "self-check failed");
}

static int load_modules(hamk_system* s) {
  DevModule* d = &s->mod();
  const Variant* v = s->curv;
  d->unload();
  HIP_TRY(hipModuleLoadData(&d->module, v->code.data()));
  if (!v->code2.empty()) HIP_TRY(hipModuleLoadData(&d->module2, v->code2.data()));
  for (int k = 0; k < K__COUNT; ++k)
    if (v->has[k]) HIP_TRY(hipModuleGetFunction(&d->fn[k], (v->use2[k] && d->module2) ? d->module2 : d->module, kKernelNames[k]));
  d->code_generation = v->generation;
  return HAMK_OK;
}

// The state of the calling thread's current device (created on first use; nothing is loaded yet).
static int current_device_state(hamk_system* s) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  if (!s->cur || s->cur->device != dev) {
    s->cur = nullptr;
    for (DevState* d : s->devs) if (d->device == dev) s->cur = d;
    if (!s->cur) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, dev));
      if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HAMK_ERR_NODEVICE, std::string("device ") + prop.gcnArchName + " is not gfx950 (MI355X); libhamk has no other code path");
      DevState* d = new DevState();
      d->device = dev;
      s->devs.push_back(d);
      s->cur = d;
    }
  }
  return HAMK_OK;
}

static int variant_for(hamk_system* s, int64_t B, int kernel, Variant** out);
static std::string check_options(const hamk_options& o, int n);

// Everything a call over B trajectories needs in place: the device state, the specialisation chosen for (n, B) --
// built on first use --, its modules loaded on this device and self-checked.
static int bind_device(hamk_system* s, int64_t B, int kernel) {
  TRY0(current_device_state(s));
  TRY0(variant_for(s, B, kernel, &s->curv));
  DevModule& m = s->mod();
  if (m.module && m.code_generation == s->curv->generation) return HAMK_OK;
  TRY0(load_modules(s));
  return self_check(s);
}

static int64_t trajectories_per_block(const Variant* v) {
  // lane kernels: one trajectory per thread; wave kernels: 64/NP trajectories per wavefront; quad: four lanes each
  const int n = v->desc.n;
  if (v->mapping == HAMK_MAP_WAVE) return 4 * (64 / (n <= 16 ? 16 : n <= 32 ? 32 : 64));
  if (v->mapping == HAMK_MAP_QUAD) return 64;
  return 256;
}

static int launch(hamk_system* s, KernelId k, int64_t B, void** args) {
  const unsigned block = 256;
  const int64_t per_block = trajectories_per_block(s->curv);
  const int64_t grid = (B + per_block - 1) / per_block;
  if (grid > 0x7fffffffLL) return fail(HAMK_ERR_INVALID, "ensemble too large for one launch");
  HIP_TRY(hipModuleLaunchKernel(s->mod().fn[k], (unsigned)grid, 1, 1, block, 1, 1, 0, s->cur->stream, args, nullptr));
  return HAMK_OK;
}

// ---- host-pointer staging -----------------------------------------------------
namespace {
struct Staged {
  void* dev = nullptr;
  void* host = nullptr;
  void* pinned = nullptr;      // non-null: `dev` aliases this pinned host block
  size_t bytes = 0;
  bool out = false;
};
constexpr size_t kPinArena = 256 << 10;    // bytes of pinned arena per handle
constexpr size_t kPinMaxBuf = 32 << 10;    // larger arrays go through device staging
class Stager {
 public:
  explicit Stager(hamk_system* s, int mem) : s_(s), host_(mem == HAMK_MEM_HOST) {}
  // returns the device pointer to use for `p` (nullptr stays nullptr)
  template <class T> int in(const T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), true, false, (void**)dev); }
  template <class T> int out(T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), false, true, (void**)dev); }
  template <class T> int inout(T* p, size_t count, T** dev) { return add((void*)p, count * sizeof(T), true, true, (void**)dev); }
  // small read-only side input (evolveHam's time grid): pinned copy, or nullptr if it does not fit
  const double* side_input(const double* p, size_t count) {
    if (!host_) return nullptr;
    void* host = nullptr; void* dev = nullptr;
    if (!pin_alloc(count * sizeof(double), &host, &dev)) return nullptr;
    std::memcpy(host, p, count * sizeof(double));
    return (const double*)dev;
  }
  int finish() {
    if (!host_) return HAMK_OK;
    for (auto& b : bufs_)
      if (b.out && !b.pinned) HIP_TRY(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, s_->cur->stream));
    HIP_TRY(hipStreamSynchronize(s_->cur->stream));
    for (auto& b : bufs_)
      if (b.out && b.pinned) std::memcpy(b.host, b.pinned, b.bytes);
    return HAMK_OK;
  }

 private:
  int add(void* p, size_t bytes, bool copy_in, bool copy_out, void** dev) {
    if (!p || !host_ || bytes == 0) { *dev = p; return HAMK_OK; }
    Staged b; b.host = p; b.bytes = bytes; b.out = copy_out;
    if (bytes <= kPinMaxBuf && pin_alloc(bytes, &b.pinned, &b.dev)) {
      if (copy_in) std::memcpy(b.pinned, p, bytes);
      bufs_.push_back(b);
      *dev = b.dev;
      return HAMK_OK;
    }
    const size_t slot = nstaged_++;
    if (slot >= s_->cur->stage_buf.size()) { s_->cur->stage_buf.push_back(nullptr); s_->cur->stage_cap.push_back(0); }
    if (s_->cur->stage_cap[slot] < bytes) {
      HIP_TRY(hipStreamSynchronize(s_->cur->stream));          // nobody may still be using the old block
      if (s_->cur->stage_buf[slot]) hipFree(s_->cur->stage_buf[slot]);
      s_->cur->stage_buf[slot] = nullptr; s_->cur->stage_cap[slot] = 0;
      HIP_TRY(hipMalloc(&s_->cur->stage_buf[slot], bytes));
      s_->cur->stage_cap[slot] = bytes;
    }
    b.dev = s_->cur->stage_buf[slot];
    bufs_.push_back(b);
    if (copy_in) HIP_TRY(hipMemcpyAsync(b.dev, p, bytes, hipMemcpyHostToDevice, s_->cur->stream));
    *dev = b.dev;
    return HAMK_OK;
  }
  // bump allocation in the handle's pinned arena; every host-pointer call ends with a stream
  // synchronisation (finish), so the arena is free again when the next call starts
  bool pin_alloc(size_t bytes, void** host, void** dev) {
    if (s_->cur->pin_failed) return false;
    if (!s_->cur->pin) {
      static const bool off = [] { const char* e = std::getenv("HAMK_PINNED"); return e && e[0] == '0'; }();
      void* h = nullptr; void* d = nullptr;
      static const bool noncoh = [] { const char* e = std::getenv("HAMK_PINNED"); return e && e[0] == 'n'; }();   // test hook: the broken variant
      if (off || hipHostMalloc(&h, kPinArena, hipHostMallocMapped | (noncoh ? 0u : hipHostMallocCoherent)) != hipSuccess ||
          hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        if (h) hipHostFree(h);
        (void)hipGetLastError();
        s_->cur->pin_failed = true;
        return false;
      }
      s_->cur->pin = (char*)h; s_->cur->pin_dev = (char*)d;
    }
    const size_t at = (pin_used_ + 255) & ~(size_t)255;
    if (at + bytes > kPinArena) return false;
    pin_used_ = at + bytes;
    *host = s_->cur->pin + at; *dev = s_->cur->pin_dev + at;
    return true;
  }
  hamk_system* s_;
  bool host_;
  size_t pin_used_ = 0;
  size_t nstaged_ = 0;
  std::vector<Staged> bufs_;
};
}  // namespace

#define TRY(expr) do { int rc_ = (expr); if (rc_ != HAMK_OK) return rc_; } while (0)

static int check_call(hamk_system* s, int64_t B, int32_t mem) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (B < 0) return fail(HAMK_ERR_INVALID, "negative ensemble size");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  return HAMK_OK;
}

// ---------------------------------------------------------------------------
// options -> specialisations
// ---------------------------------------------------------------------------
// Ensemble size below which a lane-kernel system (n <= 16) runs four lanes per trajectory instead (0: never).
// One trajectory per lane does the least work per trajectory but puts 64 of them in a wavefront: 65 536 trajectories are
// 1024 wavefronts -- one per SIMD -- and every halving of the ensemble idles half the chip, while the quad kernels spread
// the same trajectories over four times the wavefronts.  Measured on MI355X (profiles/r03_throughput_vs_B.jsonl,
// r03_throughput_vs_B_quad.jsonl; RK4 steps/s):
//     chain16   B = 8192: lane 2.85e8, quad 4.76e8, wave 2.28e8    16384: 5.67e8 / 9.48e8 / 2.42e8    32768: 1.13e9 / 9.5e8
//     chain8    B = 8192: lane 1.23e9, quad 1.12e9, wave 4.8e8     16384: 2.45e9 / 2.22e9             (lane throughout)
//     threeBodyPolar (n = 6): lane 2.1e9 at 8192 against quad 1.15e9 (lane throughout)
//     chain14   B = 8192: lane 3.86e8, quad 4.98e8    16384: 7.69e8 / 9.92e8    32768: 1.54e9 / 1.28e9
//     chain12   B = 8192: lane 5.33e8, quad 7.01e8    16384: 1.06e9 / 1.37e9    32768: 2.12e9 / 1.73e9
//     chain11   B = 8192: lane 6.69e8, quad 7.39e8    16384: 1.34e9 / 1.46e9    32768: 2.68e9 / 1.85e9
//     chain10   B = 8192: lane 7.99e8, quad 7.94e8    16384: 1.59e9 / 1.58e9    32768: 3.17e9 / 1.99e9   (a tie: lane)
// (profiles/r03_rules_probe.jsonl; chain13 the same picture, +15 %).  The wave-cooperative kernels never win at n <= 16
// for B >= 8192.
static int64_t quad_below(int n) {
  return n >= 11 ? 32768 : 0;
}

static bool env_flag(const char* name, bool* value) {           // "0" / "1" test overrides
  const char* e = std::getenv(name);
  if (!e || (e[0] != '0' && e[0] != '1')) return false;
  *value = e[0] == '1';
  return true;
}

static std::string check_options(const hamk_options& o, int n) {
  auto in = [](int v, std::initializer_list<int> ok) { for (int k : ok) if (v == k) return true; return false; };
  if (!in(o.mapping, {HAMK_AUTO, HAMK_MAP_LANE, HAMK_MAP_WAVE, HAMK_MAP_QUAD})) return "mapping must be HAMK_AUTO or a HAMK_MAP_* value";
  if (o.mapping == HAMK_MAP_LANE && n > 16) return "unsupported: HAMK_MAP_LANE needs n <= 16 (one trajectory no longer fits one lane)";
  if (o.mapping == HAMK_MAP_QUAD && n > 32) return "unsupported: HAMK_MAP_QUAD needs n <= 32 (a lane holds a quarter of K in registers)";
  if (!in(o.ad_mode, {HAMK_AUTO, HAMK_AD_H, HAMK_AD_D, HAMK_AD_R})) return "ad_mode must be HAMK_AUTO or a HAMK_AD_* value";
  if (!in(o.rk4_body, {HAMK_AUTO, HAMK_BODY_UNROLLED, HAMK_BODY_STAGE_LOOP}) || !in(o.rkf_body, {HAMK_AUTO, HAMK_BODY_UNROLLED, HAMK_BODY_STAGE_LOOP}))
    return "rk4_body / rkf_body must be HAMK_AUTO or a HAMK_BODY_* value";
  if (!in(o.trig, {HAMK_AUTO, HAMK_TRIG_DIRECT, HAMK_TRIG_TABLE, HAMK_TRIG_TABLE_ROTATE})) return "trig must be HAMK_AUTO or a HAMK_TRIG_* value";
  if (!in(o.gsl_api, {HAMK_AUTO, 1, 2})) return "gsl_api must be HAMK_AUTO, 1 (gsl_odeiv) or 2 (gsl_odeiv2)";
  if (!in(o.build, {HAMK_AUTO, HAMK_BUILD_DEFAULT, HAMK_BUILD_NOLICM})) return "build must be HAMK_AUTO or a HAMK_BUILD_* value";
  for (int v : {o.self_check, o.wave_blocked, o.k_reassoc, o.rk4_park, o.rkf_park, o.cache})
    if (!in(v, {HAMK_AUTO, HAMK_ON, HAMK_OFF})) return "switches must be HAMK_AUTO, HAMK_ON or HAMK_OFF";
  if (o.rk4_min_waves < 0 || o.rk4_min_waves > 8) return "rk4_min_waves must be 0 (auto) .. 8";
  if (o.max_substeps < 0) return "max_substeps must be >= 0";
  if (o.ensemble_size < 0) return "ensemble_size must be >= 0";
  return std::string();
}

// Which lanes serve a trajectory for an ensemble of B (hamk_options::mapping = HAMK_AUTO).
// The lane kernels do the least work per trajectory (compile-time sparsity of the seeds, everything in registers) but
// put 64 trajectories in a wavefront: below ~64 x 1024 SIMDs trajectories they leave SIMDs idle, and for the systems
// whose lane kernel is large the wave-cooperative kernels (4 trajectories per wavefront at n <= 16) then win.
// Thresholds measured on MI355X (scripts/sweep_batch.py -> profiles/r03_throughput_vs_B*.jsonl).
// Which kernels the quad module (hamk_quad.hpp) provides: all eight since the second half of round 3.  The per-kernel
// dispatch stays: a module that lacks a kernel leaves it to the module that serves the system's size otherwise.
static bool quad_has(int kernel) { (void)kernel; return true; }      // (the first version provided the four kernels of the hot path only)

// Can a lane run the per-trajectory first-order sweep of this system with compile-time seeds?  It keeps one register pair
// per DISTINCT entry of the Jacobian (hamk_codegen.cpp distinct_jacobian_entries): 2n for a chain, m n for a dense map.
static bool quad_eligible(hamk_system* s) {
  if (s->quad_eligible < 0) s->quad_eligible = distinct_jacobian_entries(s->base) <= 8 * s->base.n ? 1 : 0;
  return s->quad_eligible == 1;
}

// Is K = J^T M J semi-definite by construction?  Only then may a kernel factorise it without pivoting.  The reference
// inverts EVERY K by LU with partial pivoting (hmatrix `inv`, Hamilton.hs:321, :381), so a system with a non-positive
// inertia -- K symmetric but possibly indefinite, and still invertible -- must run on kernels that pivot: the lane
// kernels (solve_spd falls back to solve_lu per trajectory) and the wave-cooperative ones (solve_pivoted).  The four-lane
// kernels do not, and are never chosen for such a system.
static bool inertia_positive(const hamk_system* s) {
  for (double w : s->base.inertia) if (!(w > 0.0)) return false;
  return true;
}

static int choose_mapping(hamk_system* s, int64_t B, int kernel) {
  const int n = s->base.n;
  const int rest = n > 16 ? HAMK_MAP_WAVE : HAMK_MAP_LANE;  // where the kernels the quad module lacks run
  if (s->opt.mapping != HAMK_AUTO) return (s->opt.mapping == HAMK_MAP_QUAD && !quad_has(kernel)) ? rest : s->opt.mapping;
  if (s->ensemble_size > 0) B = s->ensemble_size;           // the launch is a piece of a larger ensemble: one mapping for all pieces
  bool w = false;
  const bool pos = inertia_positive(s);
  if (pos && env_flag("HAMK_QUAD", &w) && w && n <= 32) return quad_has(kernel) ? HAMK_MAP_QUAD : rest;      // tests / experiments
  const bool no_quad = !pos || (env_flag("HAMK_QUAD", &w) && !w);
  if (env_flag("HAMK_WAVE", &w)) return (w || n > 16) ? HAMK_MAP_WAVE : HAMK_MAP_LANE;
  if (n > 32) return HAMK_MAP_WAVE;
  if (n > 16) return (!no_quad && quad_has(kernel) && quad_eligible(s)) ? HAMK_MAP_QUAD : HAMK_MAP_WAVE;
  int64_t below = quad_below(n);                            // ensembles smaller than this leave the lane kernels
  if (const char* e = std::getenv("HAMK_QUAD_BELOW")) below = std::atoll(e);                // experiments: the crossover itself
  if (B < below && !no_quad && quad_eligible(s)) return quad_has(kernel) ? HAMK_MAP_QUAD : HAMK_MAP_LANE;
  return HAMK_MAP_LANE;
}

static SystemDesc make_desc(const hamk_system* s, int mapping, bool* forced_rk4, bool* forced_rkf) {
  const hamk_options& o = s->opt;
  SystemDesc d = s->base;
  const int n = d.n, m = d.m;
  bool b = false;
  d.mapping = mapping;
  d.wave = mapping == HAMK_MAP_WAVE;
  // second-order AD: measured on MI355X (scripts/sweep.py): H >= D up to n = 3, D ahead from n = 4; the reverse sweep
  // pays from n = 8 (chain8 +4 %, chain16 +12 %); below, the compiler already strips the structural zeros of the
  // directional jets and Jet2 is as cheap
  d.mode_h = (n <= 3);
  d.mode_r = (n >= 8);
  int ad = o.ad_mode;
  if (ad == HAMK_AUTO) if (const char* e = std::getenv("HAMK_AD_MODE")) ad = (e[0] == 'H' || e[0] == 'h') ? HAMK_AD_H : (e[0] == 'D' || e[0] == 'd') ? HAMK_AD_D : (e[0] == 'R' || e[0] == 'r') ? HAMK_AD_R : HAMK_AUTO;
  if (ad == HAMK_AD_H) { d.mode_h = true; d.mode_r = false; }
  if (ad == HAMK_AD_D) { d.mode_h = false; d.mode_r = false; }
  if (ad == HAMK_AD_R) { d.mode_h = false; d.mode_r = true; }
  // LDL^T in panels of 16 with the trailing blocks on the matrix cores (hamk_wave.hpp factor_blocked): measured on
  // MI355X chain32 5.06e7 -> 5.66e7, chain64 4.6e6 -> 7.6e6 RK4 steps/s (profiles/r02_wave_blocked.jsonl); a single panel
  // (n <= 16, forced wave path) has nothing to block
  d.wave_blocked = n > 16;
  if (o.wave_blocked != HAMK_AUTO) d.wave_blocked = (o.wave_blocked == HAMK_ON) && n > 16;
  else if (env_flag("HAMK_WAVE_BLOCKED", &b)) d.wave_blocked = b && n > 16;
  d.rk4_stage_loop = (n >= 7);
  *forced_rk4 = true;
  if (o.rk4_body != HAMK_AUTO) d.rk4_stage_loop = o.rk4_body == HAMK_BODY_STAGE_LOOP;
  else if (env_flag("HAMK_RK4_LOOP", &b)) d.rk4_stage_loop = b;
  else *forced_rk4 = false;
  d.rkf_stage_loop = (n >= 4);
  *forced_rkf = true;
  if (o.rkf_body != HAMK_AUTO) d.rkf_stage_loop = o.rkf_body == HAMK_BODY_STAGE_LOOP;
  else if (env_flag("HAMK_RKF_LOOP", &b)) d.rkf_stage_loop = b;
  else *forced_rkf = false;
  // n > 32 (one trajectory per wavefront): the RK4 kernel capped at 256 VGPRs -- two wavefronts per SIMD, ~160
  // spilled registers -- beats one wavefront with everything in registers: chain48 1.05e7 -> 1.45e7, chain64
  // 7.6e6 -> 9.5e6 RK4 steps/s on MI355X (profiles/r02_wave_blocked.jsonl)
  d.rk4_min_waves = 1;
  if (d.wave && n > 32) d.rk4_min_waves = 2;
  if (o.rk4_min_waves > 0) d.rk4_min_waves = o.rk4_min_waves;
  else if (const char* e = std::getenv("HAMK_RK4_WAVES")) d.rk4_min_waves = std::atoi(e);
  // RK4 stage loop with y / acc parked in LDS (hamk_device.hpp rk4_body): where one right-hand side alone fills the
  // register file the waiting state is what spills; chain16 300 spilled registers -> 34, none in the loop.  Measured
  // at B = 65 536 (profiles/r03_rules_probe.jsonl; RK4 steps/s parked / not): chain16 2.09e9 / 1.01e9, chain14
  // 2.77e9 / 2.63e9, chain13 3.19e9 / 3.46e9, chain12 3.74e9 / 3.90e9, chain10 5.03e9 / 5.42e9 -- it pays from n = 14
  d.rk4_park = mapping == HAMK_MAP_LANE && n >= 14;
  if (o.rk4_park != HAMK_AUTO) d.rk4_park = o.rk4_park == HAMK_ON;
  else if (env_flag("HAMK_RK4_PARK", &b)) d.rk4_park = b;
  if (d.rk4_park && (!d.rk4_stage_loop || mapping != HAMK_MAP_LANE || n > 16)) d.rk4_park = false;          // 2 x 2n x 2 KiB of LDS per block: n <= 16
  // RKF45 stepper whose nine vectors (18 n doubles per trajectory) wait in LDS (y, dydt, the first k's) and in a
  // run-time-indexed private array (scratch memory, touched only between right-hand sides) instead of competing with K
  // for registers (hamk_device.hpp rkf45_body_parked, hamk_quad.hpp rkf45_body_parked).  Measured on MI355X, stepHam
  // calls/s, registers / parked.  Lane kernels at B = 65 536 (profiles/r03_lane_rkf_park.jsonl): chain16 1.57e7 / 7.88e7
  // (1516 -> 24 spilled registers), chain14 2.87e7 / 1.27e8, chain12 5.08e7 / 1.80e8, chain10 7.59e7 / 2.50e8, chain8
  // 2.03e8 / 3.77e8, chain7 3.29e8 / 4.99e8, chain6 5.00e8 / 6.12e8, threeBodyPolar 7.08e8 / 8.21e8, chain5 and chain4 ties.
  // Quad kernels at B = 16 384 (profiles/r03_quad_rkf_park.jsonl): chain32 4.14e6 / 6.63e6, chain24 1.40e7 / 1.74e7,
  // chain20 2.70e7 / 2.94e7, chain17 3.34e7 / 3.48e7
  d.rkf_park = (mapping == HAMK_MAP_LANE && d.rkf_stage_loop && n >= 6) || (mapping == HAMK_MAP_QUAD && n >= 17);
  if (o.rkf_park != HAMK_AUTO) d.rkf_park = o.rkf_park == HAMK_ON;
  else if (env_flag("HAMK_RKF_PARK", &b)) d.rkf_park = b;
  if (d.rkf_park && (mapping == HAMK_MAP_WAVE || (mapping == HAMK_MAP_LANE && !d.rkf_stage_loop))) d.rkf_park = false;
  d.k_reassoc = true;
  if (o.k_reassoc != HAMK_AUTO) d.k_reassoc = o.k_reassoc == HAMK_ON;
  else if (env_flag("HAMK_K_REASSOC", &b)) d.k_reassoc = b;
  {
    // sincos in the stepping kernels (hamk_device.hpp StageTrig).  Every evaluation through the LDS table is
    // the fewest instructions, but each is a 16-byte gather at a lane-dependent address (~20-25 LDS cycles
    // per wavefront) and the CU's 16 wavefronts share one LDS unit: where a right-hand side is short and
    // trig-dense the unit saturates, and taking only the step's one full evaluation from the table (stages
    // 2-4 by rotation in registers) is faster.  Measured on MI355X (profiles/r02_sweep_trig.jsonl): rotation
    // wins for doublePendulum (2 sites per ~90-instruction RHS: 8.36 vs 8.21e10) and pendulum, the table for
    // twoBody (+8 %), threeBodyPolar (+8 %) and the chains (+24 % at n = 8); spring is a tie.  The rule
    // below reproduces those choices from an estimate of the instructions per RHS and sincos site.
    const int f_nops = (int)d.f_ops.size(), u_nops = (int)d.u_ops.size();
    std::vector<char> seen(f_nops > 0 ? f_nops : 1, 0);
    int sites = 0;
    for (int i = 0; i < f_nops; ++i)
      if ((d.f_ops[i].op == HAMK_OP_SIN || d.f_ops[i].op == HAMK_OP_COS) && !seen[d.f_ops[i].a]) { seen[d.f_ops[i].a] = 1; ++sites; }
    const double width = d.mode_h ? 1.0 + n + 0.5 * n * (n + 1) : 3.0 * n + 3.0;      // jet components carried per tape value
    const double est_rhs = (f_nops + u_nops) * width + 2.0 * m * n * n + n * n * n / 3.0;
    d.use_lut = (sites >= 1 && sites <= 4 && est_rhs / sites < 100.0) ? 2 : 1;
    // the largest lane kernels (n >= 14) without the parked state: 512 VGPRs and hundreds spilled, bound by their
    // scratch traffic; measured, the table variant schedules worse there (chain16 9.4e8 vs 1.05e9 steps/s)
    if (n >= 14 && !d.wave && !d.rk4_park) d.use_lut = 0;
  }
  if (o.trig != HAMK_AUTO) d.use_lut = o.trig == HAMK_TRIG_DIRECT ? 0 : (o.trig == HAMK_TRIG_TABLE ? 1 : 2);
  else if (const char* e = std::getenv("HAMK_TRIG_LUT")) { if (e[0] >= '0' && e[0] <= '2') d.use_lut = e[0] - '0'; }
  return d;
}

static int variant_for(hamk_system* s, int64_t B, int kernel, Variant** out) {
  const int mapping = choose_mapping(s, B, kernel);
  if (s->var[mapping]) { *out = s->var[mapping]; return HAMK_OK; }
  Variant* v = new Variant();
  v->mapping = mapping;
  v->desc = make_desc(s, mapping, &v->forced_rk4_body, &v->forced_rkf_body);
  v->source = generate_source(v->desc);
  int rc = build_code(v, s->cache_on, build_force(s));
  if (rc != HAMK_OK) { delete v; return rc; }
  // keep every kernel comfortably inside SOPP branch reach: fall back to the stage-loop bodies
  const size_t kLimit = 64 * 1024;
  const bool lane = mapping == HAMK_MAP_LANE;
  const bool big_rkf = lane && !v->forced_rkf_body && !v->desc.rkf_stage_loop && chosen_kernel_bytes(v, K_RKF45) > kLimit;
  const bool big_rk4 = lane && !v->forced_rk4_body && !v->desc.rk4_stage_loop && chosen_kernel_bytes(v, K_RK4) > kLimit;
  if (big_rkf || big_rk4) {
    if (big_rkf) v->desc.rkf_stage_loop = true;
    if (big_rk4) v->desc.rk4_stage_loop = true;
    v->source = generate_source(v->desc);
    rc = build_code(v, s->cache_on, build_force(s));
    if (rc != HAMK_OK) { delete v; return rc; }
  }
  for (int k = 0; k < K__COUNT; ++k) v->has[k] = mapping != HAMK_MAP_QUAD || quad_has(k);
  s->var[mapping] = v;
  *out = v;
  return HAMK_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char* hamk_last_error(void) { return g_last_error.c_str(); }
const char* hamk_version(void) { return "hamk 0.2 (gfx950; hiprtc-specialised jets; RK4 + GSL-semantics RKF45, gsl_odeiv2 / gsl_odeiv)"; }

int hamk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hamk_set_device(int32_t device) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return HAMK_OK;
}

int hamk_get_device(int32_t* device) {
  if (!device) return fail(HAMK_ERR_INVALID, "null device");
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return fail(HAMK_ERR_NODEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  *device = d;
  return HAMK_OK;
}

int hamk_device_malloc(void** ptr, int64_t bytes) {
  if (!ptr || bytes < 0) return fail(HAMK_ERR_INVALID, "hamk_device_malloc: null ptr / negative size");
  *ptr = nullptr;
  if (bytes == 0) return HAMK_OK;
  HIP_TRY(hipMalloc(ptr, (size_t)bytes));
  return HAMK_OK;
}

int hamk_device_free(void* ptr) {
  if (ptr) HIP_TRY(hipFree(ptr));
  return HAMK_OK;
}

int hamk_memcpy(void* dst, const void* src, int64_t bytes, int32_t kind) {
  if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(HAMK_ERR_INVALID, "hamk_memcpy: null pointer / negative size");
  if (bytes == 0) return HAMK_OK;
  hipMemcpyKind k;
  switch (kind) {
    case HAMK_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case HAMK_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case HAMK_COPY_D2D: k = hipMemcpyDefault; break;        // unified addressing: same device or a peer
    default: return fail(HAMK_ERR_INVALID, "hamk_memcpy: kind must be HAMK_COPY_H2D / D2H / D2D");
  }
  HIP_TRY(hipMemcpy(dst, src, (size_t)bytes, k));
  return HAMK_OK;
}

int hamk_gather_batch(int32_t nparts, int32_t n, const int64_t* B_parts, const double* const* parts, double* out,
                      int32_t out_mem) {
  if (nparts < 0 || n <= 0 || (nparts > 0 && (!B_parts || !parts)))
    return fail(HAMK_ERR_INVALID, "hamk_gather_batch: bad nparts / n / null arrays");
  if (out_mem != HAMK_MEM_HOST && out_mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "out_mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  int64_t total = 0;
  for (int g = 0; g < nparts; ++g) {
    if (B_parts[g] < 0 || (B_parts[g] > 0 && !parts[g])) return fail(HAMK_ERR_INVALID, "hamk_gather_batch: negative size / null part");
    total += B_parts[g];
  }
  if (total == 0) return HAMK_OK;
  if (!out) return fail(HAMK_ERR_INVALID, "hamk_gather_batch: null out");
  int here = 0;
  HIP_TRY(hipGetDevice(&here));
  std::string peer_note;                                   // why a direct xGMI path could not be set up, if so
  int64_t at = 0;
  for (int g = 0; g < nparts; ++g) {
    const int64_t Bg = B_parts[g];
    if (Bg == 0) continue;
    if (out_mem == HAMK_MEM_DEVICE) {                       // direct xGMI DMA where the devices are peers
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, parts[g]) == hipSuccess && attr.device != here) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, here, attr.device) == hipSuccess && can) {
          hipError_t e = hipDeviceEnablePeerAccess(attr.device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            peer_note = std::string(" [hipDeviceEnablePeerAccess(") + std::to_string(attr.device) + "): " + hipGetErrorString(e) + "]";
        } else {
          peer_note = " [device " + std::to_string(attr.device) + " is not a peer of device " + std::to_string(here) + "]";
        }
      }
      (void)hipGetLastError();                             // the copy below still works without peer access (staged by the runtime)
    }
    // row j of the part goes to columns [at, at + Bg) of row j of the output
    {
      hipError_t e = hipMemcpy2DAsync(out + at, (size_t)total * sizeof(double), parts[g], (size_t)Bg * sizeof(double),
                                      (size_t)Bg * sizeof(double), (size_t)n, hipMemcpyDefault, nullptr);
      if (e != hipSuccess) return fail(HAMK_ERR_HIP, std::string("hamk_gather_batch: copy of part ") + std::to_string(g) + ": " + hipGetErrorString(e) + peer_note);
    }
    at += Bg;
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
  return HAMK_OK;
}

void hamk_options_init(hamk_options* opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof *opt);
  opt->size = (uint32_t)sizeof *opt;
}

int hamk_system_create(int32_t m, int32_t n, const double* inertia, const hamk_op* f_ops, int32_t f_nops,
                       const int32_t* f_outs, const hamk_op* u_ops, int32_t u_nops, int32_t u_out, int32_t u_space,
                       hamk_system** out) {
  return hamk_system_create_ex(m, n, inertia, f_ops, f_nops, f_outs, u_ops, u_nops, u_out, u_space, nullptr, out);
}

int hamk_system_create_ex(int32_t m, int32_t n, const double* inertia, const hamk_op* f_ops, int32_t f_nops,
                          const int32_t* f_outs, const hamk_op* u_ops, int32_t u_nops, int32_t u_out, int32_t u_space,
                          const hamk_options* opt, hamk_system** out) {
  if (!out) return fail(HAMK_ERR_INVALID, "out is null");
  *out = nullptr;
  if (m <= 0 || n <= 0) return fail(HAMK_ERR_INVALID, "m and n must be positive");
  if (!inertia || !f_outs) return fail(HAMK_ERR_INVALID, "null inertia / f_outs");
  if (u_space != HAMK_U_GENERALIZED && u_space != HAMK_U_CARTESIAN) return fail(HAMK_ERR_INVALID, "bad u_space");
  if (n > 64 || m > 128)
    return fail(HAMK_ERR_UNSUPPORTED, "supported sizes: n <= 16 (one trajectory per lane), 17 <= n <= 64 with m <= 128 (wave-cooperative kernels)");
  std::string err = validate_tape(f_ops, f_nops, n, f_outs, m, "coordinate map");
  if (!err.empty()) return fail(HAMK_ERR_TAPE, err);
  const int nu = (u_space == HAMK_U_CARTESIAN) ? m : n;
  err = validate_tape(u_ops, u_nops, nu, &u_out, 1, "potential");
  if (!err.empty()) return fail(HAMK_ERR_TAPE, err);
  hamk_options o;
  hamk_options_init(&o);
  if (opt) {
    if (opt->size < 8 || opt->size > 4096) return fail(HAMK_ERR_INVALID, "hamk_options: size is not set (hamk_options_init)");
    std::memcpy(&o, opt, std::min((size_t)opt->size, sizeof o));      // a caller built against an older header: the tail stays AUTO
    o.size = (uint32_t)sizeof o;
  }
  err = check_options(o, n);
  if (err.empty() && o.mapping == HAMK_MAP_QUAD)
    for (int k = 0; k < m; ++k)
      if (!(inertia[k] > 0.0))
        err = "unsupported: HAMK_MAP_QUAD factorises K = J^T M J without pivoting and needs every inertia positive; this system has "
              "inertia[" + std::to_string(k) + "] = " + std::to_string(inertia[k]) + " (leave the mapping to the library: the lane and "
              "wave-cooperative kernels pivot as the reference's `inv` does)";
  if (!err.empty()) return fail(err.rfind("unsupported:", 0) == 0 ? HAMK_ERR_UNSUPPORTED : HAMK_ERR_INVALID, "hamk_options: " + err);

  hamk_system* s = new hamk_system();
  s->opt = o;
  s->base.m = m; s->base.n = n; s->base.u_space = u_space;
  s->base.inertia.assign(inertia, inertia + m);
  s->base.f_ops.assign(f_ops, f_ops + f_nops);
  s->base.f_outs.assign(f_outs, f_outs + m);
  s->base.u_ops.assign(u_ops, u_ops + u_nops);
  s->base.u_out = u_out;
  s->gsl_api = o.gsl_api ? o.gsl_api : 2;
  if (o.gsl_api == HAMK_AUTO) if (const char* e = std::getenv("HAMK_GSL_API")) s->gsl_api = (e[0] == '1') ? 1 : 2;
  s->self_check_on = o.self_check != HAMK_OFF;
  if (o.self_check == HAMK_AUTO) if (const char* e = std::getenv("HAMK_SELFCHECK")) if (e[0] == '0') s->self_check_on = false;
  s->cache_on = o.cache != HAMK_OFF;
  s->ensemble_size = o.ensemble_size;
  s->max_substeps = o.max_substeps > 0 ? o.max_substeps : (1 << 24);
  if (o.max_substeps == HAMK_AUTO)                          // test suites: a kernel gone wrong must end, not spin through 16M attempts per lane
    if (const char* e = std::getenv("HAMK_MAX_SUBSTEPS")) { const long k = std::atol(e); if (k > 0 && k < (1L << 24)) s->max_substeps = (int)k; }
  // the specialisation a large ensemble uses is built now: a tape the kernels cannot be specialised for fails here
  Variant* v = nullptr;
  const int rc = variant_for(s, INT64_MAX, K_RK4, &v);
  if (rc != HAMK_OK) { hamk_system_destroy(s); return rc; }
  s->info = v;
  *out = s;
  return HAMK_OK;
}

void hamk_system_destroy(hamk_system* s) {
  if (!s) return;
  for (DevState* d : s->devs) { d->release(); delete d; }
  for (Variant* v : s->var) delete v;
  (void)hipGetLastError();
  delete s;
}

int hamk_system_dims(const hamk_system* s, int32_t* m, int32_t* n) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (m) *m = s->base.m;
  if (n) *n = s->base.n;
  return HAMK_OK;
}

int hamk_system_get_options(hamk_system* s, int64_t B, hamk_options* r) {
  if (!s || !r) return fail(HAMK_ERR_INVALID, "null system handle / options");
  Variant* v = nullptr;
  TRY(variant_for(s, B < 0 ? INT64_MAX : B, K_RK4, &v));
  hamk_options_init(r);
  const SystemDesc& d = v->desc;
  r->mapping = v->mapping;
  r->ad_mode = d.mode_h ? HAMK_AD_H : (d.mode_r ? HAMK_AD_R : HAMK_AD_D);
  r->rk4_body = d.rk4_stage_loop ? HAMK_BODY_STAGE_LOOP : HAMK_BODY_UNROLLED;
  r->rkf_body = d.rkf_stage_loop ? HAMK_BODY_STAGE_LOOP : HAMK_BODY_UNROLLED;
  r->trig = d.use_lut == 0 ? HAMK_TRIG_DIRECT : (d.use_lut == 1 ? HAMK_TRIG_TABLE : HAMK_TRIG_TABLE_ROTATE);
  r->gsl_api = s->gsl_api;
  r->self_check = s->self_check_on ? HAMK_ON : HAMK_OFF;
  const int bf = build_force(s);
  r->build = bf == 0 ? HAMK_BUILD_DEFAULT : (bf == 1 ? HAMK_BUILD_NOLICM : HAMK_AUTO);
  r->wave_blocked = d.wave_blocked ? HAMK_ON : HAMK_OFF;
  r->rk4_min_waves = d.rk4_min_waves;
  r->k_reassoc = d.k_reassoc ? HAMK_ON : HAMK_OFF;
  r->rk4_park = d.rk4_park ? HAMK_ON : HAMK_OFF;
  r->rkf_park = d.rkf_park ? HAMK_ON : HAMK_OFF;
  r->max_substeps = s->max_substeps;
  r->cache = s->cache_on ? HAMK_ON : HAMK_OFF;
  r->ensemble_size = s->ensemble_size;
  r->lanes_per_trajectory = v->mapping == HAMK_MAP_LANE ? 1 : (v->mapping == HAMK_MAP_QUAD ? 4 : (d.n <= 16 ? 16 : d.n <= 32 ? 32 : 64));
  return HAMK_OK;
}

int hamk_system_describe_batch(hamk_system* s, int64_t B) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  Variant* v = nullptr;
  TRY(variant_for(s, B < 0 ? INT64_MAX : B, K_RK4, &v));
  s->info = v;
  return HAMK_OK;
}

int hamk_system_set_ensemble_size(hamk_system* s, int64_t B_total) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (B_total < 0) return fail(HAMK_ERR_INVALID, "negative ensemble size");
  s->ensemble_size = B_total;
  return HAMK_OK;
}

int hamk_set_stream(hamk_system* s, void* hip_stream) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (current_device_state(s) != HAMK_OK) return HAMK_OK;  // no device to bind to: the first launch reports that
  s->cur->stream = (hipStream_t)hip_stream;
  return HAMK_OK;
}

int hamk_synchronize(hamk_system* s) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  TRY(current_device_state(s));
  HIP_TRY(hipStreamSynchronize(s->cur->stream));
  return HAMK_OK;
}

int hamk_system_set_gsl_api(hamk_system* s, int32_t api) {
  if (!s) return fail(HAMK_ERR_INVALID, "null system handle");
  if (api != 1 && api != 2) return fail(HAMK_ERR_INVALID, "gsl api must be 1 (gsl_odeiv) or 2 (gsl_odeiv2)");
  s->gsl_api = api;
  return HAMK_OK;
}

int32_t hamk_system_get_gsl_api(const hamk_system* s) { return s ? s->gsl_api : 0; }

int64_t hamk_system_code_object(const hamk_system* s, int32_t which, void* buf, int64_t cap) {
  if (!s || !s->info || (which != 0 && which != 1)) return 0;
  const std::vector<char>& c = which ? s->info->code2 : s->info->code;
  if (buf && cap >= (int64_t)c.size() && !c.empty()) std::memcpy(buf, c.data(), c.size());
  return (int64_t)c.size();
}

const char* hamk_system_source(const hamk_system* s) { return (s && s->info) ? s->info->source.c_str() : nullptr; }
const char* hamk_system_build_info(const hamk_system* s) { return (s && s->info) ? s->info->build_info.c_str() : ""; }
int64_t hamk_system_code_size(const hamk_system* s) { return (s && s->info) ? (int64_t)(s->info->code.size() + s->info->code2.size()) : 0; }
int64_t hamk_system_kernel_bytes(const hamk_system* s, const char* kernel_name) {
  if (!s || !s->info) return 0;
  if (kernel_name)
    for (int k = 0; k < K__COUNT; ++k)
      if (std::strcmp(kernel_name, kKernelNames[k]) == 0) return (int64_t)chosen_kernel_bytes(s->info, k);
  return (int64_t)kernel_code_bytes(s->info->code, kernel_name);
}

int hamk_coords_batch(hamk_system* s, int64_t B, const double* q, double* x, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !x) return fail(HAMK_ERR_INVALID, "null q / x");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_COORDS));
  Stager st(s, mem);
  const double* dq; double* dx;
  TRY(st.in(q, (size_t)s->base.n * B, (double**)&dq));
  TRY(st.out(x, (size_t)s->base.m * B, &dx));
  long long b = B;
  void* args[] = {&dq, &dx, &b};
  TRY(launch(s, K_COORDS, B, args));
  return st.finish();
}

int hamk_to_phase_batch(hamk_system* s, int64_t B, const double* q, const double* qd, double* p, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !qd || !p) return fail(HAMK_ERR_INVALID, "null q / qd / p");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_TO_PHASE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dqd; double* dp;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(qd, cnt, (double**)&dqd));
  TRY(st.out(p, cnt, &dp));
  long long b = B;
  void* args[] = {&dq, &dqd, &dp, &b};
  TRY(launch(s, K_TO_PHASE, B, args));
  return st.finish();
}

int hamk_from_phase_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* qd, int32_t* status,
                          int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p || !qd) return fail(HAMK_ERR_INVALID, "null q / p / qd");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_FROM_PHASE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dp; double* dqd; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(p, cnt, (double**)&dp));
  TRY(st.out(qd, cnt, &dqd));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&dq, &dp, &dqd, &b, &dst};
  TRY(launch(s, K_FROM_PHASE, B, args));
  return st.finish();
}

int hamk_observe_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* ke, double* pe, double* h,
                       int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q) return fail(HAMK_ERR_INVALID, "null q");
  if (!p && (ke || h)) return fail(HAMK_ERR_INVALID, "ke / h need momenta p");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_OBSERVE));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dp; double *dke, *dpe, *dh; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(p, cnt, (double**)&dp));
  TRY(st.out(ke, (size_t)B, &dke));
  TRY(st.out(pe, (size_t)B, &dpe));
  TRY(st.out(h, (size_t)B, &dh));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&dq, &dp, &dke, &dpe, &dh, &b, &dst};
  TRY(launch(s, K_OBSERVE, B, args));
  return st.finish();
}

int hamk_observe_config_batch(hamk_system* s, int64_t B, const double* q, const double* qd, double* ke, double* lag,
                              int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !qd) return fail(HAMK_ERR_INVALID, "null q / qd");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_OBSERVE_CFG));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *dq, *dqd; double *dke, *dlag;
  TRY(st.in(q, cnt, (double**)&dq));
  TRY(st.in(qd, cnt, (double**)&dqd));
  TRY(st.out(ke, (size_t)B, &dke));
  TRY(st.out(lag, (size_t)B, &dlag));
  long long b = B;
  void* args[] = {&dq, &dqd, &dke, &dlag, &b};
  TRY(launch(s, K_OBSERVE_CFG, B, args));
  return st.finish();
}

int hamk_hameqs_batch(hamk_system* s, int64_t B, const double* q, const double* p, double* dq, double* dp,
                      int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p || !dq || !dp) return fail(HAMK_ERR_INVALID, "null q / p / dq / dp");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_HAMEQS));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const double *xq, *xp; double *xdq, *xdp; int32_t* dst;
  TRY(st.in(q, cnt, (double**)&xq));
  TRY(st.in(p, cnt, (double**)&xp));
  TRY(st.out(dq, cnt, &xdq));
  TRY(st.out(dp, cnt, &xdp));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  void* args[] = {&xq, &xp, &xdq, &xdp, &b, &dst};
  TRY(launch(s, K_HAMEQS, B, args));
  return st.finish();
}

// ---- initial conditions on the device ----------------------------------------------------------------
namespace { struct HamkBoxes { double q_lo[64], q_hi[64], qd_lo[64], qd_hi[64]; }; }     // = hamk_sample.hpp

static int sample_code(bool cache_on, const std::vector<char>** out) {     // the sampler's code object: compiled once per process
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

int hamk_sample_batch(hamk_system* s, int64_t B, int64_t first_index, uint64_t seed, const double* q_lo, const double* q_hi,
                      const double* qd_lo, const double* qd_hi, double* q, double* qd, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q_lo || !q_hi || !qd_lo || !qd_hi || !q || !qd) return fail(HAMK_ERR_INVALID, "null box / q / qd");
  if (first_index < 0) return fail(HAMK_ERR_INVALID, "negative first_index");
  if (B == 0) return HAMK_OK;
  const std::vector<char>* code = nullptr;
  TRY(sample_code(s->cache_on, &code));                     // (before the device is looked at: hiprtc needs no GPU, so a build
  TRY(current_device_state(s));                             //  machine without one still proves the kernel compiles for gfx950)
  DevState* d = s->cur;
  if (!d->sample_fn) {
    HIP_TRY(hipModuleLoadData(&d->sample_module, code->data()));
    HIP_TRY(hipModuleGetFunction(&d->sample_fn, d->sample_module, "hamk_sample_k"));
  }
  HamkBoxes bx;
  std::memset(&bx, 0, sizeof bx);
  const int n = s->base.n;
  for (int j = 0; j < n; ++j) { bx.q_lo[j] = q_lo[j]; bx.q_hi[j] = q_hi[j]; bx.qd_lo[j] = qd_lo[j]; bx.qd_hi[j] = qd_hi[j]; }
  Stager st(s, mem);
  const size_t cnt = (size_t)n * B;
  double *xq, *xqd;
  TRY(st.out(q, cnt, &xq));
  TRY(st.out(qd, cnt, &xqd));
  long long b = B, first = first_index;
  unsigned long long sd = seed;
  int nn = n;
  void* args[] = {&xq, &xqd, &b, &first, &sd, &nn, &bx};
  const int64_t grid = (B + 255) / 256;
  if (grid > 0x7fffffffLL) return fail(HAMK_ERR_INVALID, "ensemble too large for one launch");
  HIP_TRY(hipModuleLaunchKernel(d->sample_fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, d->stream, args, nullptr));
  return st.finish();
}

int hamk_rk4_steps_checked(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t nsteps, double drift_tol,
                           int32_t* status, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p) return fail(HAMK_ERR_INVALID, "null q / p");
  if (nsteps < 0) return fail(HAMK_ERR_INVALID, "negative nsteps");
  if (drift_tol != drift_tol) return fail(HAMK_ERR_INVALID, "drift_tol is NaN");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RK4));
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  double *xq, *xp; int32_t* dst;
  TRY(st.inout(q, cnt, &xq));
  TRY(st.inout(p, cnt, &xp));
  TRY(st.out(status, (size_t)B, &dst));
  long long b = B;
  int ns = nsteps;
  void* args[] = {&xq, &xp, &b, &dt, &ns, &drift_tol, &dst};
  TRY(launch(s, K_RK4, B, args));
  return st.finish();
}

int hamk_rk4_steps(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t nsteps, int32_t* status,
                   int32_t mem) {
  return hamk_rk4_steps_checked(s, B, q, p, dt, nsteps, 0.0, status, mem);
}

static int upload_times(hamk_system* s, int32_t nt, const double* ts) {
  if ((size_t)nt > s->cur->d_ts_cap) {
    if (s->cur->d_ts) hipFree(s->cur->d_ts);
    s->cur->d_ts = nullptr; s->cur->d_ts_cap = 0;
    HIP_TRY(hipMalloc((void**)&s->cur->d_ts, sizeof(double) * nt));
    s->cur->d_ts_cap = nt;
  }
  // the previous launch may still be reading d_ts / h_ts: order behind it on the stream
  HIP_TRY(hipStreamSynchronize(s->cur->stream));
  s->cur->h_ts.assign(ts, ts + nt);
  HIP_TRY(hipMemcpyAsync(s->cur->d_ts, s->cur->h_ts.data(), sizeof(double) * nt, hipMemcpyHostToDevice, s->cur->stream));
  return HAMK_OK;
}

static const double kRefEps = 1.49012e-08;   // Hamilton.hs:448
int hamk_evolve_ham_batch(hamk_system* s, int64_t B, const double* q0, const double* p0, int32_t nt, const double* ts,
                          double* qout, double* pout, double h0, double eps_abs, double eps_rel, int32_t* status,
                          int32_t* nsub, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q0 || !p0 || !qout || !pout || !ts) return fail(HAMK_ERR_INVALID, "null q0 / p0 / ts / qout / pout");
  if (nt < 2) return fail(HAMK_ERR_INVALID, "evolveHam needs at least two times (2 <= s, Hamilton.hs:435)");
  if (B == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RKF45));
  if (!(h0 > 0.0)) h0 = (ts[1] - ts[0]) / 100.0;          // Hamilton.hs:447
  if (!(eps_abs > 0.0)) eps_abs = kRefEps;
  if (!(eps_rel > 0.0)) eps_rel = kRefEps;
  if (s->gsl_api == 2) {
    // gsl_odeiv2_driver_apply: the direction is the sign of the initial step (h > 0 ? +1 : -1) and a
    // target time on the wrong side of t is GSL_EINVAL ("integration limits and/or step direction not
    // consistent"); t equals the previous grid time exactly whenever a trajectory gets this far
    const double sgn = h0 > 0.0 ? 1.0 : -1.0;
    for (int32_t r = 1; r < nt; ++r)
      if (sgn * (ts[r] - ts[r - 1]) < 0.0)
        return fail(HAMK_ERR_INVALID, "evolveHam (gsl_odeiv2 semantics): integration limits and/or step direction not consistent "
                                      "(the time grid must be monotone in the direction of ts[1] - ts[0])");
  }
  Stager st(s, mem);
  // time grid: two times travel as kernel arguments; a longer grid as a pinned side input of a
  // small host-pointer call, else through the handle's device scratch
  double ts0 = ts[0], ts1 = ts[1];
  const double* dts = nullptr;
  if (nt > 2) {
    dts = (size_t)nt * sizeof(double) <= kPinMaxBuf ? st.side_input(ts, (size_t)nt) : nullptr;
    if (!dts) { TRY(upload_times(s, nt, ts)); dts = s->cur->d_ts; }
  }
  const size_t cnt = (size_t)s->base.n * B;
  const double *xq, *xp; double *xqo, *xpo; int32_t *dst, *dns;
  TRY(st.in(q0, cnt, (double**)&xq));
  TRY(st.in(p0, cnt, (double**)&xp));
  TRY(st.out(qout, cnt * nt, &xqo));
  TRY(st.out(pout, cnt * nt, &xpo));
  TRY(st.out(status, (size_t)B, &dst));
  TRY(st.out(nsub, (size_t)B, &dns));
  long long b = B;
  int nt_ = nt, flags = rkf_flags(0, 0, s->gsl_api), max_sub = s->max_substeps;
  int ncalls = 1, it_every = 0;
  void* args[] = {&xq, &xp, &xqo, &xpo, &b, &nt_, &dts, &ts0, &ts1, &h0, &eps_abs, &eps_rel, &flags, &max_sub, &dst, &dns, &ncalls, &it_every};
  TRY(launch(s, K_RKF45, B, args));
  return st.finish();
}

int hamk_step_ham_iterate(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t ncalls, int32_t out_every,
                          double* qout, double* pout, int32_t* status, int32_t* nsub, int32_t mem) {
  TRY(check_call(s, B, mem));
  if (!q || !p) return fail(HAMK_ERR_INVALID, "null q / p");
  if (ncalls < 0 || out_every < 0) return fail(HAMK_ERR_INVALID, "negative ncalls / out_every");
  if (out_every > 0 && (!qout || !pout)) return fail(HAMK_ERR_INVALID, "out_every > 0 needs qout / pout");
  if (B == 0 || ncalls == 0) return HAMK_OK;
  TRY(bind_device(s, B, K_RKF45));
  double ts0 = 0.0, ts1 = dt;                               // evolveHam over (0, r), Hamilton.hs:401
  double h0 = dt / 100.0, eps_abs = kRefEps, eps_rel = kRefEps;
  Stager st(s, mem);
  const size_t cnt = (size_t)s->base.n * B;
  const size_t rows = out_every > 0 ? (size_t)(ncalls / out_every) : 0;
  double *xq, *xp, *xqo = nullptr, *xpo = nullptr; int32_t *dst, *dns;
  TRY(st.inout(q, cnt, &xq));
  TRY(st.inout(p, cnt, &xp));
  if (rows > 0) { TRY(st.out(qout, cnt * rows, &xqo)); TRY(st.out(pout, cnt * rows, &xpo)); }
  TRY(st.out(status, (size_t)B, &dst));
  TRY(st.out(nsub, (size_t)B, &dns));
  long long b = B;
  // in place on (q, p); the kernel's qout/pout receive the frames (hamk_device.hpp rkf45_body, flags)
  int nt_ = 2, flags = rkf_flags(1, 2, s->gsl_api), max_sub = s->max_substeps;
  int nc = ncalls, every = rows > 0 ? out_every : 0;
  const double* dts = nullptr;
  const double *cq = xq, *cp = xp;
  void* args[] = {&cq, &cp, &xqo, &xpo, &b, &nt_, &dts, &ts0, &ts1, &h0, &eps_abs, &eps_rel, &flags, &max_sub, &dst, &dns, &nc, &every};
  TRY(launch(s, K_RKF45, B, args));
  return st.finish();
}

int hamk_step_ham_batch(hamk_system* s, int64_t B, double* q, double* p, double dt, int32_t* status, int32_t* nsub,
                        int32_t mem) {
  return hamk_step_ham_iterate(s, B, q, p, dt, 1, 0, nullptr, nullptr, status, nsub, mem);
}


// ---- ensemble checkpoint: flat binary dump / restore of the SoA state ----------
namespace {
struct CkHeader {              // 64 bytes, little endian
  char magic[8];               // "HAMKCKP1"
  int32_t n, version;
  int64_t B, steps_done;
  uint64_t seed;
  double t;
  unsigned char reserved[16];
};
static_assert(sizeof(CkHeader) == 64, "checkpoint header layout");
constexpr char kCkMagic[8] = {'H', 'A', 'M', 'K', 'C', 'K', 'P', '1'};
constexpr size_t kCkChunk = 8u << 20;   // staging chunk for device-resident state
}  // namespace

int hamk_checkpoint_write(const char* path, int32_t n, int64_t B, const double* q, const double* p, int32_t mem,
                          int64_t steps_done, uint64_t seed, double t) {
  if (!path || n <= 0 || B < 0 || (B > 0 && (!q || !p))) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_write: bad path / n / B / null state");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  CkHeader h;
  std::memset(&h, 0, sizeof h);
  std::memcpy(h.magic, kCkMagic, 8);
  h.n = n; h.version = 1; h.B = B; h.steps_done = steps_done; h.seed = seed; h.t = t;
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  std::FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_write: cannot create ") + tmp);
  Sha256 sha;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  sha.update(&h, sizeof h);
  std::vector<char> stage;
  const size_t bytes = (size_t)n * (size_t)B * sizeof(double);
  for (const double* arr : {q, p}) {
    for (size_t off = 0; off < bytes && ok; off += kCkChunk) {
      const size_t k = std::min(kCkChunk, bytes - off);
      const char* src = (const char*)arr + off;
      if (mem == HAMK_MEM_DEVICE) {
        stage.resize(k);
        hipError_t e = hipMemcpy(stage.data(), src, k, hipMemcpyDeviceToHost);    // synchronises with the null stream
        if (e != hipSuccess) { std::fclose(f); std::remove(tmp.c_str()); return fail(HAMK_ERR_HIP, std::string("hamk_checkpoint_write: ") + hipGetErrorString(e)); }
        src = stage.data();
      }
      ok = std::fwrite(src, 1, k, f) == k;
      sha.update(src, k);
    }
  }
  unsigned char dg[32];
  sha.finish(dg);
  ok = ok && std::fwrite(dg, 1, 32, f) == 32;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_write: write to ") + path + " failed"); }
  return HAMK_OK;
}

static int read_ck_header(std::FILE* f, const char* path, CkHeader* h) {
  if (std::fread(h, sizeof *h, 1, f) != 1 || std::memcmp(h->magic, kCkMagic, 8) != 0 || h->version != 1 || h->n <= 0 || h->B < 0)
    return fail(HAMK_ERR_INVALID, std::string(path) + " is not a hamk checkpoint");
  // the header must describe THIS file before anything is sized from it: 64 + 2 n B 8 + 32 bytes, without overflow
  // (a corrupt or crafted header would otherwise ask for an arbitrary allocation before the digest is looked at)
  struct stat st;
  if (::fstat(fileno(f), &st) != 0) return fail(HAMK_ERR_INVALID, std::string(path) + ": cannot stat");
  const uint64_t n = (uint64_t)h->n, B = (uint64_t)h->B, lim = (uint64_t)1 << 60;
  if (n > 64 || (B != 0 && n * 16 > lim / B) || (uint64_t)st.st_size != sizeof(CkHeader) + 2 * n * B * 8 + 32)
    return fail(HAMK_ERR_INVALID, std::string(path) + ": header does not match the file size (truncated or corrupted checkpoint)");
  return HAMK_OK;
}

int hamk_checkpoint_info(const char* path, int32_t* n, int64_t* B, int64_t* steps_done, uint64_t* seed, double* t) {
  if (!path) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_info: null path");
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_info: cannot open ") + path);
  CkHeader h;
  const int rc = read_ck_header(f, path, &h);
  std::fclose(f);
  if (rc != HAMK_OK) return rc;
  if (n) *n = h.n;
  if (B) *B = h.B;
  if (steps_done) *steps_done = h.steps_done;
  if (seed) *seed = h.seed;
  if (t) *t = h.t;
  return HAMK_OK;
}

int hamk_checkpoint_read(const char* path, int32_t n, int64_t B, double* q, double* p, int32_t mem) {
  if (!path || (B > 0 && (!q || !p))) return fail(HAMK_ERR_INVALID, "hamk_checkpoint_read: null path / state");
  if (mem != HAMK_MEM_HOST && mem != HAMK_MEM_DEVICE) return fail(HAMK_ERR_INVALID, "mem must be HAMK_MEM_HOST or HAMK_MEM_DEVICE");
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return fail(HAMK_ERR_INVALID, std::string("hamk_checkpoint_read: cannot open ") + path);
  CkHeader h;
  int rc = read_ck_header(f, path, &h);
  if (rc == HAMK_OK && (h.n != n || h.B != B)) rc = fail(HAMK_ERR_INVALID, "hamk_checkpoint_read: the file holds a different ensemble (n, B)");
  if (rc != HAMK_OK) { std::fclose(f); return rc; }
  Sha256 sha;
  sha.update(&h, sizeof h);
  const size_t bytes = (size_t)n * (size_t)B * sizeof(double);
  // the digest is checked BEFORE anything reaches the caller's arrays (host or device)
  std::vector<char> all(2 * bytes);
  bool ok = bytes == 0 || std::fread(all.data(), 1, 2 * bytes, f) == 2 * bytes;
  unsigned char want[32], got[32];
  ok = ok && std::fread(want, 1, 32, f) == 32;
  std::fclose(f);
  if (ok) { sha.update(all.data(), 2 * bytes); sha.finish(got); ok = std::memcmp(want, got, 32) == 0; }
  if (!ok) return fail(HAMK_ERR_INVALID, std::string(path) + ": truncated or corrupted checkpoint (digest mismatch)");
  if (bytes == 0) return HAMK_OK;
  if (mem == HAMK_MEM_HOST) {
    std::memcpy(q, all.data(), bytes);
    std::memcpy(p, all.data() + bytes, bytes);
  } else {
    HIP_TRY(hipMemcpy(q, all.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p, all.data() + bytes, bytes, hipMemcpyHostToDevice));
  }
  return HAMK_OK;
}

}  // extern "C"

// qe_jit.cpp -- hiprtc compile + code-object cache for the generated fused kernels.
//
// The analogue of BytecodeCompiler.compile's class definition step
// (evaluator/BytecodeCompiler.kt:15-20,171-174): generated code is compiled
// once per distinct plan, kept in memory, and persisted as .hsaco files in the
// JIT cache directory (the reference dumps its generated classes to
// target/classes, :124-126,167-169).  hiprtc cross-compiles for gfx950 without
// a device, so build() can pre-populate the cache for prepared plans.
#include "qe_internal.h"

#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <fstream>

namespace qe {

static uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

static const char *kArch = "gfx950";
// part of the cache key: bump when the compile options below change
static const char *kOptionsTag = "O3-nocontract-noatomicopt-v2";

std::vector<char> Jit::compile(const std::string &source) {
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, source.c_str(), "qe_fused.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) fail(QE_ERR_HIP, std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r));
    // -ffp-contract=off: the JVM's DMUL;DADD are separately rounded
    // -amdgpu-atomic-optimizer-strategy=None: every atomic of the generated kernels is issued by ONE lane and
    // its result is consumed much later; the optimizer's wave-reduction form reads the result right away
    // (s_waitcnt vmcnt(0) + v_readfirstlane), which exposed the full round trip of every ticket
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                          "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"};
    r = hiprtcCompileProgram(prog, 6, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        fail(QE_ERR_HIP, std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + log);
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    return code;
}

// Scratch (register spill) bytes per lane of a code object = the LARGEST of its kernels': the msgpack metadata note carries
// ".private_segment_fixed_size" followed by a small unsigned integer once per kernel (the module of a plan holds several
// entry points -- the probe, the single-pass kernel, the local form's scan ... -- and none of them may spill).
int Jit::scratch_bytes(const std::vector<char> &code) {
    static const char key[] = ".private_segment_fixed_size";
    const size_t kl = sizeof(key) - 1;
    int worst = -1;
    for (size_t i = 0; i + kl + 5 < code.size(); i++) {
        if (std::memcmp(code.data() + i, key, kl) != 0) continue;
        const unsigned char *p = (const unsigned char *)code.data() + i + kl;
        int v = -1;
        if (p[0] < 0x80) v = p[0];
        else if (p[0] == 0xcc) v = p[1];
        else if (p[0] == 0xcd) v = (p[1] << 8) | p[2];
        else if (p[0] == 0xce) v = (int)(((unsigned)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4]);
        worst = std::max(worst, v);
        i += kl;
    }
    return worst;
}

// the compiler is part of a code object's identity: a cache directory that outlives a ROCm update must not serve stale code
static std::string toolchain_tag() {
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    return std::string(kOptionsTag) + "-rtc" + std::to_string(major) + "." + std::to_string(minor);
}

static bool looks_like_code_object(const std::vector<char> &code) {
    return code.size() > 64 && std::memcmp(code.data(), "\x7f" "ELF", 4) == 0;
}

std::string Jit::source_key(const std::string &source) {
    char key[64];
    std::snprintf(key, sizeof key, "%016llx_%zu_%s", (unsigned long long)fnv1a(source + toolchain_tag()), source.size(), kArch);
    return key;
}

// Measured decisions about a plan (which kernel geometry won) are kept next to its code object, so that a new context /
// a new process runs the same geometry without exploring again.
int Jit::load_choice(const std::string &source, double *margin) const {
    if (margin) *margin = 1.0;
    if (cache_dir_.empty()) return -1;
    std::ifstream f(cache_dir_ + "/" + source_key(source) + ".geo");
    int v = -1;
    if (f && (f >> v) && v >= 0 && v <= 2) {
        // second line: "default <a> ms, wide <b> ms, mid <c> ms (...)": how far the winner was ahead of the runner-up
        std::string word;
        double t[3] = {0, 0, 0};
        if (margin && (f >> word >> t[0] >> word >> word >> t[1] >> word >> word >> t[2]) && t[0] > 0 && t[1] > 0 && t[2] > 0) {
            std::sort(t, t + 3);
            *margin = (t[1] - t[0]) / t[0];
        } else if (margin) {
            *margin = 0.0;   // a note of an older layout: measure again
        }
        return v;
    }
    return -1;
}

void Jit::store_choice(const std::string &source, int chosen, const std::string &note) const {
    if (cache_dir_.empty()) return;
    ::mkdir(cache_dir_.c_str(), 0777);
    const std::string path = cache_dir_ + "/" + source_key(source) + ".geo";
    const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
    std::ofstream f(tmp);
    if (!f) return;
    f << chosen << "\n" << note << "\n";
    f.close();
    if (f.good()) std::rename(tmp.c_str(), path.c_str());
    else std::remove(tmp.c_str());
}

// A code object the plan builder REJECTED (it spills: get_plan lowers the occupancy request / the sub-tile and generates
// again) does not stay in the cache -- a cache that keeps rejected objects invites a wrong hit after a key change.  What
// stays is a marker with the measured scratch size, so that the next process learns the verdict without compiling again.
void Jit::reject(const std::string &source, int scratch) {
    const std::string key = source_key(source);
    auto it = loaded_.find(key);
    if (it != loaded_.end()) {
        if (it->second.module) (void)hipModuleUnload(it->second.module);
        loaded_.erase(it);
    }
    if (cache_dir_.empty()) return;
    const std::string base = cache_dir_ + "/" + key;
    std::remove((base + ".hsaco").c_str());
    std::remove((base + ".hip").c_str());
    std::remove((base + ".geo").c_str());
    const std::string tmp = base + ".rej.tmp" + std::to_string((long)::getpid());
    std::ofstream f(tmp);
    if (!f) return;
    f << scratch << "\n";
    f.close();
    if (f.good()) std::rename(tmp.c_str(), (base + ".rej").c_str());
    else std::remove(tmp.c_str());
}

Kernel Jit::get(const std::string &source, const char *entry, bool load) {
    const std::string key = source_key(source);
    auto it = loaded_.find(key);
    if (it != loaded_.end()) {
        mem_hits++;
        last_scratch = it->second.scratch;
        return it->second;
    }
    if (!load && !cache_dir_.empty()) {   // a verdict of an earlier process: this source spills
        std::ifstream rj(cache_dir_ + "/" + key + ".rej");
        int sc = 0;
        if (rj && (rj >> sc) && sc > 0) {
            Kernel k;
            k.scratch = sc;
            last_scratch = sc;
            disk_hits++;
            return k;
        }
    }
    std::vector<char> code;
    std::string path;
    if (!cache_dir_.empty()) {
        path = cache_dir_ + "/" + key + ".hsaco";
        std::ifstream f(path, std::ios::binary);
        if (f) {
            code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            if (looks_like_code_object(code)) disk_hits++;
            else code.clear();   // truncated / foreign file: compile again (and overwrite it)
        }
    }
    bool from_disk = !code.empty();
    auto compile_and_store = [&]() {
        code = compile(source);
        compiles++;
        from_disk = false;
        if (path.empty()) return;
        ::mkdir(cache_dir_.c_str(), 0777);
        const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return;
        f.write(code.data(), (std::streamsize)code.size());
        f.close();
        if (!f.good()) {   // e.g. ENOSPC: never leave a truncated code object behind
            std::remove(tmp.c_str());
            return;
        }
        std::rename(tmp.c_str(), path.c_str());
        std::ofstream src(path.substr(0, path.size() - 6) + ".hip");
        if (src) src << source;
    };
    if (code.empty()) compile_and_store();
    Kernel k;
    k.scratch = scratch_bytes(code);
    last_scratch = k.scratch;
    if (!load) return k;
    hipError_t le = hipModuleLoadData(&k.module, code.data());
    if (le != hipSuccess && from_disk) {   // a cached file the runtime rejects: drop it and compile once
        (void)hipGetLastError();
        std::remove(path.c_str());
        compile_and_store();
        k.scratch = scratch_bytes(code);
        last_scratch = k.scratch;
        le = hipModuleLoadData(&k.module, code.data());
    }
    QE_HIP(le);
    QE_HIP(hipModuleGetFunction(&k.fn, k.module, entry));
    loaded_[key] = k;
    return k;
}

Jit::~Jit() {
    for (auto &kv : loaded_)
        if (kv.second.module) (void)hipModuleUnload(kv.second.module);
}

}  // namespace qe

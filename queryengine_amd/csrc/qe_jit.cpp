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

// Scratch (register spill) bytes per lane of the first kernel of a code object: the msgpack metadata
// note carries ".private_segment_fixed_size" followed by a small unsigned integer.
int Jit::scratch_bytes(const std::vector<char> &code) {
    static const char key[] = ".private_segment_fixed_size";
    const size_t kl = sizeof(key) - 1;
    for (size_t i = 0; i + kl + 5 < code.size(); i++) {
        if (std::memcmp(code.data() + i, key, kl) != 0) continue;
        const unsigned char *p = (const unsigned char *)code.data() + i + kl;
        if (p[0] < 0x80) return p[0];
        if (p[0] == 0xcc) return p[1];
        if (p[0] == 0xcd) return (p[1] << 8) | p[2];
        if (p[0] == 0xce) return (int)(((unsigned)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4]);
        return -1;
    }
    return -1;
}

Kernel Jit::get(const std::string &source, const char *entry, bool load) {
    char key[64];
    std::snprintf(key, sizeof key, "%016llx_%zu_%s", (unsigned long long)fnv1a(source + kOptionsTag), source.size(), kArch);
    auto it = loaded_.find(key);
    if (it != loaded_.end()) {
        mem_hits++;
        last_scratch = it->second.scratch;
        return it->second;
    }
    std::vector<char> code;
    std::string path;
    if (!cache_dir_.empty()) {
        path = cache_dir_ + "/" + key + ".hsaco";
        std::ifstream f(path, std::ios::binary);
        if (f) {
            code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            if (!code.empty()) disk_hits++;
        }
    }
    if (code.empty()) {
        code = compile(source);
        compiles++;
        if (!path.empty()) {
            ::mkdir(cache_dir_.c_str(), 0777);
            std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
            std::ofstream f(tmp, std::ios::binary);
            if (f) {
                f.write(code.data(), (std::streamsize)code.size());
                f.close();
                std::rename(tmp.c_str(), path.c_str());
                std::ofstream src(path.substr(0, path.size() - 6) + ".hip");
                if (src) src << source;
            }
        }
    }
    Kernel k;
    k.scratch = scratch_bytes(code);
    last_scratch = k.scratch;
    if (!load) return k;
    QE_HIP(hipModuleLoadData(&k.module, code.data()));
    QE_HIP(hipModuleGetFunction(&k.fn, k.module, entry));
    loaded_[key] = k;
    return k;
}

Jit::~Jit() {
    for (auto &kv : loaded_)
        if (kv.second.module) (void)hipModuleUnload(kv.second.module);
}

}  // namespace qe

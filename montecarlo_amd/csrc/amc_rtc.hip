// amc_rtc.hip -- kernels compiled at run time: script-defined potentials, rewards, policies, actions and the Float32 state
// (amc_create_custom / _model / _policy_model / _action_model / _vector_policy_model / _mixed_model).  hiprtc is resolved with
// dlopen; the kernel sources travel in the library as string literals (amc_rtc_sources.gen.h, made by embed_sources.py).
#define AMC_KERNEL_LINKAGE static      // the kernel headers are included for their types only: no kernel of theirs in this object
#include "amc_internal.h"
#include "amc_rtc_sources.gen.h"

#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <spawn.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <cerrno>

extern char** environ;

namespace {

// ---- kernels compiled at run time for a user-defined potential (AMC_POTENTIAL_CUSTOM) ---------------------------
// `potential` is a free function of the driver script in the reference (MC_harmonic_oscillator.jl:4); here it is a
// C expression in `x`, and the templates of amc_kernels.h are instantiated for it by hiprtc (resolved with dlopen,
// like RCCL: no link-time dependency).  One hiprtc program per kernel instantiation, compiled on first use
// (~1 s each) and cached per process by (expression, instantiation); modules are loaded per handle (= per device).
struct Hiprtc {
    void* lib = nullptr;
    int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*AddNameExpression)(void*, const char*) = nullptr;
    int (*CompileProgram)(void*, int, const char**) = nullptr;
    int (*GetProgramLogSize)(void*, size_t*) = nullptr;
    int (*GetProgramLog)(void*, char*) = nullptr;
    int (*GetCodeSize)(void*, size_t*) = nullptr;
    int (*GetCode)(void*, char*) = nullptr;
    int (*GetLoweredName)(void*, const char*, const char**) = nullptr;
    int (*DestroyProgram)(void**) = nullptr;
    int (*Version)(int*, int*) = nullptr;
};

std::mutex g_rtc_mu;
Hiprtc g_hiprtc;
std::map<std::string, RtcCode> g_rtc_code;        // key: expression '\n' instantiation
std::map<std::string, std::string> g_rtc_broken;  // same key: instantiations the compiler DIED on (its last words), not to be asked for again

int load_hiprtc(Hiprtc& r)
{
    if (r.lib) return AMC_OK;
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return fail(AMC_ERR_HIP, "custom potential: cannot dlopen libhiprtc: %s", dlerror());
#define AMC_RTC_SYM(field, name)                                                        \
    r.field = (decltype(r.field))dlsym(r.lib, name);                                    \
    if (!r.field) { r.lib = nullptr; return fail(AMC_ERR_HIP, "libhiprtc is missing %s", name); }
    AMC_RTC_SYM(CreateProgram, "hiprtcCreateProgram");
    AMC_RTC_SYM(AddNameExpression, "hiprtcAddNameExpression");
    AMC_RTC_SYM(CompileProgram, "hiprtcCompileProgram");
    AMC_RTC_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
    AMC_RTC_SYM(GetProgramLog, "hiprtcGetProgramLog");
    AMC_RTC_SYM(GetCodeSize, "hiprtcGetCodeSize");
    AMC_RTC_SYM(GetCode, "hiprtcGetCode");
    AMC_RTC_SYM(GetLoweredName, "hiprtcGetLoweredName");
    AMC_RTC_SYM(DestroyProgram, "hiprtcDestroyProgram");
    AMC_RTC_SYM(Version, "hiprtcVersion");
#undef AMC_RTC_SYM
    return AMC_OK;
}

}  // namespace

// The expression becomes the body of a function-like macro: keep it to one line of ordinary expression text.
int validate_potential_expr(const char* expr, const char* what, const char* var)
{
    if (!expr) return fail(AMC_ERR_BAD_ARG, "%s: expression is NULL", what);
    const size_t n = std::strlen(expr);
    if (n == 0 || n > 4000) return fail(AMC_ERR_BAD_ARG, "%s: expression must have 1..4000 characters", what);
    bool has_x = false;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)expr[i];
        if (c < 0x20 || c > 0x7e || c == '#' || c == '\\' || c == ';' || c == '{' || c == '}' || c == '"' || c == '\'' ||
            c == '`' || c == '$' || c == '@')
            return fail(AMC_ERR_BAD_ARG, "%s: character 0x%02x at offset %zu is not allowed in the expression", what, c, i);
        const bool ident_before = i > 0 && (std::isalnum((unsigned char)expr[i - 1]) || expr[i - 1] == '_');
        const size_t vl = std::strlen(var);
        if (!ident_before && std::strncmp(expr + i, var, vl) == 0 &&
            !(i + vl < n && (std::isalnum((unsigned char)expr[i + vl]) || expr[i + vl] == '_')))
            has_x = true;
    }
    if (!has_x && var[0] != 0) return fail(AMC_ERR_BAD_ARG, "%s: the expression does not mention %s", what, var);
    return AMC_OK;
}

namespace {

// Optional on-disk cache of compiled code objects (AMC_RTC_CACHE_DIR; unset = in-process cache only): one file per
// (expression, instantiation, kernel sources), named by a 64-bit FNV-1a hash of all three, holding the lowered name
// and the code object.  A corrupt or truncated file is ignored and recompiled.
uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

std::string rtc_cache_path(const std::string& expr, const std::string& inst, const std::string& arch, const std::string& toolchain)
{
    const char* dir = std::getenv("AMC_RTC_CACHE_DIR");
    if (!dir || !*dir) return std::string();
    uint64_t h = fnv1a(expr);
    h = fnv1a(inst, h ^ 0x9E3779B97F4A7C15ull);
    h = fnv1a(arch, h ^ 0xC2B2AE3D27D4EB4Full);          // a code object is good for one ISA ...
    h = fnv1a(toolchain, h);                              // ... and one compiler release
    for (int i = 0; i < AMC_RTC_N_SOURCES; ++i) h = fnv1a(AMC_RTC_SOURCE_TEXTS[i], h);      // every kernel source (embed_sources.py)
    char name[64];
    std::snprintf(name, sizeof(name), "/amc_rtc_%016llx.bin", (unsigned long long)h);
    return std::string(dir) + name;
}

bool rtc_cache_load(const std::string& path, RtcCode* out)
{
    if (path.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    uint64_t hdr[3] = {0, 0, 0};                       // magic, name length, code length
    bool ok = std::fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] == 0x31435452434d41ull && hdr[1] > 0 && hdr[1] < 4096 &&
              hdr[2] > 0 && hdr[2] < (1ull << 30);
    if (ok) {
        out->lowered.resize((size_t)hdr[1]);
        out->code.resize((size_t)hdr[2]);
        ok = std::fread(&out->lowered[0], 1, (size_t)hdr[1], f) == hdr[1] &&
             std::fread(out->code.data(), 1, (size_t)hdr[2], f) == hdr[2] && std::fgetc(f) == EOF;
    }
    std::fclose(f);
    return ok;
}

void rtc_cache_store(const std::string& path, const RtcCode& rc)
{
    if (path.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;                                    // an unwritable cache directory is not an error
    const uint64_t hdr[3] = {0x31435452434d41ull, rc.lowered.size(), rc.code.size()};
    const bool ok = std::fwrite(hdr, sizeof(hdr), 1, f) == 1 && std::fwrite(rc.lowered.data(), 1, rc.lowered.size(), f) == rc.lowered.size() &&
                    std::fwrite(rc.code.data(), 1, rc.code.size(), f) == rc.code.size();
    std::fclose(f);
    if (ok) std::rename(tmp.c_str(), path.c_str()); else std::remove(tmp.c_str());   // atomic publish
}


// ---- where the compiler runs ------------------------------------------------------------------------------------------------
// What one build attempt gave: stage 0 = a code object; 1 / 2 / 3 / 4 = the hiprtc call that failed (create, name expression,
// compile -- the script's own errors, log attached --, no code), with hiprtc's status.
struct Built { int stage = -1; int status = 0; std::string log, lowered; std::vector<char> code;
               bool died = false; };      // died: the compiler process ended by a signal or a failure status -- what it will do again for this input

// The compile inside THIS process: a developer knob (AMC_RTC_IN_PROCESS=1, e.g. under a debugger).  A fatal error of the compiler
// then is a fatal error of the host -- which is why it is not the default (build_in_child).
int build_in_process(const std::string& src, const std::string& inst, const std::vector<std::string>& opts, Built* out)
{
    { const int rc = load_hiprtc(g_hiprtc); if (rc != AMC_OK) return rc; }
    void* prog = nullptr;
    int e = g_hiprtc.CreateProgram(&prog, src.c_str(), "amc_custom_potential.hip", AMC_RTC_N_SOURCES, AMC_RTC_SOURCE_TEXTS, AMC_RTC_SOURCE_NAMES);
    if (e != 0) { out->stage = 1; out->status = e; return AMC_OK; }
    e = g_hiprtc.AddNameExpression(prog, inst.c_str());
    if (e != 0) { g_hiprtc.DestroyProgram(&prog); out->stage = 2; out->status = e; return AMC_OK; }
    std::vector<const char*> op;
    for (const auto& o : opts) op.push_back(o.c_str());
    e = g_hiprtc.CompileProgram(prog, (int)op.size(), op.data());
    size_t ls = 0;
    if (g_hiprtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
        out->log.resize(ls);
        g_hiprtc.GetProgramLog(prog, &out->log[0]);
        while (!out->log.empty() && out->log.back() == 0) out->log.pop_back();
    }
    if (e != 0) { g_hiprtc.DestroyProgram(&prog); out->stage = 3; out->status = e; return AMC_OK; }
    size_t cs = 0;
    const char* lowered = nullptr;
    if (g_hiprtc.GetCodeSize(prog, &cs) != 0 || cs == 0 || g_hiprtc.GetLoweredName(prog, inst.c_str(), &lowered) != 0 || !lowered) {
        g_hiprtc.DestroyProgram(&prog);
        out->stage = 4;
        return AMC_OK;
    }
    out->code.resize(cs);
    g_hiprtc.GetCode(prog, out->code.data());
    out->lowered = lowered;
    g_hiprtc.DestroyProgram(&prog);
    out->stage = 0;
    return AMC_OK;
}

// The compile in a short-lived CHILD process (amc_rtc_worker, built beside libamc.so from amc_rtc_worker.cpp) that never touches
// the GPU: request over a socket pair, answer back over it, the compiler's stderr over a pipe.  The child is a new program started
// with posix_spawn -- not a fork of this process's HIP state, never an exec of the engine's process.  A child that dies (LLVM's
// fatal errors end in abort()), answers nonsense or does not come back within AMC_RTC_TIMEOUT_S seconds (default 600) is
// AMC_ERR_COMPILE with the tail of its stderr in amc_last_error(); the host lives.  Reference convention: a model that cannot be
// used raises, src/metropolis.jl:35.
const uint64_t WORKER_MAGIC_REQUEST = 0x3151435452434d41ull;      // "AMCRTCQ1" (amc_rtc_worker.cpp)
const uint64_t WORKER_MAGIC_ANSWER = 0x3141435452434d41ull;       // "AMCRTCA1"

std::string worker_path()
{
    if (const char* env = std::getenv("AMC_RTC_WORKER")) return env;
    Dl_info info;
    if (dladdr((const void*)&worker_path, &info) == 0 || !info.dli_fname) return std::string();
    std::string dir = info.dli_fname;
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    return dir + "/amc_rtc_worker";
}

void put_blob(std::string* req, const void* p, size_t n)
{
    const uint64_t len = n;
    req->append((const char*)&len, sizeof(len));
    req->append((const char*)p, n);
}

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int build_in_child(const std::string& src, const std::string& inst, const std::vector<std::string>& opts, Built* out)
{
    const std::string exe = worker_path();
    if (exe.empty() || access(exe.c_str(), X_OK) != 0)
        return fail(AMC_ERR_COMPILE, "run-time kernel build: the compiler program %s is missing or not executable (built by montecarlo_amd/csrc/Makefile "
                                     "beside libamc.so; AMC_RTC_WORKER names another place)", exe.empty() ? "amc_rtc_worker" : exe.c_str());
    std::string req;
    req.reserve(512u << 10);
    const uint32_t counts[2] = {(uint32_t)opts.size(), (uint32_t)AMC_RTC_N_SOURCES};
    req.append((const char*)&WORKER_MAGIC_REQUEST, sizeof(WORKER_MAGIC_REQUEST));
    req.append((const char*)counts, sizeof(counts));
    put_blob(&req, src.data(), src.size());
    put_blob(&req, "amc_custom_potential.hip", 24);
    put_blob(&req, inst.data(), inst.size());
    for (const auto& o : opts) put_blob(&req, o.data(), o.size());
    for (int i = 0; i < AMC_RTC_N_SOURCES; ++i) put_blob(&req, AMC_RTC_SOURCE_NAMES[i], std::strlen(AMC_RTC_SOURCE_NAMES[i]));
    for (int i = 0; i < AMC_RTC_N_SOURCES; ++i) put_blob(&req, AMC_RTC_SOURCE_TEXTS[i], std::strlen(AMC_RTC_SOURCE_TEXTS[i]));

    double timeout_s = 600.0;
    if (const char* env = std::getenv("AMC_RTC_TIMEOUT_S")) { const double v = std::atof(env); if (v > 0.0) timeout_s = v; }

    int sv[2] = {-1, -1}, ep[2] = {-1, -1};
    if (socketpair(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0, sv) != 0) return fail(AMC_ERR_COMPILE, "run-time kernel build: socketpair: %s", std::strerror(errno));
    if (pipe2(ep, O_CLOEXEC) != 0) { close(sv[0]); close(sv[1]); return fail(AMC_ERR_COMPILE, "run-time kernel build: pipe: %s", std::strerror(errno)); }
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, sv[1], 0);        // (dup2 clears close-on-exec on the copies; everything else of ours has it set)
    posix_spawn_file_actions_adddup2(&fa, sv[1], 1);
    posix_spawn_file_actions_adddup2(&fa, ep[1], 2);
    // the child's environment: this process's, minus what would load a profiler's tool library into the compiler (rocprofv3 preloads
    // one into every process it starts: the compile is not part of anybody's kernel trace, and the tool initialises the GPU)
    std::vector<std::string> env_keep;
    for (char** e = environ; e && *e; ++e) {
        const std::string kv = *e;
        if (kv.rfind("LD_PRELOAD=", 0) == 0) {
            std::string kept;
            size_t from = 11;
            while (from <= kv.size()) {
                size_t to = kv.find_first_of(": ", from);
                if (to == std::string::npos) to = kv.size();
                const std::string one = kv.substr(from, to - from);
                if (!one.empty() && one.find("rocprof") == std::string::npos) kept += (kept.empty() ? "" : ":") + one;
                from = to + 1;
            }
            if (!kept.empty()) env_keep.push_back("LD_PRELOAD=" + kept);
            continue;
        }
        if (kv.rfind("HSA_TOOLS_LIB=", 0) == 0 || kv.rfind("ROCP_TOOL_LIBRARIES=", 0) == 0 || kv.rfind("ROCPROFILER_", 0) == 0) continue;
        env_keep.push_back(kv);
    }
    std::vector<char*> envp;
    for (auto& kv : env_keep) envp.push_back(&kv[0]);
    envp.push_back(nullptr);
    char* const argv[] = {const_cast<char*>(exe.c_str()), nullptr};
    pid_t pid = -1;
    const int se = posix_spawn(&pid, exe.c_str(), &fa, nullptr, argv, envp.data());
    posix_spawn_file_actions_destroy(&fa);
    close(sv[1]);
    close(ep[1]);
    if (se != 0) { close(sv[0]); close(ep[0]); return fail(AMC_ERR_COMPILE, "run-time kernel build: cannot start %s: %s", exe.c_str(), std::strerror(se)); }

    // feed the request, collect answer and stderr; one deadline for the lot
    fcntl(sv[0], F_SETFL, fcntl(sv[0], F_GETFL) | O_NONBLOCK);
    fcntl(ep[0], F_SETFL, fcntl(ep[0], F_GETFL) | O_NONBLOCK);
    const double deadline = now_s() + timeout_s;
    std::string ans, err;
    size_t sent = 0;
    bool out_open = true, err_open = true, timed_out = false;
    while (out_open || err_open) {
        const double left = deadline - now_s();
        if (left <= 0.0) { timed_out = true; break; }
        pollfd fds[2] = {{sv[0], (short)(POLLIN | (sent < req.size() ? POLLOUT : 0)), 0}, {ep[0], POLLIN, 0}};
        if (!out_open) fds[0].fd = -1;
        if (!err_open) fds[1].fd = -1;
        const int pr = poll(fds, 2, (int)std::min(left * 1000.0 + 1.0, 1000.0));
        if (pr < 0 && errno != EINTR) break;
        if (pr <= 0) continue;
        if (out_open && sent < req.size() && (fds[0].revents & POLLOUT)) {
            const ssize_t put = send(sv[0], req.data() + sent, req.size() - sent, MSG_NOSIGNAL);       // (a dead child: EPIPE, not SIGPIPE)
            if (put > 0) sent += (size_t)put;
            else if (put < 0 && errno != EAGAIN && errno != EINTR) sent = req.size();               // nothing more to say to it
            if (sent == req.size()) shutdown(sv[0], SHUT_WR);
        }
        char buf[65536];
        if (out_open && (fds[0].revents & (POLLIN | POLLHUP | POLLERR))) {
            const ssize_t got = recv(sv[0], buf, sizeof(buf), 0);
            if (got > 0) { if (ans.size() < (1u << 30)) ans.append(buf, (size_t)got); }
            else if (got == 0 || (errno != EAGAIN && errno != EINTR)) out_open = false;
        }
        if (err_open && (fds[1].revents & (POLLIN | POLLHUP | POLLERR))) {
            const ssize_t got = read(ep[0], buf, sizeof(buf));
            if (got > 0) { err.append(buf, (size_t)got); if (err.size() > (1u << 20)) err.erase(0, err.size() - (1u << 19)); }
            else if (got == 0 || (errno != EAGAIN && errno != EINTR)) err_open = false;
        }
    }
    close(sv[0]);
    close(ep[0]);
    int status = 0;
    if (timed_out) kill(pid, SIGKILL);
    else {
        // both ends closed: the child is on its way out -- give it until the deadline, then stop waiting for it
        while (waitpid(pid, &status, WNOHANG) == 0) {
            if (now_s() > deadline) { timed_out = true; kill(pid, SIGKILL); break; }
            usleep(2000);
        }
    }
    if (timed_out) {
        while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
        return fail(AMC_ERR_COMPILE, "run-time kernel build of %s: the compiler did not come back within %.0f s (AMC_RTC_TIMEOUT_S) and was stopped", inst.c_str(), timeout_s);
    }
    // the tail of what the compiler said on its way down, on one line
    auto tail = [&]() {
        std::string t = err.size() > 900 ? err.substr(err.size() - 900) : err;
        while (!t.empty() && (t.back() == '\n' || t.back() == ' ')) t.pop_back();
        return t.empty() ? std::string("(nothing on its stderr)") : t;
    };
    out->died = WIFSIGNALED(status) || !WIFEXITED(status) || WEXITSTATUS(status) != 0;
    if (WIFSIGNALED(status))
        return fail(AMC_ERR_COMPILE, "run-time kernel build of %s: the compiler died (signal %d, %s): %s", inst.c_str(), WTERMSIG(status), strsignal(WTERMSIG(status)), tail().c_str());
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
        return fail(AMC_ERR_COMPILE, "run-time kernel build of %s: the compiler process ended with status %d: %s", inst.c_str(), WIFEXITED(status) ? WEXITSTATUS(status) : -1, tail().c_str());
    // parse the answer
    size_t at = 0;
    auto take = [&](void* dst, size_t n) { if (ans.size() - at < n) return false; std::memcpy(dst, ans.data() + at, n); at += n; return true; };
    auto take_blob = [&](std::string* dst) {
        uint64_t n = 0;
        if (!take(&n, sizeof(n)) || ans.size() - at < n) return false;
        dst->assign(ans.data() + at, (size_t)n);
        at += (size_t)n;
        return true;
    };
    uint64_t magic = 0;
    int32_t head[2] = {0, 0};
    std::string code;
    if (!take(&magic, sizeof(magic)) || magic != WORKER_MAGIC_ANSWER || !take(head, sizeof(head)) || !take_blob(&out->log) || !take_blob(&out->lowered) ||
        !take_blob(&code) || at != ans.size() || head[0] < 0 || head[0] > 4)
        return fail(AMC_ERR_COMPILE, "run-time kernel build of %s: the compiler process returned %zu bytes that are not an answer: %s", inst.c_str(), ans.size(), tail().c_str());
    out->stage = head[0];
    out->status = head[1];
    out->code.assign(code.begin(), code.end());
    return AMC_OK;
}

}  // namespace

// Compiles (or finds) the code object holding ONE instantiation, e.g. "amc::sweep_kernel<2,false,false,false,true,false>".
// Needs no device.  On a compile error the hiprtc log goes into the error message (and *log_out).
int rtc_compile(const std::string& expr_in, const std::string& inst, const std::string& arch, const RtcCode** out, std::string* log_out)
{
    std::lock_guard<std::mutex> lock(g_rtc_mu);
    const std::string key = arch + "\n" + expr_in + "\n" + inst + (std::getenv("AMC_NO_GAUSS_CLASS_ROWS") ? "\nno-gauss-rows" : "") +
                            (std::getenv("AMC_NO_SIGMA_MEMO") ? "\nno-sigma-memo" : "") +
                            (std::getenv("AMC_RTC_WAVES") ? std::string("\nwaves") + std::getenv("AMC_RTC_WAVES") : std::string());
    auto it = g_rtc_code.find(key);
    if (it != g_rtc_code.end()) { *out = &it->second; return AMC_OK; }
    {
        auto bk = g_rtc_broken.find(key);
        if (bk != g_rtc_broken.end()) return fail(AMC_ERR_COMPILE, "%s", bk->second.c_str());
    }
    { const int rc = load_hiprtc(g_hiprtc); if (rc != AMC_OK) return rc; }
    int rtc_major = 0, rtc_minor = 0;
    (void)g_hiprtc.Version(&rtc_major, &rtc_minor);
    // the K > 1 fused sweep + estimator kernels are built with Machine LICM off, like their offline twins (amc_pg_fused.hip),
    // (decided from the instantiation's FOURTH template argument, SWEEP == 2 -- `<POT, NL, BETA, SWEEP, REDUCE, MIDFLUSH>`: a
    // substring test would also catch NL = 2 followed by BETA)
    const bool licm_off = [&] {
        // AMC_RTC_LICM=all-off / est-off / off-for-none (developer knob, A/B): every form, every estimator form, no form
        if (const char* env = std::getenv("AMC_RTC_LICM")) {
            const std::string v = env;
            if (v == "all-off") return true;
            if (v == "off-for-none") return false;
            if (v == "est-off") return inst.rfind("amc::pg_estimate_kernel<", 0) == 0;
        }
        const std::string head = "amc::pg_estimate_kernel<";
        if (inst.rfind(head, 0) != 0) return false;
        size_t at = head.size();
        for (int arg = 0; arg < 3; ++arg) {
            at = inst.find(',', at);
            if (at == std::string::npos) return false;
            ++at;
        }
        const size_t end = inst.find_first_of(",>", at);
        if (end != std::string::npos && inst.substr(at, end - at) == "2") return true;
        // ... and the estimator forms of policies with several parameters ('\x0e' section of the expression): with 8 to 19 two-level
        // accumulator columns live, what the pass hoists costs registers the kernel does not have (two-parameter fused step: 133 -> 123
        // VGPRs, 66 -> 20 scalar spills, 151 -> 144 us at 1e7 chains; the one-parameter forms are indifferent)
        return expr_in.find('\x0e') != std::string::npos;
    }();
    const std::string cache_file = rtc_cache_path(expr_in, inst, arch, "hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) +
                                                                           (licm_off ? " licm-off" : "") +
                                                                           (std::getenv("AMC_NO_GAUSS_CLASS_ROWS") ? " no-gauss-rows" : "") +
                                                                           (std::getenv("AMC_NO_SIGMA_MEMO") ? " no-sigma-memo" : "") +
                                                                           (std::getenv("AMC_RTC_WAVES") ? std::string(" waves") + std::getenv("AMC_RTC_WAVES") : std::string()));
    {
        RtcCode cached;
        if (rtc_cache_load(cache_file, &cached)) {
            if (log_out) log_out->clear();
            *out = &g_rtc_code.emplace(key, std::move(cached)).first->second;
            return AMC_OK;
        }
        // ... or the note that this compiler release dies on this instantiation (the file's name carries the release)
        if (!cache_file.empty()) {
            if (FILE* f = std::fopen((cache_file + ".broken").c_str(), "rb")) {
                char msg[1600];
                const size_t n = std::fread(msg, 1, sizeof(msg) - 1, f);
                std::fclose(f);
                msg[n] = 0;
                if (n > 0) {
                    g_rtc_broken[key] = msg;
                    return fail(AMC_ERR_COMPILE, "%s", msg);
                }
            }
        }
    }
    // expr_in = [ '\x02' (Float32 state) ] [ potential [ '\x01' reward ] [ '\x03' scale ] [ '\x04' sample '\x05' logq [ '\x06' dlogq ]
    //             [ '\x07' perform ] [ '\x08' invert ] ] ]
    const bool f32 = !expr_in.empty() && expr_in[0] == '\x02';
    const std::string expr_full = expr_in.substr(f32 ? 1 : 0);
    std::string expr = expr_full;
    std::string src;
    // a template instantiation is all this translation unit is asked for: the headers' plain kernels (initial ensemble, parameter
    // tables, accumulate / update, the selftest hooks: ~9 000 instructions of ISA) stay out of it.  (The two plain kernels that ARE
    // built at run time, for Float32 state, are asked for by name: those builds keep them.)
    if (inst.find('<') != std::string::npos) src += "#define AMC_PLAIN_KERNELS 0\n";
    if (f32) src += "#define AMC_STATE_F32 1\n";
    if (std::getenv("AMC_NO_SIGMA_MEMO")) src += "#define AMC_NO_SIGMA_MEMO 1\n";      // A/B: amc_log(sigma) per lane and step in K > 1 sweeps (amc_model.h SigmaArg)
    if (const char* w = std::getenv("AMC_RTC_WAVES")) src += "#define AMC_RTC_WAVES " + std::to_string(std::atoi(w)) + "\n";       // A/B: amdgpu_waves_per_eu of the script-defined estimator forms
    auto cut_tail = [&](char mark) -> std::string {      // removes and returns what follows the LAST section mark
        const size_t at = expr.find(mark);
        if (at == std::string::npos) return std::string();
        const std::string tail = expr.substr(at + 1);
        expr.erase(at);
        return tail;
    };
    const std::string e_classes = cut_tail('\x0f');      // [ '\x0f' n_classes { sections of the classes 1 .. } ]: pools that mix policies / actions
    const std::string e_np = cut_tail('\x0e');           // [ '\x0e' P ]: parameters of the policy, when more than one; the dlogq section then holds P
                                                         // expressions, '\x0b' between them
    const std::string e_invert = cut_tail('\x08'), e_perform = cut_tail('\x07');
    const std::string e_dlogq = cut_tail('\x06'), e_logq = cut_tail('\x05'), e_sample = cut_tail('\x04'), e_scale = cut_tail('\x03');
    if (!e_perform.empty()) src += "#define AMC_USER_PERFORM(x, delta) (" + e_perform + ")\n";
    if (!e_invert.empty()) src += "#define AMC_USER_INVERT(delta, x) (" + e_invert + ")\n";
    if (!e_sample.empty()) src += "#define AMC_USER_SAMPLE(z, x, sigma) (" + e_sample + ")\n";
    if (!e_logq.empty()) src += "#define AMC_USER_LOGQ(delta, x, sigma) (" + e_logq + ")\n";
    if (!e_np.empty()) src += "#define AMC_NP " + e_np + "\n";
    if (!e_dlogq.empty()) {
        size_t from = 0;
        for (int pidx = 0; from <= e_dlogq.size(); ++pidx) {
            const size_t to = e_dlogq.find('\x0b', from);
            const std::string one = e_dlogq.substr(from, to == std::string::npos ? std::string::npos : to - from);
            src += "#define AMC_USER_DLOGQ" + (pidx == 0 ? std::string() : std::to_string(pidx)) + "(delta, x, sigma) (" + one + ")\n";
            if (to == std::string::npos) break;
            from = to + 1;
        }
    }
    if (!e_scale.empty()) src += "#define AMC_USER_SCALE(x) (" + e_scale + ")\n";
    // a class whose expressions are the built-in Gaussian displacement's, as the host mirror writes them out (montecarlo_amd/metropolis.py
    // GAUSS_SAMPLE / GAUSS_LOGQ, the displacement's own perform / invert): the sweep takes its density from the move's table row
    // (amc_model.h GaussRow)
    auto is_gauss = [](const std::string& sample, const std::string& logq, const std::string& perform, const std::string& invert) {
        return sample == "sigma*z" && logq == "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0" &&
               perform.empty() && invert.empty();
    };
    // ... derivative included (GAUSS_DLOGQ): the estimator then knows the backward density and derivative without forming them
    auto is_gauss_d = [](const std::string& dlogq) { return dlogq == "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma"; };
    unsigned gauss_mask = is_gauss(e_sample, e_logq, e_perform, e_invert) ? 1u : 0u;
    unsigned gauss_est_mask = (gauss_mask && is_gauss_d(e_dlogq)) ? 1u : 0u;
    if (!e_classes.empty()) {
        const size_t first = e_classes.find('\x10');
        src += "#define AMC_NCLASS " + e_classes.substr(0, first) + "\n";
        size_t at = first;
        for (int c = 1; at != std::string::npos; ++c) {
            const size_t nxt = e_classes.find('\x10', at + 1);
            const std::string blob = e_classes.substr(at + 1, nxt == std::string::npos ? std::string::npos : nxt - at - 1);
            const size_t m1 = blob.find('\x11'), m2 = blob.find('\x12'), m3 = blob.find('\x13'), m4 = blob.find('\x14');
            const std::string sfx = "_" + std::to_string(c);
            const std::string c_sample = blob.substr(0, m1), c_logq = blob.substr(m1 + 1, m2 - m1 - 1), c_dlogq = blob.substr(m2 + 1, m3 - m2 - 1),
                              c_perform = blob.substr(m3 + 1, m4 - m3 - 1), c_invert = blob.substr(m4 + 1);
            src += "#define AMC_USER_SAMPLE" + sfx + "(z, x, sigma) (" + c_sample + ")\n";
            src += "#define AMC_USER_LOGQ" + sfx + "(delta, x, sigma) (" + c_logq + ")\n";
            if (!c_dlogq.empty()) src += "#define AMC_USER_DLOGQ" + sfx + "(delta, x, sigma) (" + c_dlogq + ")\n";
            src += "#define AMC_USER_PERFORM" + sfx + "(x, delta) (" + (c_perform.empty() ? std::string("(x) + (delta)") : c_perform) + ")\n";
            src += "#define AMC_USER_INVERT" + sfx + "(delta, x) (" + (c_invert.empty() ? std::string("-(delta)") : c_invert) + ")\n";
            if (is_gauss(c_sample, c_logq, c_perform, c_invert)) {
                gauss_mask |= 1u << c;
                if (is_gauss_d(c_dlogq)) gauss_est_mask |= 1u << c;
            }
            at = nxt;
        }
        if (gauss_mask != 0u && std::getenv("AMC_NO_GAUSS_CLASS_ROWS") == nullptr) {
            src += "#define AMC_CLASS_GAUSS_MASK " + std::to_string(gauss_mask) + "\n";
            if (gauss_est_mask != 0u) src += "#define AMC_CLASS_GAUSS_EST_MASK " + std::to_string(gauss_est_mask) + "\n";
        }
    }
    const size_t cut = expr.find('\x01');
    if (!expr.empty()) src += "#define AMC_USER_POTENTIAL(x) (" + expr.substr(0, cut) + ")\n";
    if (cut != std::string::npos) src += "#define AMC_USER_REWARD(delta, x) (" + expr.substr(cut + 1) + ")\n";
    src += "#include \"amc_kernels.h\"\n";
    // the kernel sources travel in the library (amc_rtc_sources.gen.h): hiprtc finds every `#include "amc_*.h"` among them by name.
    // The flags of the offline build (Makefile): only the explicit fma()s may fuse.
    // (-disable-machine-licm is one of LLVM's generic code-generation options; a back end without it does not return an error but ends
    // the compiling process in its option parser -- the child's death below, not the host's)
    const std::string arch_opt = "--offload-arch=" + arch;
    std::vector<std::string> opts = {arch_opt, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
    if (licm_off) { opts.push_back("-mllvm"); opts.push_back("-disable-machine-licm"); }
    Built built;
    {
        const char* inproc = std::getenv("AMC_RTC_IN_PROCESS");
        const int rcb = (inproc && inproc[0] == '1') ? build_in_process(src, inst, opts, &built) : build_in_child(src, inst, opts, &built);
        if (rcb != AMC_OK) {                           // the compiler could not be run, died or timed out: AMC_ERR_COMPILE, message set
            if (built.died) {                          // ... died: it will again -- remembered for the process and, with a cache directory, beyond
                const std::string msg = amc_last_error();
                g_rtc_broken[key] = msg;
                if (!cache_file.empty())
                    if (FILE* f = std::fopen((cache_file + ".broken").c_str(), "wb")) { std::fwrite(msg.data(), 1, msg.size(), f); std::fclose(f); }
            }
            return rcb;
        }
    }
    if (log_out) *log_out = built.log;
    if (built.stage == 1) return fail(AMC_ERR_HIP, "hiprtcCreateProgram failed (%d)", built.status);
    if (built.stage == 2) return fail(AMC_ERR_HIP, "hiprtcAddNameExpression(%s) failed (%d)", inst.c_str(), built.status);
    if (built.stage == 3) {
        // the first diagnostic is what the user needs (not the "In file included from" lines in front of it); keep the message bounded
        const std::string& log = built.log;
        size_t from = log.find("error:");
        from = from == std::string::npos ? 0 : log.rfind('\n', from) + 1;       // (npos + 1 == 0: the log's first line)
        return fail(AMC_ERR_BAD_ARG, "%s: %.400s", expr_full.empty() ? "run-time kernel build failed" : "custom potential does not compile",
                    log.empty() ? "(no log)" : log.c_str() + from);
    }
    if (built.stage != 0 || built.code.empty() || built.lowered.empty()) return fail(AMC_ERR_HIP, "hiprtc produced no code for %s", inst.c_str());
    RtcCode rc;
    rc.code = std::move(built.code);
    rc.lowered = std::move(built.lowered);
    rtc_cache_store(cache_file, rc);
    *out = &g_rtc_code.emplace(key, std::move(rc)).first->second;
    return AMC_OK;
}

// The function of instantiation `inst` for this handle's expression, loaded on this handle's device.
int rtc_function(amc_handle* h, const std::string& inst, hipFunction_t* fn)
{
    auto it = h->rtc_fn.find(inst);
    if (it != h->rtc_fn.end()) { *fn = it->second; return AMC_OK; }
    const RtcCode* code = nullptr;
    { const int rc = rtc_compile(h->pot_expr, inst, h->arch, &code, nullptr); if (rc != AMC_OK) return rc; }
    hipModule_t mod = nullptr;
    AMC_HIP(hipModuleLoadData(&mod, code->code.data()));
    h->rtc_mods.push_back(mod);
    hipFunction_t f = nullptr;
    AMC_HIP(hipModuleGetFunction(&f, mod, code->lowered.c_str()));
    h->rtc_fn[inst] = f;
    *fn = f;
    return AMC_OK;
}

int rtc_launch(amc_handle* h, const std::string& inst, int grid, void** params)
{
    hipFunction_t fn = nullptr;
    { const int rc = rtc_function(h, inst, &fn); if (rc != AMC_OK) return rc; }
    AMC_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, AMC_BLOCK, 1, 1, 0, h->stream, params, nullptr));
    return AMC_OK;
}


extern "C" {

int amc_potential_check(const char* potential_expr, char* log, int log_capacity)
{
    if (log && log_capacity > 0) log[0] = 0;
    { const int rc = validate_potential_expr(potential_expr); if (rc != AMC_OK) return rc; }
    const RtcCode* code = nullptr;
    std::string text;
    const int rc = rtc_compile(potential_expr, "amc::energy_kernel<2>", AMC_BUILD_ARCH, &code, &text);
    if (log && log_capacity > 0) {
        std::strncpy(log, text.c_str(), (size_t)log_capacity - 1);
        log[log_capacity - 1] = 0;
    }
    return rc;
}

}  // extern "C"

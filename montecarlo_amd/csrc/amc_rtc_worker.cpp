// amc_rtc_worker -- the run-time compiler of script-defined models in a process of its own.
//
// libamc.so (amc_rtc.hip) starts this program as a CHILD for every kernel instantiation it has to build: request on stdin,
// answer on stdout, the compiler's own chatter on stderr.  The program never touches a GPU: it dlopens libhiprtc (which
// drives comgr / LLVM on the host for the ISA named in the options) and nothing of the HIP runtime.  What it buys: a fatal
// error inside the compiler -- LLVM's report_fatal_error on a back-end bug, an assertion, a stack overflow on hostile
// expression text, an option the back end does not know -- ends THIS process; the engine's host (a Julia session, a Python
// driver) gets AMC_ERR_COMPILE with the child's stderr in amc_last_error() and goes on.  The reference's convention is to
// raise, never to die: error("No ... is defined"), src/metropolis.jl:35; files closed in `finally`, src/simulation.jl:176,194-199.
//
// Wire format (host byte order; both ends are the same machine).  Every blob is `u64 length` + bytes.
//   request :  u64 magic 'AMCRTCQ1', u32 n_options, u32 n_headers,
//              blobs: program source, program name, name expression (the instantiation), n_options options,
//                     n_headers header names, n_headers header texts
//   answer  :  u64 magic 'AMCRTCA1', i32 stage (0 = code object follows; 1 create, 2 name expression, 3 compile, 4 no code:
//              the hiprtc call that failed), i32 hiprtc status, blobs: compiler log, lowered name, code object
// Exit status 0 whenever an answer was written (a script that does not compile is an answer); 2 = bad request, 3 = libhiprtc
// cannot be loaded.  Anything else -- a signal above all -- is the compiler dying, which the parent reports.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

const uint64_t MAGIC_REQUEST = 0x3151435452434d41ull;      // "AMCRTCQ1"
const uint64_t MAGIC_ANSWER = 0x3141435452434d41ull;       // "AMCRTCA1"

bool read_all(int fd, void* dst, size_t n)
{
    char* p = (char*)dst;
    while (n > 0) {
        const ssize_t got = read(fd, p, n);
        if (got <= 0) return false;
        p += got;
        n -= (size_t)got;
    }
    return true;
}

bool write_all(int fd, const void* src, size_t n)
{
    const char* p = (const char*)src;
    while (n > 0) {
        const ssize_t put = write(fd, p, n);
        if (put <= 0) return false;
        p += put;
        n -= (size_t)put;
    }
    return true;
}

bool read_blob(int fd, std::string* out)
{
    uint64_t n = 0;
    if (!read_all(fd, &n, sizeof(n)) || n > (64ull << 20)) return false;
    out->resize((size_t)n);
    return n == 0 || read_all(fd, &(*out)[0], (size_t)n);
}

bool write_blob(int fd, const void* p, size_t n)
{
    const uint64_t len = n;
    return write_all(fd, &len, sizeof(len)) && (n == 0 || write_all(fd, p, n));
}

int answer(int stage, int status, const std::string& log, const std::string& lowered, const std::vector<char>& code)
{
    const int32_t head[2] = {stage, status};
    const bool ok = write_all(1, &MAGIC_ANSWER, sizeof(MAGIC_ANSWER)) && write_all(1, head, sizeof(head)) &&
                    write_blob(1, log.data(), log.size()) && write_blob(1, lowered.data(), lowered.size()) &&
                    write_blob(1, code.data(), code.size());
    return ok ? 0 : 4;
}

}  // namespace

int main()
{
    // developer / test hook: a compiler that dies the way LLVM does (report_fatal_error ends in abort()), or one that never
    // comes back -- what the parent has to survive.  Read before anything else so that no request is needed to provoke it.
    if (const char* f = std::getenv("AMC_RTC_WORKER_FAULT")) {
        if (std::strcmp(f, "abort") == 0) { std::fprintf(stderr, "LLVM ERROR: injected fatal error (AMC_RTC_WORKER_FAULT=abort)\n"); std::abort(); }
        if (std::strcmp(f, "hang") == 0) for (;;) pause();
        if (std::strcmp(f, "garbage") == 0) { std::fputs("this is not an answer", stdout); return 0; }
    }
    uint64_t magic = 0;
    uint32_t counts[2] = {0, 0};
    if (!read_all(0, &magic, sizeof(magic)) || magic != MAGIC_REQUEST || !read_all(0, counts, sizeof(counts)) || counts[0] > 64 || counts[1] > 256) {
        std::fprintf(stderr, "amc_rtc_worker: not a request (this program is started by libamc.so)\n");
        return 2;
    }
    std::string src, name, inst;
    if (!read_blob(0, &src) || !read_blob(0, &name) || !read_blob(0, &inst)) return 2;
    std::vector<std::string> opts(counts[0]), hnames(counts[1]), htexts(counts[1]);
    for (auto& o : opts) if (!read_blob(0, &o)) return 2;
    for (auto& h : hnames) if (!read_blob(0, &h)) return 2;
    for (auto& h : htexts) if (!read_blob(0, &h)) return 2;

    void* lib = nullptr;
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) { std::fprintf(stderr, "amc_rtc_worker: cannot dlopen libhiprtc: %s\n", dlerror()); return 3; }
    auto CreateProgram = (int (*)(void**, const char*, const char*, int, const char**, const char**))dlsym(lib, "hiprtcCreateProgram");
    auto AddNameExpression = (int (*)(void*, const char*))dlsym(lib, "hiprtcAddNameExpression");
    auto CompileProgram = (int (*)(void*, int, const char**))dlsym(lib, "hiprtcCompileProgram");
    auto GetProgramLogSize = (int (*)(void*, size_t*))dlsym(lib, "hiprtcGetProgramLogSize");
    auto GetProgramLog = (int (*)(void*, char*))dlsym(lib, "hiprtcGetProgramLog");
    auto GetCodeSize = (int (*)(void*, size_t*))dlsym(lib, "hiprtcGetCodeSize");
    auto GetCode = (int (*)(void*, char*))dlsym(lib, "hiprtcGetCode");
    auto GetLoweredName = (int (*)(void*, const char*, const char**))dlsym(lib, "hiprtcGetLoweredName");
    if (!CreateProgram || !AddNameExpression || !CompileProgram || !GetProgramLogSize || !GetProgramLog || !GetCodeSize || !GetCode || !GetLoweredName) {
        std::fprintf(stderr, "amc_rtc_worker: libhiprtc lacks a hiprtc entry point\n");
        return 3;
    }

    std::vector<const char*> hn, ht, op;
    for (auto& h : hnames) hn.push_back(h.c_str());
    for (auto& h : htexts) ht.push_back(h.c_str());
    for (auto& o : opts) op.push_back(o.c_str());
    const std::vector<char> none;
    void* prog = nullptr;
    int e = CreateProgram(&prog, src.c_str(), name.c_str(), (int)hn.size(), ht.data(), hn.data());
    if (e != 0) return answer(1, e, "", "", none);
    e = AddNameExpression(prog, inst.c_str());
    if (e != 0) return answer(2, e, "", "", none);
    e = CompileProgram(prog, (int)op.size(), op.data());
    std::string log;
    size_t ls = 0;
    if (GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
        log.resize(ls);
        GetProgramLog(prog, &log[0]);
        while (!log.empty() && log.back() == 0) log.pop_back();
    }
    if (e != 0) return answer(3, e, log, "", none);
    size_t cs = 0;
    const char* lowered = nullptr;
    if (GetCodeSize(prog, &cs) != 0 || cs == 0 || GetLoweredName(prog, inst.c_str(), &lowered) != 0 || !lowered) return answer(4, 0, log, "", none);
    std::vector<char> code(cs);
    GetCode(prog, code.data());
    // (no hiprtcDestroyProgram: the process ends here, and its teardown is one more thing that could go wrong after the answer is out)
    const int rc = answer(0, 0, log, lowered, code);
    std::fflush(stdout);
    _exit(rc);
}

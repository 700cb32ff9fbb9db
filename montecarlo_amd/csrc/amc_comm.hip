// amc_comm.hip -- the engine's own RCCL communicator over xGMI (amc_comm_*, amc_allreduce_sum, amc_allreduce_xsum): one process per
// GPU, the two cross-shard sums of the path as one ncclAllReduce each, used as a gather of exact records.
// RCCL is resolved with dlopen: no link-time dependency, and the instance a host process has loaded already is shared.
#define AMC_KERNEL_LINKAGE static      // the kernel headers are included for their types only: no kernel of theirs in this object
#include "amc_internal.h"

// Back to a single shard: the communicator and what was allocated for it.
void comm_release(amc_handle* h)
{
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->comm && h->rccl.CommDestroy) h->rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_ranks = 1;
    h->comm_capacity = 0;
    h->pg_tail_valid = false;
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    h->comm_stream = nullptr;
    if (h->ev_comm_main) (void)hipEventDestroy(h->ev_comm_main);
    h->ev_comm_main = nullptr;
    h->comm_main_pending = false;
    (void)hipFree(h->d_comm);
    h->d_comm = nullptr;
    (void)hipHostFree(h->h_comm);
    h->h_comm = nullptr;
}

extern "C" {

// ---- RCCL over xGMI, for hosts that have no torch.distributed (the Julia binding) ----
static int load_rccl(Rccl& r)
{
    if (r.lib) return AMC_OK;
    // AMC_RCCL_LIBRARY=<file>: that library and no other (a site's own RCCL build; the tests' shared-memory stand-in that
    // lets several ranks share the one GPU of a test box, tests/aux/fake_rccl.c)
    if (const char* forced = std::getenv("AMC_RCCL_LIBRARY")) {
        r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return fail(AMC_ERR_COMM, "cannot dlopen AMC_RCCL_LIBRARY=%s: %s", forced, dlerror());
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
    }
    if (!r.lib) return fail(AMC_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
    r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    r.CommCount = (int (*)(void*, int*))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (int (*)(void*, int*))dlsym(r.lib, "ncclCommUserRank");
    r.GetVersion = (int (*)(int*))dlsym(r.lib, "ncclGetVersion");
    void* init = dlsym(r.lib, "ncclCommInitRank");
    r.CommInitRank = (int (*)(void**, int, const void*, int))init;
    if (!r.GetUniqueId || !r.AllReduce || !r.CommDestroy || !init) {
        r.lib = nullptr;
        return fail(AMC_ERR_COMM, "librccl is missing a required symbol");
    }
    return AMC_OK;
}

int amc_comm_unique_id(void* id128)
{
    if (!id128) return fail(AMC_ERR_BAD_ARG, "amc_comm_unique_id: NULL argument");
    static Rccl r;
    const int rc = load_rccl(r);
    if (rc != AMC_OK) return rc;
    const int e = r.GetUniqueId(id128);
    if (e != 0) return fail(AMC_ERR_COMM, "ncclGetUniqueId failed: %s", r.GetErrorString ? r.GetErrorString(e) : "?");
    return AMC_OK;
}


int amc_comm_init(amc_handle* h, int rank, int n_ranks, const void* id128)
{
    if (!h || !id128) return fail(AMC_ERR_BAD_ARG, "amc_comm_init: NULL argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(AMC_ERR_BAD_ARG, "amc_comm_init: bad rank/n_ranks");
    AMC_HIP(hipSetDevice(h->device));
    { const int rcr = pg_resolve(h); if (rcr != AMC_OK) return rcr; }
    const int rc = load_rccl(h->rccl);
    if (rc != AMC_OK) return rc;
    if (h->comm) return fail(AMC_ERR_STATE, "amc_comm_init: this handle already has a communicator");
    // everything the communicator's users need exists BEFORE the communicator does: a failure below leaves the handle
    // a clean single shard (amc_allreduce_sum the identity again, a later amc_comm_init welcome)
    // room for a gather of every shard's callback records (amc_allreduce_xsum) and, in d_out, of its estimator records
    const int capacity = n_ranks * (AMC_RED_HEADER + AMC_MAX_MOVES) * amc::xs::XS_WORDS > 256
                             ? n_ranks * (AMC_RED_HEADER + AMC_MAX_MOVES) * amc::xs::XS_WORDS : 256;
    hipError_t he = hipMalloc(&h->d_comm, (size_t)capacity * sizeof(double));
    if (he == hipSuccess) he = hipHostMalloc(&h->h_comm, (size_t)capacity * sizeof(double), hipHostMallocDefault);
    if (he == hipSuccess && n_ranks > h->d_out_ranks) {
        double* bigger = nullptr;
        he = hipStreamSynchronize(h->stream);
        if (he == hipSuccess) he = hipMalloc(&bigger, (size_t)n_ranks * PG_MAX_COLS * amc::xs::XS_WORDS * sizeof(double));
        if (he == hipSuccess) {
            (void)hipFree(h->d_out);
            h->d_out = bigger;
            h->d_out_ranks = n_ranks;
            h->pg_tail_valid = false;           // the estimator's record holds the old pointer
        }
    }
    // (a higher stream priority changes nothing for these few bytes between device-filling sweeps: measured, round 3)
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&h->ev_comm_main, hipEventDisableTiming);
    if (he != hipSuccess) {
        comm_release(h);
        return fail(he == hipErrorOutOfMemory ? AMC_ERR_OOM : AMC_ERR_HIP, "amc_comm_init: %s", hipGetErrorString(he));
    }
    // ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId /* 128-byte struct BY VALUE */, int rank)
    struct Id { char b[128]; } id;
    std::memcpy(&id, id128, sizeof(id));
    typedef int (*init_fn)(void**, int, Id, int);
    const int e = ((init_fn)(void*)h->rccl.CommInitRank)(&h->comm, n_ranks, id, rank);
    if (e != 0) {
        h->comm = nullptr;
        comm_release(h);
        return fail(AMC_ERR_COMM, "ncclCommInitRank failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(e) : "?");
    }
    h->comm_rank = rank;
    h->comm_ranks = n_ranks;
    h->comm_capacity = capacity;
    h->pg_tail_valid = false;                   // rank / n_ranks are part of the estimator's record
    return AMC_OK;
}

int amc_comm_destroy(amc_handle* h)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_comm_destroy: NULL handle");
    AMC_HIP(hipSetDevice(h->device));
    { const int rcr = pg_resolve(h); if (rcr != AMC_OK) return rcr; }
    if (h->stream) AMC_HIP(hipStreamSynchronize(h->stream));      // the estimator's collectives run there
    comm_release(h);
    return AMC_OK;
}

static void copy_path_of(const void* symbol, char* out, int capacity)
{
    if (!out || capacity < 1) return;
    out[0] = 0;
    Dl_info info;
    if (symbol && dladdr(symbol, &info) && info.dli_fname) {
        std::strncpy(out, info.dli_fname, (size_t)capacity - 1);
        out[capacity - 1] = 0;
    }
}

int amc_comm_info(amc_handle* h, int* n_ranks, int* rank, int* rccl_version, char* librccl_path, int path_capacity)
{
    if (!h) return fail(AMC_ERR_BAD_ARG, "amc_comm_info: NULL handle");
    if (n_ranks) *n_ranks = 1;
    if (rank) *rank = 0;
    if (rccl_version) *rccl_version = 0;
    if (librccl_path && path_capacity > 0) librccl_path[0] = 0;
    if (!h->comm) return AMC_OK;
    // asked of RCCL itself, not remembered from amc_comm_init's arguments: the point is what the communicator spans
    if (!h->rccl.CommCount || !h->rccl.CommUserRank)
        return fail(AMC_ERR_COMM, "amc_comm_info: this librccl exports no ncclCommCount / ncclCommUserRank");
    int v = 0;
    int e = h->rccl.CommCount(h->comm, &v);
    if (e != 0) return fail(AMC_ERR_COMM, "ncclCommCount failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(e) : "?");
    if (n_ranks) *n_ranks = v;
    e = h->rccl.CommUserRank(h->comm, &v);
    if (e != 0) return fail(AMC_ERR_COMM, "ncclCommUserRank failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(e) : "?");
    if (rank) *rank = v;
    if (rccl_version && h->rccl.GetVersion && h->rccl.GetVersion(&v) == 0) *rccl_version = v;
    copy_path_of((const void*)h->rccl.AllReduce, librccl_path, path_capacity);
    return AMC_OK;
}

int amc_runtime_info(int* hip_runtime_version, char* hip_runtime_path, int path_capacity)
{
    if (hip_runtime_version) {
        int v = 0;
        AMC_HIP(hipRuntimeGetVersion(&v));
        *hip_runtime_version = v;
    }
    copy_path_of((const void*)&hipRuntimeGetVersion, hip_runtime_path, path_capacity);
    return AMC_OK;
}

int amc_allreduce_sum(amc_handle* h, double* buf, int n)
{
    if (!h || !buf) return fail(AMC_ERR_BAD_ARG, "amc_allreduce_sum: NULL argument");
    if (n < 0) return fail(AMC_ERR_BAD_ARG, "amc_allreduce_sum: n < 0");
    if (!h->comm) return AMC_OK;   // single shard: the local sum is the global sum
    if (n > h->comm_capacity) return fail(AMC_ERR_BAD_ARG, "amc_allreduce_sum: n must be in [0, %d]", h->comm_capacity);
    AMC_HIP(hipSetDevice(h->device));
    // The values are the caller's (host) numbers: nothing here depends on the sweeps queued on the engine's stream, so the
    // collective runs on comm_stream and the host waits for THAT only (bench.py keeps ten sweeps in flight behind a
    // callback).  One communicator serves both streams, and RCCL wants its collectives issued and run in one order on
    // every rank: a collective queued on the engine's stream earlier (the estimator's in-place all-reduce) is waited for
    // here first -- which drains the engine's stream up to that point, the price of sharing the communicator; with no
    // estimator in the run nothing is pending and nothing waits.  The other direction needs no event: this call returns
    // only when its collective is complete.  Every rank calls in the same order (same host program).
    if (h->comm_main_pending) {
        AMC_HIP(hipStreamWaitEvent(h->comm_stream, h->ev_comm_main, 0));
        h->comm_main_pending = false;
    }
    std::memcpy(h->h_comm, buf, (size_t)n * sizeof(double));
    AMC_HIP(hipMemcpyAsync(h->d_comm, h->h_comm, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->comm_stream));
    const int e = h->rccl.AllReduce(h->d_comm, h->d_comm, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, h->comm_stream);
    if (e != 0) return fail(AMC_ERR_COMM, "ncclAllReduce failed: %s", h->rccl.GetErrorString ? h->rccl.GetErrorString(e) : "?");
    AMC_HIP(hipMemcpyAsync(h->h_comm, h->d_comm, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->comm_stream));
    AMC_HIP(hipStreamSynchronize(h->comm_stream));
    std::memcpy(buf, h->h_comm, (size_t)n * sizeof(double));
    return AMC_OK;
}

// records[i] <- the sum over all shards of records[i], for every i < n_records: a GATHER of the shards' records (each shard
// fills its own slot of a zeroed buffer, so the all-reduce(sum) adds one value and zeros per word: exact in any order),
// then the integer merge of amc_xsum.h in rank order -- which, the merge being exact, is any order.  Every shard ends
// with the same bits, and they are the bits a single shard holding all the chains would have.
int amc_allreduce_xsum(amc_handle* h, double* records, int n_records)
{
    if (!h || !records) return fail(AMC_ERR_BAD_ARG, "amc_allreduce_xsum: NULL argument");
    if (n_records < 0) return fail(AMC_ERR_BAD_ARG, "amc_allreduce_xsum: n_records < 0");
    if (!h->comm || n_records == 0) return AMC_OK;
    const size_t per = (size_t)n_records * amc::xs::XS_WORDS;
    if (per * (size_t)h->comm_ranks > (size_t)h->comm_capacity)
        return fail(AMC_ERR_BAD_ARG, "amc_allreduce_xsum: at most %d records", h->comm_capacity / (h->comm_ranks * amc::xs::XS_WORDS));
    std::vector<double> buf(per * (size_t)h->comm_ranks, 0.0);
    std::memcpy(buf.data() + per * (size_t)h->comm_rank, records, per * sizeof(double));
    const int rc = amc_allreduce_sum(h, buf.data(), (int)buf.size());
    if (rc != AMC_OK) return rc;
    std::memcpy(records, buf.data(), per * sizeof(double));
    for (int r = 1; r < h->comm_ranks; ++r)
        for (int i = 0; i < n_records; ++i)
            amc::xs::rec_merge(records + (size_t)i * amc::xs::XS_WORDS, buf.data() + per * (size_t)r + (size_t)i * amc::xs::XS_WORDS);
    return AMC_OK;
}

// 1 when AMC_RCCL_LIBRARY replaced librccl for this process (a site's own build -- or the tests' stand-in): a result obtained
// that way must say so.
int amc_comm_library_forced(int* forced)
{
    if (!forced) return fail(AMC_ERR_BAD_ARG, "amc_comm_library_forced: NULL argument");
    const char* f = std::getenv("AMC_RCCL_LIBRARY");
    *forced = (f && *f) ? 1 : 0;
    return AMC_OK;
}

}  // extern "C"

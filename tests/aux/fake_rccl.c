/* fake_rccl.c -- TEST INFRASTRUCTURE: a stand-in for librccl that lets N ranks share ONE GPU.
 *
 * RCCL refuses two ranks on one device ("Duplicate GPU detected") and the GPU boxes this repository runs on have one GPU,
 * so the engine's N > 1 code paths (amc_comm_init, the estimator's in-place all-reduce on the engine's stream,
 * amc_allreduce_sum on its communication stream, amc_comm_info, bench.py --gpus N) never meet more than one rank there.
 * This library exports the eight RCCL entry points libamc.so resolves with dlsym and carries the collective over POSIX
 * shared memory between processes of one host: every rank copies its buffer device -> host, the ranks meet at a barrier,
 * each adds the contributions in rank order and copies the sum host -> device.  Same results as a real all-reduce(sum);
 * NOT the same execution model (it blocks the calling host thread until the stream is idle and all ranks have arrived), and
 * it says nothing about RCCL or xGMI themselves.  Selected with AMC_RCCL_LIBRARY=<this .so>; never loaded otherwise.
 *
 * Build: gcc -O2 -shared -fPIC tests/aux/fake_rccl.c -o <out>.so -ldl -lrt -pthread
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define FAKE_MAX_RANKS 8
#define FAKE_MAX_COUNT 4096      /* 8 ranks x 32 sums x 12-word records */
#define FAKE_TIMEOUT_S 60.0

/* AMC_FAKE_RCCL_TIMEOUT_S: how long a rank waits for the others; AMC_FAKE_RCCL_FAIL_RANK: that rank's ncclCommInitRank fails
 * (the others then time out waiting for it, the way a real communicator set-up hangs or fails when one rank is missing) */
static double timeout_s(void)
{
    const char* e = getenv("AMC_FAKE_RCCL_TIMEOUT_S");
    return (e && atof(e) > 0.0) ? atof(e) : FAKE_TIMEOUT_S;
}

typedef struct {
    _Atomic int arrived;      /* barrier: ranks that have arrived in the current generation */
    _Atomic int generation;
    _Atomic int attached;     /* ranks that have called ncclCommInitRank */
    double data[FAKE_MAX_RANKS][FAKE_MAX_COUNT];
} shared_t;

typedef struct {
    shared_t* sh;
    int rank, n_ranks;
    char name[128];
} comm_t;

typedef struct { char b[128]; } unique_id_t;

typedef int (*memcpy_fn)(void*, const void*, size_t, int);
typedef int (*sync_fn)(void*);
static memcpy_fn p_memcpy;
static sync_fn p_stream_sync;

static int resolve_hip(void)
{
    if (p_memcpy && p_stream_sync) return 0;
    /* the runtime the host process already uses: libamc.so's own dependency, loaded with local scope (ctypes), so it is
     * found by soname among the loaded objects (RTLD_NOLOAD: never a second copy), not in the global scope */
    const char* names[] = {"libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"};
    void* lib = NULL;
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    void* scope = lib ? lib : RTLD_DEFAULT;
    p_memcpy = (memcpy_fn)dlsym(scope, "hipMemcpy");
    p_stream_sync = (sync_fn)dlsym(scope, "hipStreamSynchronize");
    return (p_memcpy && p_stream_sync) ? 0 : 1;
}

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* sense-reversing barrier over the shared segment; 0 = ok, 1 = timed out (a rank died or the call order differs) */
static int barrier(comm_t* c)
{
    const int gen = atomic_load(&c->sh->generation);
    if (atomic_fetch_add(&c->sh->arrived, 1) + 1 == c->n_ranks) {
        atomic_store(&c->sh->arrived, 0);
        atomic_fetch_add(&c->sh->generation, 1);
        return 0;
    }
    const double t0 = now_s();
    while (atomic_load(&c->sh->generation) == gen) {
        sched_yield();
        if (now_s() - t0 > timeout_s()) return 1;
    }
    return 0;
}

int ncclGetUniqueId(void* id128)
{
    unique_id_t* id = (unique_id_t*)id128;
    memset(id, 0, sizeof(*id));
    snprintf(id->b, sizeof(id->b), "/amc_fake_rccl_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
    /* rank 0 creates the segment here, so that it exists before any other rank has the id */
    int fd = shm_open(id->b, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)sizeof(shared_t)) != 0) { close(fd); return 2; }
    close(fd);
    return 0;
}

int ncclCommInitRank(void** comm, int n_ranks, unique_id_t id, int rank)
{
    if (!comm || n_ranks < 1 || n_ranks > FAKE_MAX_RANKS || rank < 0 || rank >= n_ranks) return 4;
    if (resolve_hip() != 0) return 3;
    { const char* bad = getenv("AMC_FAKE_RCCL_FAIL_RANK"); if (bad && atoi(bad) == rank) return 5; }
    int fd = shm_open(id.b, O_RDWR, 0600);
    if (fd < 0) return 2;
    shared_t* sh = (shared_t*)mmap(NULL, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (sh == MAP_FAILED) return 2;
    comm_t* c = (comm_t*)calloc(1, sizeof(comm_t));
    c->sh = sh;
    c->rank = rank;
    c->n_ranks = n_ranks;
    snprintf(c->name, sizeof(c->name), "%s", id.b);
    atomic_fetch_add(&sh->attached, 1);
    /* like the real call, return only when every rank has joined */
    const double t0 = now_s();
    while (atomic_load(&sh->attached) < n_ranks) {
        sched_yield();
        if (now_s() - t0 > timeout_s()) { munmap(sh, sizeof(shared_t)); free(c); return 6; }
    }
    *comm = c;
    return 0;
}

/* dtype 8 = ncclFloat64, op 0 = ncclSum: all this path uses */
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* stream)
{
    comm_t* c = (comm_t*)comm;
    if (!c || dtype != 8 || op != 0 || count > FAKE_MAX_COUNT) return 4;
    if (p_stream_sync(stream) != 0) return 1;                                     /* what was queued before the collective */
    if (p_memcpy(c->sh->data[c->rank], send, count * sizeof(double), 2 /* hipMemcpyDeviceToHost */) != 0) return 1;
    if (barrier(c)) return 6;
    double sum[FAKE_MAX_COUNT];
    for (size_t i = 0; i < count; ++i) {
        double s = c->sh->data[0][i];
        for (int r = 1; r < c->n_ranks; ++r) s += c->sh->data[r][i];              /* rank order: the same bits on every rank */
        sum[i] = s;
    }
    if (barrier(c)) return 6;                                                     /* nobody overwrites a slot still being read */
    if (p_memcpy(recv, sum, count * sizeof(double), 1 /* hipMemcpyHostToDevice */) != 0) return 1;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    comm_t* c = (comm_t*)comm;
    if (!c) return 0;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->sh, sizeof(shared_t));
    free(c);
    return 0;
}

int ncclCommCount(void* comm, int* count) { if (!comm || !count) return 4; *count = ((comm_t*)comm)->n_ranks; return 0; }
int ncclCommUserRank(void* comm, int* rank) { if (!comm || !rank) return 4; *rank = ((comm_t*)comm)->rank; return 0; }
int ncclGetVersion(int* v) { if (!v) return 4; *v = 1; return 0; }               /* version 1: nobody mistakes it for RCCL */

const char* ncclGetErrorString(int e)
{
    switch (e) {
    case 0: return "fake rccl: ok";
    case 1: return "fake rccl: HIP call failed";
    case 2: return "fake rccl: shared memory segment";
    case 3: return "fake rccl: HIP runtime not found in the process";
    case 4: return "fake rccl: invalid argument";
    case 5: return "fake rccl: this rank was told to fail (AMC_FAKE_RCCL_FAIL_RANK)";
    case 6: return "fake rccl: timed out waiting for the other ranks";
    default: return "fake rccl: unknown error";
    }
}

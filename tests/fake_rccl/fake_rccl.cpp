// fake_rccl.cpp -- TEST INFRASTRUCTURE: a host-staged stand-in for the ten RCCL entry points libneutfem_hip.so uses,
// so that the real multi-process code path (nf_comm_init, interface planes between ranks, scalar all-reduces, bench.py
// --gpus N) can run with several processes on the ONE GPU of a test box -- RCCL itself refuses two ranks on one device.
// Selected with NEUTFEM_RCCL_LIB=<this .so>; never loaded otherwise.
//
// Semantics kept: calls are asynchronous and ordered on the stream they are given (device -> pinned staging copy, a host
// function that talks to the other ranks through a POSIX shared-memory segment, staging -> device copy).  Sends are
// buffered in a one-message mailbox per ordered rank pair, so the send/recv order of neutfem_hip's grouped exchange
// cannot deadlock; all-reduces sum in rank order on every rank (bitwise identical results everywhere, like RCCL).
// Operations on ONE communicator are serialized in the order of the calls, whatever streams they are given (an event chain per
// communicator: a call on another stream than its predecessor's first waits for the predecessor) -- that is what RCCL does (every launch of
// a communicator goes through the communicator's own device stream), and it turns a schedule whose stream dependencies contradict the
// order of the calls into a hang here as it would be there.  FAKE_RCCL_SERIALIZE=0 switches the chain off; FAKE_RCCL_REPORT=1 prints the
// number of calls and of cross-stream waits per rank when the communicator is destroyed.
// Only what neutfem_hip needs: fp64, sum / max, counts <= 20480 for all-reduce (scalars and the vectors of block partials), nranks <= 8.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {
const int MAXR = 8;
const int AR_MAX = 20480;                        // doubles per all-reduce
struct Mailbox { std::atomic<long> written, consumed; };
struct Shared {
    std::atomic<int> joined;
    std::atomic<long> ar_arrive[MAXR];
    double ar_val[2][MAXR][AR_MAX];
    Mailbox box[MAXR][MAXR];                      // [src][dst]
    // message payloads follow: box_data(src, dst) = data + (src * MAXR + dst) * cap
};
struct Comm {
    int nranks, rank; long cap; size_t bytes;
    Shared *sh; double *data; char name[80];
    long ar_seq = 0, send_seq[MAXR] = {0}, recv_seq[MAXR] = {0};
    double *stage_ar = nullptr, *stage_send[MAXR] = {nullptr}, *stage_recv[MAXR] = {nullptr};
    bool serialize = true, have_last = false; hipEvent_t ev = nullptr; hipStream_t last = nullptr; long calls = 0, cross = 0;
};
// the communicator's chain: the call waits for its predecessor if that one went to another stream, and becomes the predecessor
bool chain_begin(Comm *c, hipStream_t st)
{
    ++c->calls;
    if (!c->serialize || !c->have_last || c->last == st) return true;
    ++c->cross;
    return hipStreamWaitEvent(st, c->ev, 0) == hipSuccess;
}
bool chain_end(Comm *c, hipStream_t st)
{
    if (!c->serialize) return true;
    c->have_last = true; c->last = st;
    return hipEventRecord(c->ev, st) == hipSuccess;
}
double *box_data(Comm *c, int src, int dst) { return c->data + ((size_t)src * MAXR + dst) * c->cap; }
bool wait_until(const std::atomic<long> &a, long v)
{
    for (int spin = 0; spin < 20000; ++spin) {                  // peers are usually microseconds apart
        if (a.load(std::memory_order_acquire) >= v) return true;
        __builtin_ia32_pause();
    }
    const time_t t0 = time(nullptr);
    while (a.load(std::memory_order_acquire) < v) {
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "fake_rccl: peer did not arrive within 120 s\n"); return false; }
        usleep(5);
    }
    return true;
}
struct Op { Comm *c; int peer; size_t count; long seq; int op; };
void cb_send(void *p)
{
    Op *o = (Op *)p; Comm *c = o->c; Mailbox &m = c->sh->box[c->rank][o->peer];
    if (wait_until(m.consumed, o->seq - 1)) {                    // previous message taken
        memcpy(box_data(c, c->rank, o->peer), c->stage_send[o->peer], o->count * sizeof(double));
        m.written.store(o->seq, std::memory_order_release);
    }
    delete o;
}
void cb_recv(void *p)
{
    Op *o = (Op *)p; Comm *c = o->c; Mailbox &m = c->sh->box[o->peer][c->rank];
    if (wait_until(m.written, o->seq)) {
        memcpy(c->stage_recv[o->peer], box_data(c, o->peer, c->rank), o->count * sizeof(double));
        m.consumed.store(o->seq, std::memory_order_release);
    }
    delete o;
}
void cb_allreduce(void *p)
{
    Op *o = (Op *)p; Comm *c = o->c; Shared *sh = c->sh;
    const int buf = (int)(o->seq & 1);
    for (size_t q = 0; q < o->count; ++q) sh->ar_val[buf][c->rank][q] = c->stage_ar[q];
    sh->ar_arrive[c->rank].store(o->seq, std::memory_order_release);
    bool ok = true;
    for (int r = 0; r < c->nranks && ok; ++r) ok = wait_until(sh->ar_arrive[r], o->seq);
    if (ok)
        for (size_t q = 0; q < o->count; ++q) {
            double s = sh->ar_val[buf][0][q];
            for (int r = 1; r < c->nranks; ++r) { const double v = sh->ar_val[buf][r][q]; s = o->op == 2 ? (v > s ? v : s) : s + v; }
            c->stage_ar[q] = s;
        }
    delete o;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef Comm *ncclComm_t;

const char *ncclGetErrorString(int e) { return e == 0 ? "success" : "fake_rccl error"; }
int ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}
int ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (nranks > MAXR) return 1;
    Comm *c = new Comm; c->nranks = nranks; c->rank = rank;
    const char *e = getenv("FAKE_RCCL_CAP"); c->cap = e ? atol(e) : (1L << 18);            // doubles per message
    strncpy(c->name, id.internal, sizeof c->name - 1); c->name[sizeof c->name - 1] = 0;
    c->bytes = sizeof(Shared) + (size_t)MAXR * MAXR * c->cap * sizeof(double);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { perror("fake_rccl shm_open"); return 1; }
    if (ftruncate(fd, (off_t)c->bytes) != 0) { perror("fake_rccl ftruncate"); return 1; }      // zero-filled by the kernel
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("fake_rccl mmap"); return 1; }
    c->sh = (Shared *)m; c->data = (double *)((char *)m + sizeof(Shared));
    if (hipHostMalloc((void **)&c->stage_ar, AR_MAX * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
    for (int r = 0; r < nranks; ++r) {
        if (hipHostMalloc((void **)&c->stage_send[r], c->cap * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
        if (hipHostMalloc((void **)&c->stage_recv[r], c->cap * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
    }
    const char *se = getenv("FAKE_RCCL_SERIALIZE"); c->serialize = !(se && atoi(se) == 0);
    if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) return 1;
    c->sh->joined.fetch_add(1);
    const time_t t0 = time(nullptr);
    while (c->sh->joined.load() < nranks) { if (time(nullptr) - t0 > 120) return 1; usleep(100); }
    *out = c;
    return 0;
}
int ncclCommDestroy(ncclComm_t c)
{
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    if (getenv("FAKE_RCCL_REPORT")) fprintf(stderr, "fake_rccl: rank %d of %d: %ld calls, %ld cross-stream waits\n", c->rank, c->nranks, c->calls, c->cross);
    (void)hipEventDestroy(c->ev);
    for (int r = 0; r < c->nranks; ++r) { (void)hipHostFree(c->stage_send[r]); (void)hipHostFree(c->stage_recv[r]); }
    (void)hipHostFree(c->stage_ar);
    munmap((void *)c->sh, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}
int ncclCommCount(const ncclComm_t c, int *n) { if (!c || !n) return 1; *n = c->nranks; return 0; }
int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t st)
{
    if (dtype != 8 || (op != 0 && op != 2) || count > (size_t)AR_MAX) return 1;               // ncclSum / ncclMax
    if (!chain_begin(c, st)) return 1;
    if (hipMemcpyAsync(c->stage_ar, send, count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipLaunchHostFunc(st, cb_allreduce, new Op{ c, -1, count, ++c->ar_seq, op }) != hipSuccess) return 1;
    if (hipMemcpyAsync(recv, c->stage_ar, count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return chain_end(c, st) ? 0 : 1;
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t st)
{
    if (dtype != 8 || (long)count > c->cap || peer < 0 || peer >= c->nranks) return 1;
    if (!chain_begin(c, st)) return 1;
    if (hipMemcpyAsync(c->stage_send[peer], buf, count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipLaunchHostFunc(st, cb_send, new Op{ c, peer, count, ++c->send_seq[peer], 0 }) != hipSuccess) return 1;
    return chain_end(c, st) ? 0 : 1;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t st)
{
    if (dtype != 8 || (long)count > c->cap || peer < 0 || peer >= c->nranks) return 1;
    if (!chain_begin(c, st)) return 1;
    if (hipLaunchHostFunc(st, cb_recv, new Op{ c, peer, count, ++c->recv_seq[peer], 0 }) != hipSuccess) return 1;
    if (hipMemcpyAsync(buf, c->stage_recv[peer], count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    return chain_end(c, st) ? 0 : 1;
}
}

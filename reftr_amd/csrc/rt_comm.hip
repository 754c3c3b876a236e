// rt_comm_* — the data-parallel gradient exchange behind the C ABI (SURVEY.md §8b's export set; replaces the reference's
// torch DistributedDataParallel all-reduce, main_vg.py:290-296, and the rendezvous of util/misc.py:392-431).
//
// RCCL (the collectives library of ROCm: ring / tree all-reduce over xGMI) is bound at RUN time with dlopen, not at link
// time: the library still loads on a box without RCCL (the CPU build container, single-GPU inference), and a process that
// already holds a copy of RCCL -- torch.distributed's backend 'nccl' IS RCCL and ships its own librccl.so -- gets THAT copy
// (two RCCL instances in one process fight over the same IPC handles): the loaded objects are searched for "librccl" first,
// the system library is the fall-back.
//
// One communicator per process and device (one process per GPU); the caller passes the stream every collective runs on.
#include "rt_common.h"
#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <string.h>

namespace {

// the slice of rccl.h this file needs (rccl.h: ncclUniqueId = 128 opaque bytes, ncclSum = 0, ncclFloat32 = 7, ncclBfloat16 = 9)
typedef struct { char internal[128]; } NcclId;
typedef void* NcclComm;
typedef int (*GetUniqueId_t)(NcclId*);
typedef int (*CommInitRank_t)(NcclComm*, int, NcclId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroy_t)(NcclComm);
typedef int (*Group_t)(void);
typedef const char* (*GetErrorString_t)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueId_t get_unique_id = nullptr;
    CommInitRank_t comm_init_rank = nullptr;
    AllReduce_t all_reduce = nullptr;
    CommDestroy_t comm_destroy = nullptr;
    Group_t group_start = nullptr, group_end = nullptr;
    GetErrorString_t error_string = nullptr;
    bool tried = false;
};
Rccl g_rccl;

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        strncpy((char*)data, info->dlpi_name, 1023);
        return 1;
    }
    return 0;
}

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.handle != nullptr;
    g_rccl.tried = true;
    char path[1024] = {0};
    dl_iterate_phdr(find_loaded_rccl, path);
    void* h = path[0] ? dlopen(path, RTLD_NOW | RTLD_GLOBAL) : nullptr;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.get_unique_id = (GetUniqueId_t)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRank_t)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (AllReduce_t)dlsym(h, "ncclAllReduce");
    g_rccl.comm_destroy = (CommDestroy_t)dlsym(h, "ncclCommDestroy");
    g_rccl.group_start = (Group_t)dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (Group_t)dlsym(h, "ncclGroupEnd");
    g_rccl.error_string = (GetErrorString_t)dlsym(h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) return false;
    g_rccl.handle = h;
    return true;
}

struct Comm { NcclComm comm; int rank, world; };

int nccl_rc(int rc, const char* what) {
    if (rc == 0) return RT_OK;
    fprintf(stderr, "[reftr_hip] %s failed: RCCL error %d (%s)\n", what, rc, g_rccl.error_string ? g_rccl.error_string(rc) : "?");
    return RT_ERR_COMM;
}

}  // namespace

extern "C" int rt_comm_unique_id(void* id128) {
    if (!id128) return RT_ERR_BADARG;
    if (!load_rccl()) return RT_ERR_UNSUPPORTED;
    NcclId id;
    const int rc = nccl_rc(g_rccl.get_unique_id(&id), "ncclGetUniqueId");
    if (rc != RT_OK) return rc;
    memcpy(id128, id.internal, 128);
    return RT_OK;
}

extern "C" int rt_comm_init(const void* id128, int rank, int world, rt_comm_t* out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return RT_ERR_BADARG;
    if (!load_rccl()) return RT_ERR_UNSUPPORTED;
    NcclId id;
    memcpy(id.internal, id128, 128);
    Comm* c = new Comm{nullptr, rank, world};
    const int rc = nccl_rc(g_rccl.comm_init_rank(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc != RT_OK) { delete c; return rc; }
    *out = (rt_comm_t)c;
    return RT_OK;
}

extern "C" int rt_comm_allreduce(rt_comm_t comm, void* const* bufs, const int64_t* counts, int n, int dtype, rt_stream_t stream) {
    if (!comm || !bufs || !counts || n <= 0) return RT_ERR_BADARG;
    if (dtype != RT_COMM_F32 && dtype != RT_COMM_BF16) return RT_ERR_UNSUPPORTED;
    Comm* c = (Comm*)comm;
    const int nd = dtype == RT_COMM_F32 ? 7 : 9;
    // the pieces of one boundary go out as ONE group: a single launch sequence on the stream
    if (n > 1 && g_rccl.group_start) { const int rc = nccl_rc(g_rccl.group_start(), "ncclGroupStart"); if (rc != RT_OK) return rc; }
    int rc = RT_OK;
    for (int i = 0; i < n && rc == RT_OK; ++i) {
        if (counts[i] <= 0) continue;
        if (!bufs[i]) { rc = RT_ERR_BADARG; break; }
        rc = nccl_rc(g_rccl.all_reduce(bufs[i], bufs[i], (size_t)counts[i], nd, /*ncclSum*/ 0, c->comm, (hipStream_t)stream), "ncclAllReduce");
    }
    if (n > 1 && g_rccl.group_end) { const int rc2 = nccl_rc(g_rccl.group_end(), "ncclGroupEnd"); if (rc == RT_OK) rc = rc2; }
    return rc;
}

extern "C" int rt_comm_destroy(rt_comm_t comm) {
    if (!comm) return RT_ERR_BADARG;
    Comm* c = (Comm*)comm;
    const int rc = g_rccl.comm_destroy ? nccl_rc(g_rccl.comm_destroy(c->comm), "ncclCommDestroy") : RT_OK;
    delete c;
    return rc;
}

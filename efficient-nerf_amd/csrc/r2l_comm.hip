// Multi-GPU assembly of row-sharded frames: the ONE collective of the path (SURVEY 8(b) seam 3, 8(e)), as a C-ABI
// entry point over RCCL.  The reference has no counterpart (it renders on one GPU, main.py:473).
//
// RCCL is bound at run time (dlsym / dlopen), not at link time: a single-GPU client of this library needs no RCCL,
// and a PyTorch process already carries its own copy -- the copy that is already mapped is preferred.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <link.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/r2l_hip.h"
#include "r2l_host_util.h"

namespace {
struct Api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string where;
};
Api g_api;
std::once_flag g_once;

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        *static_cast<std::string*>(data) = info->dlpi_name;
        return 1;
    }
    return 0;
}

void bind_rccl() {
    void* h = nullptr;
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !h && i < 3; ++i) {
        h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (h) loaded = names[i];
    }
    if (!h) {
        const char* why = dlerror();     // captured here: a later dlerror() returns NULL once the message was consumed
        g_api.where = why ? why : "dlopen failed without a message";
        return;
    }
    g_api.where = loaded;
#define BIND(f) g_api.f = reinterpret_cast<decltype(g_api.f)>(dlsym(h, "nccl" #f))
    BIND(GetUniqueId); BIND(CommInitRank); BIND(CommDestroy); BIND(AllGather); BIND(Broadcast); BIND(GroupStart); BIND(GroupEnd);
    BIND(GetErrorString);
#undef BIND
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllGather && g_api.Broadcast &&
               g_api.GroupStart && g_api.GroupEnd && g_api.GetErrorString;
    if (!g_api.ok) g_api.where = loaded + " lacks one of the nccl* entry points this library binds";
}

int need_rccl() {
    std::call_once(g_once, bind_rccl);
    if (!g_api.ok) return r2l_set_error(R2L_ENOGPU, "RCCL is not available (librccl.so could not be bound: %s)", g_api.where.c_str());
    return R2L_OK;
}
}  // namespace

struct r2l_comm {
    ncclComm_t comm;
    int rank, world;
};

#define NCHK(call, what)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) return r2l_set_error(R2L_EHIP, what ": %s", g_api.GetErrorString(r_));            \
    } while (0)

int r2l_comm_available(void) { return need_rccl(); }

int r2l_comm_unique_id(char* id_out128) {
    if (!id_out128) return r2l_set_error(R2L_EINVAL, "NULL id");
    int rc = need_rccl();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCHK(g_api.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out128, &id, 128);
    return R2L_OK;
}

int r2l_comm_create(r2l_comm** out, int rank, int world, const char* id128) {
    if (!out) return r2l_set_error(R2L_EINVAL, "r2l_comm_create: out is NULL");
    *out = nullptr;
    if (!id128) return r2l_set_error(R2L_EINVAL, "r2l_comm_create: the 128-byte id is NULL (rank 0 makes it with r2l_comm_unique_id)");
    if (world < 1 || rank < 0 || rank >= world) return r2l_set_error(R2L_EINVAL, "r2l_comm_create: rank %d is not in [0, world = %d)", rank, world);
    int rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    NCHK(g_api.CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    r2l_comm* c = new r2l_comm();
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    *out = c;
    return R2L_OK;
}

void r2l_comm_destroy(r2l_comm* c) {
    if (!c) return;
    if (g_api.ok && c->comm) (void)g_api.CommDestroy(c->comm);
    delete c;
}

// rows [r0, r1) of rank `rank` (the first H % world ranks get one more): the same rule as dist.row_shard
static void row_shard(int H, int rank, int world, int* r0, int* r1) {
    const int base = H / world, rem = H % world;
    *r0 = rank * base + (rank < rem ? rank : rem);
    *r1 = *r0 + base + (rank < rem ? 1 : 0);
}

int r2l_gather_image(r2l_comm* c, const float* local_rows_dev, float* full_image_dev, int n_frames, int H, int row_floats,
                     void* stream) {
    if (!c) return r2l_set_error(R2L_EINVAL, "r2l_gather_image: comm is NULL");
    if (!local_rows_dev || !full_image_dev) return r2l_set_error(R2L_EINVAL, "r2l_gather_image: NULL device pointer");
    if (n_frames < 1 || H < 1 || row_floats < 1)
        return r2l_set_error(R2L_EINVAL, "r2l_gather_image: n_frames=%d H=%d row_floats=%d must all be >= 1", n_frames, H, row_floats);
    if (H < c->world) return r2l_set_error(R2L_EINVAL, "H=%d rows cannot be sharded over %d ranks", H, c->world);
    hipStream_t s = (hipStream_t)stream;
    int r0, r1;
    row_shard(H, c->rank, c->world, &r0, &r1);
    const size_t mine = (size_t)(r1 - r0) * row_floats, frame = (size_t)H * row_floats;
    // ONE grouped launch.  Frame f of the result is full + f*frame; rank r's rows land at its row offset, so the
    // frames come out in [frame][row] order with no copy behind the collective.
    NCHK(g_api.GroupStart(), "ncclGroupStart");
    // an error inside the bracket must still close it: an open group would silently queue every later collective of the
    // thread, torch.distributed's included
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    if (H % c->world == 0) {
        for (int f = 0; f < n_frames && bad == ncclSuccess; ++f) {
            bad = g_api.AllGather(local_rows_dev + (size_t)f * mine, full_image_dev + (size_t)f * frame, mine, ncclFloat, c->comm, s);
            what = "ncclAllGather";
        }
    } else {  // ragged shards: every rank broadcasts its rows into place
        for (int f = 0; f < n_frames && bad == ncclSuccess; ++f)
            for (int r = 0; r < c->world && bad == ncclSuccess; ++r) {
                int a, b;
                row_shard(H, r, c->world, &a, &b);
                float* dst = full_image_dev + (size_t)f * frame + (size_t)a * row_floats;
                const float* src = r == c->rank ? local_rows_dev + (size_t)f * mine : dst;
                bad = g_api.Broadcast(src, dst, (size_t)(b - a) * row_floats, ncclFloat, r, c->comm, s);
                what = "ncclBroadcast";
            }
    }
    const ncclResult_t ended = g_api.GroupEnd();
    if (bad != ncclSuccess) return r2l_set_error(R2L_EHIP, "%s: %s", what, g_api.GetErrorString(bad));
    if (ended != ncclSuccess) return r2l_set_error(R2L_EHIP, "ncclGroupEnd: %s", g_api.GetErrorString(ended));
    return R2L_OK;
}

// Multi-GPU exchange of the final pose records: one RCCL all-gather over xGMI per batch (the reference
// has no multi-GPU call site -- tools/5_evaluation_bop_basic.py:289-304 walks its detections one by one on one GPU).
//
// Detections are independent, so every rank runs the whole pipeline on its shard and nothing crosses GPUs on the data path.
// The only exchange is the (R, t, score) records at the end: 168-byte p2p_pose structs (they carry the score_type-2 mask sums), n_max per rank, gathered DEVICE to DEVICE
// on the communicator's own stream once the batch's tail has finished (the collect has waited for it and taken the operand-range verdict;
// the NEXT batch's tail, already queued on the tail stream, is not waited for) -- the records never visit the host before the gather --
// and copied to the host once, after it: an event behind the copy, not a stream synchronisation.
// 2048 detections on 8 ranks = 30 KB per rank: latency-bound, so ring / tree and bucket sizes are irrelevant.
//
// RCCL is bound at run time (dlopen), not at link time: the library must load on machines without RCCL (every single-GPU user), and
// a host program that already carries an RCCL -- PyTorch's wheel ships its own librccl.so -- must end up with ONE copy in the process,
// so an already-mapped librccl is reused before the system one is opened.
#include "pipeline.h"

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <cstring>

namespace p2p {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return P2P_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    char path[512] = {0};
};

static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
        strncpy(static_cast<char*>(data), info->dlpi_name, 511);
        return 1;
    }
    return 0;
}

static Rccl* rccl()
{
    static Rccl R;
    static bool tried = false;
    if (tried) return R.handle ? &R : nullptr;
    tried = true;
    char loaded[512] = {0};
    dl_iterate_phdr(find_loaded_rccl, loaded);
    const char* cands[] = {loaded, getenv("P2P_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : cands) {
        if (!c || !*c) continue;
        R.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (R.handle) { strncpy(R.path, c, sizeof(R.path) - 1); break; }
    }
    if (!R.handle) { set_error("RCCL is not available: %s", dlerror()); return nullptr; }
#define P2P_SYM(field, name)                                                             \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name));              \
    if (!R.field) { set_error("%s lacks %s", R.path, name); dlclose(R.handle); R.handle = nullptr; return nullptr; }
    P2P_SYM(get_unique_id, "ncclGetUniqueId")
    P2P_SYM(comm_init_rank, "ncclCommInitRank")
    P2P_SYM(comm_destroy, "ncclCommDestroy")
    P2P_SYM(all_gather, "ncclAllGather")
    P2P_SYM(error_string, "ncclGetErrorString")
#undef P2P_SYM
    return &R;
}

struct Comm {
    Rccl* lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    DevBuf send, recv;
    PinnedBuf h_recv;
    hipStream_t stream = nullptr;      // the gathers run here
    hipEvent_t landed = nullptr;       // behind the D2H copy of a gather
    ~Comm()
    {
        if (comm && lib) (void)lib->comm_destroy(comm);
        send.release(); recv.release(); h_recv.release();
        if (landed) hipEventDestroy(landed);
        if (stream) hipStreamDestroy(stream);
    }
};

// send[src_index of detection i] = poses[i]  (the batch is processed sorted by object; the gathered records are in the caller's order).
// force_status != 0: the batch left the split-f16 operand range -- the records travel with that status (P2P_POSE_RANGE) so that the peers see
// detections whose poses are not to be used instead of ordinary-looking garbage.
__global__ void gather_pack_kernel(const DetInfo* __restrict__ dets, const p2p_pose* __restrict__ poses, int n, p2p_pose* __restrict__ send, int force_status,
                                   const unsigned long long* __restrict__ mstat)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p2p_pose r = poses[i];
    if (force_status) r.status = force_status;
    if (mstat) {          // score_type-2 sums of the batch (mask_iou_kernel: {inter, |det mask|, |valid mask|}) -> {inter, union, valid}
        const long long inter = (long long)mstat[3 * i], dc = (long long)mstat[3 * i + 1], vc = (long long)mstat[3 * i + 2];
        r.mask_stats[0] = inter; r.mask_stats[1] = dc + vc - inter; r.mask_stats[2] = vc;
    }
    send[dets[i].src_index] = r;
}

int comm_world(const Comm& C) { return C.world; }

// s == nullptr: this rank has nothing valid to send this step (an empty shard, or a local error the caller reports after the collective):
// it still joins, with n_max padding records.  `ts` is only used when the communicator has no stream of its own.
int comm_gather(Ctx& X, Comm& C, Slot* s, bool range_event, hipStream_t ts, int n_max, p2p_pose* gathered)
{
    int rc;
    (void)X;
    if (C.stream) ts = C.stream;
    const size_t rec = sizeof(p2p_pose), per = (size_t)n_max * rec;
    // The receive side must exist to join at all: an allocation failure HERE (first step, or a larger n_max than before) is the one error that
    // cannot be reported through the collective.  Everything after it is: a local failure is recorded, this rank still enters ncclAllGather
    // with padding records, and the error is returned after the collective -- the peers are never left waiting.
    if ((rc = C.send.reserve(per)) || (rc = C.recv.reserve(per * C.world)) || (rc = C.h_recv.reserve(per * C.world))) return rc;
    int local = P2P_OK;
    char local_msg[256] = "";
    auto fail = [&](const char* what, hipError_t e) {
        if (local == P2P_OK) { local = P2P_ERR_HIP; snprintf(local_msg, sizeof(local_msg), "comm_gather: %s failed: %s (this rank joined the collective with padding records)", what, hipGetErrorString(e)); }
    };
    hipError_t e;
    if (!C.landed && (e = hipEventCreateWithFlags(&C.landed, hipEventDisableTiming)) != hipSuccess) { C.landed = nullptr; fail("hipEventCreate", e); }
    if ((e = hipMemsetAsync(C.send.p, 0xFF, per, ts)) != hipSuccess) fail("hipMemsetAsync", e);      // padding records: status = -1 (P2P_POSE_ABSENT)
    static const bool inject = dev_env("P2P_COMM_INJECT_PACK_FAILURE") != nullptr;                     // development builds: tests/test_comm_gpu.py
    if (local == P2P_OK && s && s->n > 0) {
        if (inject) fail("gather_pack_kernel (injected)", hipErrorLaunchFailure);
        else {
            hipLaunchKernelGGL(gather_pack_kernel, dim3((s->n + 255) / 256), dim3(256), 0, ts, s->det.as<DetInfo>(), s->poses.as<p2p_pose>(), s->n, C.send.as<p2p_pose>(),
                               range_event ? (int)P2P_POSE_RANGE : 0, s->opt.det_mask ? s->mstat.as<unsigned long long>() : nullptr);
            if ((e = hipGetLastError()) != hipSuccess) { fail("gather_pack_kernel", e); (void)hipMemsetAsync(C.send.p, 0xFF, per, ts); }
        }
    }
    const ncclResult_t r = C.lib->all_gather(C.send.p, C.recv.p, per, ncclChar, C.comm, ts);
    if (r != ncclSuccess) { set_error("ncclAllGather failed: %s", C.lib->error_string(r)); return P2P_ERR_HIP; }
    HIP_TRY(hipMemcpyAsync(C.h_recv.p, C.recv.p, per * C.world, hipMemcpyDeviceToHost, ts));
    if (C.landed) {
        HIP_TRY(hipEventRecord(C.landed, ts));
        HIP_TRY(hipEventSynchronize(C.landed));
    } else HIP_TRY(hipStreamSynchronize(ts));
    memcpy(gathered, C.h_recv.p, per * C.world);
    if (local != P2P_OK) { set_error("%s", local_msg); return local; }
    return P2P_OK;
}

}  // namespace p2p

using namespace p2p;

extern "C" {

int p2p_comm_unique_id(char* id)
{
    if (!id) { set_error("p2p_comm_unique_id: null argument"); return P2P_ERR_INVALID_ARG; }
    Rccl* L = rccl();
    if (!L) return P2P_ERR_HIP;
    static_assert(P2P_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    const ncclResult_t r = L->get_unique_id(&u);
    if (r != ncclSuccess) { set_error("ncclGetUniqueId failed: %s", L->error_string(r)); return P2P_ERR_HIP; }
    memcpy(id, u.internal, P2P_COMM_ID_BYTES);
    return P2P_OK;
}

int p2p_comm_create(p2p_ctx* ctx, int rank, int world, const char* id, p2p_comm** out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) { set_error("p2p_comm_create: bad arguments"); return P2P_ERR_INVALID_ARG; }
    *out = nullptr;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    Rccl* L = rccl();
    if (!L) return P2P_ERR_HIP;
    Comm* C = new Comm();
    C->lib = L; C->rank = rank; C->world = world; C->device = c->device;
    ncclUniqueId u;
    memcpy(u.internal, id, P2P_COMM_ID_BYTES);
    const ncclResult_t r = L->comm_init_rank(&C->comm, world, u, rank);
    if (r != ncclSuccess) { set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, L->error_string(r)); C->comm = nullptr; delete C; return P2P_ERR_HIP; }
    if (hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking) != hipSuccess) C->stream = nullptr;      // falls back to the batch's tail stream
    *out = reinterpret_cast<p2p_comm*>(C);
    return P2P_OK;
}

void p2p_comm_destroy(p2p_comm* comm)
{
    if (!comm) return;
    Comm* C = reinterpret_cast<Comm*>(comm);
    hipSetDevice(C->device);
    delete C;
}

const char* p2p_comm_library(void)
{
    Rccl* L = rccl();
    return L ? L->path : "";
}

}  // extern "C"

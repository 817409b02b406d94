// dcs_gather: the one exchange of the multi-GPU path ("RCCL only for the final gather") behind the C ABI.
// Every rank has separated its own tiles / clips; what travels is the scripts' output, int16 PCM (or any bytes).  RCCL is
// loaded on first use (dlopen, no link-time dependency: a single-GPU user never maps librccl), the communicator is the
// caller's -- one process per GPU, created with ncclCommInitRank by whatever launcher the host uses.
#include "dcs_internal.h"

#include <dlfcn.h>
#include <stddef.h>

namespace {

// the slice of rccl.h this file calls (ncclResult_t and ncclDataType_t are ints in the ABI; ncclSuccess = 0, ncclInt8 = 0)
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_send)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_void)();
typedef int (*fn_comm_int)(const void*, int*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* lib = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_void group_start = nullptr, group_end = nullptr;
    fn_comm_int count = nullptr, user_rank = nullptr;
    fn_errstr errstr = nullptr;
    bool ok = false;
    char why[256] = "";      // why RCCL is unusable, captured at load time
};

Rccl* rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            x.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) {
            const char* e = dlerror();               // read ONCE: dlerror() clears the error state it returns
            snprintf(x.why, sizeof(x.why), "%s", e ? e : "librccl.so not found");
            return x;
        }
        x.all_gather = (fn_all_gather)dlsym(x.lib, "ncclAllGather");
        x.send = (fn_send)dlsym(x.lib, "ncclSend");
        x.recv = (fn_recv)dlsym(x.lib, "ncclRecv");
        x.group_start = (fn_void)dlsym(x.lib, "ncclGroupStart");
        x.group_end = (fn_void)dlsym(x.lib, "ncclGroupEnd");
        x.count = (fn_comm_int)dlsym(x.lib, "ncclCommCount");
        x.user_rank = (fn_comm_int)dlsym(x.lib, "ncclCommUserRank");
        x.errstr = (fn_errstr)dlsym(x.lib, "ncclGetErrorString");
        x.ok = x.all_gather && x.send && x.recv && x.group_start && x.group_end && x.count && x.user_rank;
        if (!x.ok) snprintf(x.why, sizeof(x.why), "librccl.so lacks one of ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / "
                                                  "ncclGroupEnd / ncclCommCount / ncclCommUserRank");
        return x;
    }();
    return &r;
}

}  // namespace

#define DCS_RCCL(call_)                                                                                    \
    do {                                                                                                   \
        const int rc__ = (call_);                                                                          \
        if (rc__ != 0) DCS_FAIL(DCS_EHIP, "%s: %s", #call_, R->errstr ? R->errstr(rc__) : "rccl error");   \
    } while (0)

extern "C" int dcs_gather(dcs_ctx* ctx, void* nccl_comm, const void* shard_d, int64_t bytes, void* full_d, int root) {
    if (!ctx || !nccl_comm || !shard_d) DCS_FAIL(DCS_EINVAL, "dcs_gather: null argument");
    if (bytes < 0) DCS_FAIL(DCS_EINVAL, "dcs_gather: %lld bytes", (long long)bytes);
    Rccl* R = rccl();
    if (!R->ok) DCS_FAIL(DCS_EUNSUPPORTED, "dcs_gather: RCCL unavailable (%s)", R->why);
    DCS_ON_DEVICE(ctx->device);
    int n = 0, rank = -1;
    DCS_RCCL(R->count(nccl_comm, &n));
    DCS_RCCL(R->user_rank(nccl_comm, &rank));
    if (root >= n) DCS_FAIL(DCS_EINVAL, "dcs_gather: root %d of %d ranks", root, n);
    if ((root < 0 || rank == root) && !full_d) DCS_FAIL(DCS_EINVAL, "dcs_gather: this rank receives and full_d is null");
    if (bytes == 0) return DCS_OK;
    if (root < 0) {   // every rank ends with every rank's shard, rank order
        DCS_RCCL(R->all_gather(shard_d, full_d, (size_t)bytes, 0 /* ncclInt8 */, nccl_comm, ctx->stream));
        return DCS_OK;
    }
    // to one rank (the writer): RCCL has no gather primitive -- grouped point-to-point, one xGMI hop per shard
    DCS_RCCL(R->group_start());
    int rc_send = 0, rc_recv = 0;
    if (rank == root) {
        for (int r = 0; r < n && rc_recv == 0; ++r) {
            if (r == rank) continue;
            rc_recv = R->recv((char*)full_d + (size_t)r * (size_t)bytes, (size_t)bytes, 0, r, nccl_comm, ctx->stream);
        }
    } else {
        rc_send = R->send(shard_d, (size_t)bytes, 0, root, nccl_comm, ctx->stream);
    }
    const int rc_end = R->group_end();
    if (rc_send || rc_recv || rc_end)
        DCS_FAIL(DCS_EHIP, "dcs_gather: %s", R->errstr ? R->errstr(rc_send ? rc_send : (rc_recv ? rc_recv : rc_end)) : "rccl error");
    if (rank == root && (char*)full_d + (size_t)rank * (size_t)bytes != (const char*)shard_d)
        DCS_HIP(hipMemcpyAsync((char*)full_d + (size_t)rank * (size_t)bytes, shard_d, (size_t)bytes, hipMemcpyDeviceToDevice,
                               ctx->stream));
    return DCS_OK;
}

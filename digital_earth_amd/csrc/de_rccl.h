// de_rccl.h — the collectives of the frame partitions behind the C ABI: RCCL bound at run time, the sum-reduce of the tile partition, the
// rank-ordered gather-sum of the sample partition (include/digital_earth.h: de_comm_*, de_reduce*, de_last_reduce_ms).
#pragma once
#include "de_launch.h"

namespace {

// ---- RCCL, bound at run time (dlopen): a single-GPU user never loads it, and a process that already holds RCCL (PyTorch
// ships its own copy) keeps ONE instance — the already-loaded library is preferred over a second load.
struct UniqueId128;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, struct UniqueId128, int) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;      // the four below: de_reduce_ordered only (optional at load)
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;   // optional: which rank of an external communicator this context is
    const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId128 { char internal[128]; };        // ncclUniqueId (rccl.h:43), passed BY VALUE to ncclCommInitRank
Rccl g_rccl;
std::mutex g_rccl_mutex;       // contexts of different threads may reach the first use together
int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.h) return DE_OK;
    void* h = nullptr;
    bool only_env = false;
    if (const char* e = getenv("DE_RCCL_PATH")) {
        h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        only_env = getenv("DE_RCCL_STRICT") != nullptr;   // test hook: no search beyond DE_RCCL_PATH
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h && !only_env; ++pass)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) {
        const char* why = dlerror();      // ONE call: dlerror() clears the message it returns
        return fail(DE_ERR_STATE, std::string("cannot load librccl.so: ") + (why ? why : "not found"));
    }
    Rccl r; r.h = h;
    r.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, UniqueId128, int))dlsym(h, "ncclCommInitRank");
    r.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t))dlsym(h, "ncclReduce");
    r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    r.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    r.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    r.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
    if (!r.GetUniqueId || !r.CommInitRank || !r.Reduce || !r.CommDestroy || !r.GetErrorString)
        return fail(DE_ERR_STATE, "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy / ncclGetErrorString");
    g_rccl = r;
    return DE_OK;
}
#define RCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        int r_ = (expr);                                                                                         \
        if (r_ != 0) return fail(DE_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));             \
    } while (0)

}  // namespace

extern "C" {

/* ---- in-library collective (SURVEY §8b `de_reduce`, §8e) */
int de_comm_unique_id(void* out128) {
    if (!out128) return fail(DE_ERR_INVALID, "out is null");
    int rc = load_rccl();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(out128));
    return DE_OK;
}
int de_comm_init(de_ctx* c, const void* id128, int rank, int world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(DE_ERR_INVALID, "bad communicator arguments");
    if (c->comm) return fail(DE_ERR_STATE, "the context already has a communicator: de_comm_destroy first");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    UniqueId128 id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return DE_OK;
}
int de_comm_destroy(de_ctx* c) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    if (!c->comm) return DE_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rc2 = sync_all(c); if (rc2) return rc2; }
    RCCL_TRY(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    return DE_OK;
}
namespace {
int reduce_impl(de_ctx* c, void* comm, int root, bool progressive) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    void* use = comm ? comm : c->comm;
    if (!use) return fail(DE_ERR_STATE, "no communicator: pass an ncclComm_t or call de_comm_init");
    if (root < 0 || (!comm && root >= c->comm_world)) return fail(DE_ERR_INVALID, "root out of range");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = join_slots(c);
    if (rc) return rc;
    rc = frame_status(c);      // what is known so far; the host synchronises (and asks again) before it reads the assembled frame
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    float* recv = c->d_hdr;
    if (progressive) {
        // out of place: every rank keeps accumulating into its own buffer; the root receives the assembled frame elsewhere
        if (!c->d_assembled) {
            HIP_TRY(hipMalloc(&c->d_assembled, count * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->d_assembled, 0, count * sizeof(float), c->stream));      // non-root ranks: scratch, but never uninitialised
        }
        recv = c->d_assembled;
    } else {
        touched_hdr(c);
    }
    if (!c->ev_r0) { HIP_TRY(hipEventCreate(&c->ev_r0)); HIP_TRY(hipEventCreate(&c->ev_r1)); }
    HIP_TRY(hipEventRecord(c->ev_r0, c->stream));
    // every rank's buffer is its own tiles plus zeros elsewhere; the root ends with the assembled frame
    RCCL_TRY(g_rccl.Reduce(c->d_hdr, recv, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, root, use, c->stream));
    HIP_TRY(hipEventRecord(c->ev_r1, c->stream));
    c->reduce_timing_valid = true; c->last_reduce_standin = false;
    if (progressive) {
        touched_hdr(c);                        // the collective READS d_hdr: the next accumulate_kernel is ordered after it
        // only the root receives the assembled frame: elsewhere ncclReduce leaves recvbuff alone, and the display / fetch entry
        // points keep reading the rank's own sums
        int my_rank = c->comm_rank;
        if (comm) { if (!g_rccl.CommUserRank || g_rccl.CommUserRank(use, &my_rank) != 0) my_rank = root; }      // cannot ask: behave as every rank did before (the root is right, the others show scratch)
        if (my_rank == root) c->display_src = c->d_assembled;
    }
    return DE_OK;
}
}  // namespace

namespace {
// The collective of a SAMPLE partition (de_set_sample_partition): every pixel's sum has a term on every rank, so the order of the additions
// is part of the result.  ncclReduce leaves that order to the algorithm RCCL picks; here the ranks' buffers are GATHERED on the root (one
// ncclSend / ncclRecv pair per peer in one group: the root's ingress runs over its xGMI links in parallel) and added by ordered_sum_kernel
// in rank order, so the assembled frame is a function of the partition alone (tests hold it to the oracle's partial sums added the same way).
// the root's buffers: one slot of W*H*3 floats per rank (the root's own stays unused: its part is read in place), the assembled frame when out of place
int ordered_root_buffers(de_ctx* c, int world, bool out_of_place) {
    const size_t count = (size_t)c->W * c->H * 3;
    if (c->gather_world < world) {
        { int rc2 = sync_all(c); if (rc2) return rc2; }
        if (c->d_gather) hipFree(c->d_gather);
        c->d_gather = nullptr; c->gather_world = 0;
        if (hipMalloc(&c->d_gather, (size_t)world * count * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail(DE_ERR_NOMEM, "no room for the gather buffer of de_reduce_ordered"); }
        c->gather_world = world;
    }
    if (out_of_place && !c->d_assembled) HIP_TRY(hipMalloc(&c->d_assembled, count * sizeof(float)));
    return DE_OK;
}
// the root's sum, on the context stream: part r = rank r's gathered buffer, the root's own accumulation buffer in place
int launch_ordered_sum(de_ctx* c, int world, int root, bool out_of_place) {
    const size_t count = (size_t)c->W * c->H * 3;
    OrderedSumArgs o;
    for (int r = 0; r < world; ++r) o.part[r] = (r == root) ? c->d_hdr : c->d_gather + (size_t)r * count;
    o.out = out_of_place ? c->d_assembled : c->d_hdr; o.n_parts = world; o.n4 = count / 4; o.n = count;
    hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)c->n_cus * 8u), dim3(256), 0, c->stream, o);
    HIP_TRY(hipGetLastError());
    if (out_of_place) c->display_src = c->d_assembled;
    return DE_OK;
}
int reduce_ordered_impl(de_ctx* c, void* comm, int root, bool out_of_place) {
    if (!c) return fail(DE_ERR_INVALID, "null context");
    void* use = comm ? comm : c->comm;
    if (!use) return fail(DE_ERR_STATE, "no communicator: pass an ncclComm_t or call de_comm_init");
    int rc = load_rccl();
    if (rc) return rc;
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(DE_ERR_STATE, "librccl.so lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    int world = c->comm_world, my_rank = c->comm_rank;
    if (comm) {
        if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(DE_ERR_STATE, "librccl.so lacks ncclCommCount / ncclCommUserRank: an external communicator cannot be asked for its shape");
        RCCL_TRY(g_rccl.CommCount(use, &world));
        RCCL_TRY(g_rccl.CommUserRank(use, &my_rank));
    }
    if (root < 0 || root >= world) return fail(DE_ERR_INVALID, "root out of range");
    if (world > 16) return fail(DE_ERR_INVALID, "de_reduce_ordered: at most 16 ranks");
    HIP_TRY(hipSetDevice(c->device));
    rc = join_slots(c);
    if (rc) return rc;
    rc = frame_status(c);
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    if (!c->ev_r0) { HIP_TRY(hipEventCreate(&c->ev_r0)); HIP_TRY(hipEventCreate(&c->ev_r1)); }
    if (my_rank == root) { int rc2 = ordered_root_buffers(c, world, out_of_place); if (rc2) return rc2; }
    HIP_TRY(hipEventRecord(c->ev_r0, c->stream));
    if (my_rank == root) {
        RCCL_TRY(g_rccl.GroupStart());
        for (int r = 0; r < world; ++r)
            if (r != root) { int e = g_rccl.Recv(c->d_gather + (size_t)r * count, count, /*ncclFloat32*/ 7, r, use, c->stream); if (e != 0) { g_rccl.GroupEnd(); return fail(DE_ERR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(e)); } }
        RCCL_TRY(g_rccl.GroupEnd());
        { int rc2 = launch_ordered_sum(c, world, root, out_of_place); if (rc2) return rc2; }
    } else {
        RCCL_TRY(g_rccl.Send(c->d_hdr, count, /*ncclFloat32*/ 7, root, use, c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev_r1, c->stream));
    c->reduce_timing_valid = true; c->last_reduce_standin = false;
    touched_hdr(c);                        // the collective reads (root, in place: rewrites) d_hdr: the next accumulate_kernel is ordered after it
    return DE_OK;
}
}  // namespace
int de_reduce_ordered(de_ctx* c, void* comm, int root, int out_of_place) { return reduce_ordered_impl(c, comm, root, out_of_place != 0); }

/* include/digital_earth_debug.h: the root's half of de_reduce_ordered with the parts coming from the host instead of ncclRecv */
int de_debug_ordered_sum(de_ctx* c, const float* parts, int n_parts, int root, int out_of_place, float* out) {
    if (!c || !parts || !out || n_parts < 2 || n_parts > 16 || root < 0 || root >= n_parts) return fail(DE_ERR_INVALID, "de_debug_ordered_sum: 2..16 parts, 0 <= root < n_parts");
    HIP_TRY(hipSetDevice(c->device));
    int rc = join_slots(c);
    if (rc) return rc;
    rc = ordered_root_buffers(c, n_parts, out_of_place != 0);
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    touched_hdr(c);
    for (int r = 0; r < n_parts; ++r)
        HIP_TRY(hipMemcpyAsync(r == root ? c->d_hdr : c->d_gather + (size_t)r * count, parts + (size_t)r * count, count * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = launch_ordered_sum(c, n_parts, root, out_of_place != 0);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, out_of_place ? c->d_assembled : c->d_hdr, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DE_OK;
}

/* include/digital_earth_debug.h: a one-GPU stand-in for the frame's collective, where de_reduce goes on the context stream */
int de_debug_standin_reduce(de_ctx* c, int extra_copies) {
    if (!c || extra_copies < 0 || extra_copies > 15) return fail(DE_ERR_INVALID, "de_debug_standin_reduce: 0..15 extra copies");
    HIP_TRY(hipSetDevice(c->device));
    int rc = join_slots(c);
    if (rc) return rc;
    rc = frame_status(c);
    if (rc) return rc;
    const size_t count = (size_t)c->W * c->H * 3;
    if (!c->d_standin) {
        // operand, output and copy target of its own ([3][H][W][3]): never the assembled buffer a progressive display may be reading
        HIP_TRY(hipMalloc(&c->d_standin, 3 * count * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->d_standin, 0, 3 * count * sizeof(float), c->stream));
    }
    const unsigned e = c->standin_count % 16u;
    if (!c->ev_standin[e][0]) { HIP_TRY(hipEventCreate(&c->ev_standin[e][0])); HIP_TRY(hipEventCreate(&c->ev_standin[e][1])); }
    HIP_TRY(hipEventRecord(c->ev_standin[e][0], c->stream));
    for (int k = 0; k < extra_copies; ++k) HIP_TRY(hipMemcpyAsync(c->d_standin + 2 * count, c->d_standin, count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    OrderedSumArgs o;
    o.part[0] = c->d_hdr; o.part[1] = c->d_standin; o.out = c->d_standin + count; o.n_parts = 2; o.n4 = count / 4; o.n = count;
    hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)c->n_cus * 8u), dim3(256), 0, c->stream, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_standin[e][1], c->stream));
    c->standin_count++;
    c->reduce_timing_valid = true; c->last_reduce_standin = true;
    touched_hdr(c);                        // it READS d_hdr, like the collective: the next accumulate_kernel is ordered after it
    return DE_OK;
}

int de_reduce(de_ctx* c, void* comm, int root) { return reduce_impl(c, comm, root, false); }
int de_reduce_progressive(de_ctx* c, void* comm, int root) { return reduce_impl(c, comm, root, true); }
int de_last_reduce_ms(de_ctx* c, float* ms) {
    if (!c || !ms) return fail(DE_ERR_INVALID, "null argument");
    if (!c->reduce_timing_valid) return fail(DE_ERR_STATE, "no de_reduce has been issued");
    HIP_TRY(hipSetDevice(c->device));
    if (c->last_reduce_standin) {
        // de_debug_standin_reduce: the mean over the last (up to 16) stand-ins, from the moment the stream reached each to its end
        const unsigned n = c->standin_count < 16u ? c->standin_count : 16u;
        double sum = 0.0;
        for (unsigned k = 0; k < n; ++k) {
            float t = 0.f;
            HIP_TRY(hipEventSynchronize(c->ev_standin[k][1]));
            HIP_TRY(hipEventElapsedTime(&t, c->ev_standin[k][0], c->ev_standin[k][1]));
            sum += t;
        }
        *ms = n ? (float)(sum / n) : 0.f;
        return DE_OK;
    }
    HIP_TRY(hipEventSynchronize(c->ev_r1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_r0, c->ev_r1));
    return DE_OK;
}
}  // extern "C"


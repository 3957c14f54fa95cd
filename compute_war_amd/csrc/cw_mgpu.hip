// cw_mgpu.hip -- several GPUs of one node behind the C ABI (the reference's dormant --gpu-offload seam,
// src/hashandcompress/HashAndCompress.cpp:305,331, has one device at most).
//
// The path shards by blocks: device g of G owns the contiguous range [g*N/G, (g+1)*N/G) of the block index space and
// runs the ordinary single-device entry points on it; nothing on the data path crosses devices.  The one exchange is the
// result gather at the end of a pass -- every device's digests (ncclAllGather) and byte totals (ncclAllReduce, sum) --
// over RCCL, i.e. over the xGMI links between the GPUs of the node.  One process drives all devices
// (ncclCommInitAll), one host thread per device does the work (cw_set_device).
//
// RCCL is resolved at run time (dlopen of librccl.so.1) and only here: a process that never creates a cw_mgpu object
// never loads it, so the library can live beside a framework that carries its own RCCL.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/cw_hashcompress.h"

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_lock;
Rccl g_rccl;
thread_local char t_mgpu_err[256] = "";

bool load_rccl()
{
    std::lock_guard<std::mutex> g(g_rccl_lock);
    if (g_rccl.so) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *so = nullptr;
    for (const char *n : names)
        if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!so) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "RCCL not loadable: %s", dlerror()); return false; }
    Rccl r;
    r.so = so;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(so, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(so, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(so, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(so, "ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(so, "ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(so, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(so, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
        snprintf(t_mgpu_err, sizeof t_mgpu_err, "RCCL lacks a required symbol");
        dlclose(so);
        return false;
    }
    g_rccl = r;
    return true;
}

} // namespace

struct cw_mgpu {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};

extern "C" {

void cw_shard_range(size_t n, int g, int G, size_t *first, size_t *last)
{
    if (G < 1) G = 1;
    if (g < 0) g = 0;
    if (g >= G) g = G - 1;
    // [g*n/G, (g+1)*n/G): contiguous, covers [0, n), sizes differ by at most one
    const unsigned __int128 a = (unsigned __int128)n * (unsigned)g / (unsigned)G, b = (unsigned __int128)n * (unsigned)(g + 1) / (unsigned)G;
    if (first) *first = (size_t)a;
    if (last) *last = (size_t)b;
}

const char *cw_mgpu_last_error(void) { return t_mgpu_err; }

cw_mgpu_t *cw_mgpu_create(const int *devices, int ndev)
{
    if (!devices || ndev < 1 || ndev > 16) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "cw_mgpu_create: 1..16 devices"); return nullptr; }
    for (int i = 0; i < ndev; i++)
        for (int k = 0; k < i; k++)
            if (devices[i] == devices[k]) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "cw_mgpu_create: device %d listed twice", devices[i]); return nullptr; }
    const int before = cw_get_device();
    // every exit path leaves the calling thread on the device it came with (cw_init and hipSetDevice below switch it)
    auto back = [&]() { if (before >= 0) (void)cw_set_device(before); };
    for (int i = 0; i < ndev; i++)
        if (cw_init(devices[i]) != CW_OK) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "%s", cw_last_error()); back(); return nullptr; }
    if (!load_rccl()) { back(); return nullptr; }
    cw_mgpu *m = new cw_mgpu;
    m->devices.assign(devices, devices + ndev);
    m->comms.assign((size_t)ndev, nullptr);
    m->streams.assign((size_t)ndev, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), ndev, devices);
    if (r != ncclSuccess) {
        snprintf(t_mgpu_err, sizeof t_mgpu_err, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
        delete m;
        back();
        return nullptr;
    }
    for (int i = 0; i < ndev; i++) {
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&m->streams[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
            snprintf(t_mgpu_err, sizeof t_mgpu_err, "stream on device %d: %s", devices[i], hipGetErrorString(hipGetLastError()));
            cw_mgpu_destroy(m);
            back();
            return nullptr;
        }
    }
    (void)cw_set_device(before >= 0 ? before : devices[0]);
    return m;
}

void cw_mgpu_destroy(cw_mgpu_t *m)
{
    if (!m) return;
    for (size_t i = 0; i < m->devices.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (m->streams[i]) { (void)hipStreamSynchronize(m->streams[i]); (void)hipStreamDestroy(m->streams[i]); }
        if (m->comms[i]) (void)g_rccl.CommDestroy(m->comms[i]);
    }
    delete m;
}

int cw_mgpu_ndev(const cw_mgpu_t *m) { return m ? (int)m->devices.size() : 0; }
int cw_mgpu_device(const cw_mgpu_t *m, int rank) { return m && rank >= 0 && rank < (int)m->devices.size() ? m->devices[(size_t)rank] : -1; }

int cw_mgpu_gather(cw_mgpu_t *m, const void *const *d_local, size_t bytes_each, void *const *d_all, uint64_t *const *d_totals, size_t ntotals)
{
    if (!m) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "NULL cw_mgpu"); return CW_ERR_BAD_ARG; }
    const int G = (int)m->devices.size();
    const int before = cw_get_device();
    ncclResult_t r = g_rccl.GroupStart();
    const bool grouped = r == ncclSuccess; // a group that never opened must not be closed
    for (int g = 0; g < G && r == ncclSuccess; g++) {
        if (hipSetDevice(m->devices[(size_t)g]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
        if (bytes_each && d_local && d_all)
            r = g_rccl.AllGather(d_local[g], d_all[g], bytes_each, ncclUint8, m->comms[(size_t)g], m->streams[(size_t)g]);
        if (r == ncclSuccess && ntotals && d_totals)
            r = g_rccl.AllReduce(d_totals[g], d_totals[g], ntotals, ncclUint64, ncclSum, m->comms[(size_t)g], m->streams[(size_t)g]);
    }
    if (grouped) {
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = re;
    }
    hipError_t he = hipSuccess;
    for (int g = 0; g < G; g++) {
        (void)hipSetDevice(m->devices[(size_t)g]);
        const hipError_t e = hipStreamSynchronize(m->streams[(size_t)g]);
        if (e != hipSuccess) he = e;
    }
    if (before >= 0) (void)cw_set_device(before);
    if (r != ncclSuccess) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "RCCL gather: %s", g_rccl.GetErrorString(r)); return CW_ERR_HIP; }
    if (he != hipSuccess) { snprintf(t_mgpu_err, sizeof t_mgpu_err, "gather streams: %s", hipGetErrorString(he)); return CW_ERR_HIP; }
    return CW_OK;
}

} // extern "C"

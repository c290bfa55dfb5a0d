// Multi-GPU exchange step behind the C ABI: the bands every GPU has rendered are gathered to the first GPU over xGMI with
// RCCL and put back into frame order there. Single process, one communicator per device (ncclCommInitAll). RCCL is looked
// up at run time (dlopen), so the library loads on machines without it and reports RT_ERR_NO_DEVICE-style errors instead.
#include "../../include/rt_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>

#include <mutex>
#include <new>
#include <vector>

int rt_set_error(int code, const char* fmt, ...);     // rt_host.cpp

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds an RCCL (PyTorch brings its own) gets that one: same SONAME
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
        r.Gather = (decltype(r.Gather))dlsym(r.handle, "ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Gather && r.GetErrorString;
    });
    return r;
}
}  // namespace

struct rt_comm {
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    bool broken = false;          // a gather failed after some ranks had posted theirs: the communicators were aborted
};

#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r_ = (expr);                                                                                   \
        if (r_ != ncclSuccess) return rt_set_error(RT_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString(r_));  \
    } while (0)
#define HIP_TRY_C(expr)                                                                                             \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return rt_set_error(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

extern "C" int rt_comm_create(int32_t n_devices, const int32_t* devices, rt_comm** out) {
    if (!out || n_devices < 1 || n_devices > 64) return rt_set_error(RT_ERR_INVALID, "rt_comm_create: bad argument");
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return rt_set_error(RT_ERR_NO_DEVICE, "no HIP device");
    if (!rccl().ok) {
        const char* why = dlerror();          // a second call would return NULL: the message is handed out once
        return rt_set_error(RT_ERR_HIP, "RCCL (librccl.so.1) cannot be loaded: %s", why ? why : "symbols missing");
    }
    rt_comm* c = new (std::nothrow) rt_comm();
    if (!c) return rt_set_error(RT_ERR_NOMEM, "out of host memory");
    try {
        for (int i = 0; i < n_devices; ++i) {
            const int d = devices ? devices[i] : i;
            if (d < 0 || d >= visible) { delete c; return rt_set_error(RT_ERR_INVALID, "rt_comm_create: device %d is not visible (%d devices)", d, visible); }
            for (int k : c->devices)
                if (k == d) { delete c; return rt_set_error(RT_ERR_INVALID, "rt_comm_create: device %d listed twice", d); }
            c->devices.push_back(d);
        }
        c->comms.resize((size_t)n_devices, nullptr);
    } catch (...) { delete c; return rt_set_error(RT_ERR_NOMEM, "out of host memory"); }
    int current = 0;
    (void)hipGetDevice(&current);
    ncclResult_t r = rccl().CommInitAll(c->comms.data(), n_devices, c->devices.data());
    (void)hipSetDevice(current);
    if (r != ncclSuccess) { delete c; return rt_set_error(RT_ERR_HIP, "ncclCommInitAll failed: %s", rccl().GetErrorString(r)); }
    *out = c;
    return RT_OK;
}

extern "C" void rt_comm_destroy(rt_comm* c) {
    if (!c) return;
    if (rccl().ok && !c->broken)
        for (ncclComm_t k : c->comms)
            if (k) (void)rccl().CommDestroy(k);
    delete c;
}

extern "C" int32_t rt_comm_size(const rt_comm* c) { return c ? (int32_t)c->devices.size() : 0; }

extern "C" int rt_gather_bands(rt_comm* c, const rt_params* p, void* const* d_parts, void* d_stacked, void* d_frame, int32_t elem_bytes,
                               void* const* streams) {
    if (!c || !p || !d_parts || !d_stacked || !d_frame || elem_bytes <= 0) return rt_set_error(RT_ERR_INVALID, "rt_gather_bands: bad argument");
    const int n = (int)c->devices.size();
    if (p->part_count != n) return rt_set_error(RT_ERR_INVALID, "rt_gather_bands: part_count %d but the communicator has %d devices", p->part_count, n);
    const size_t part_bytes = (size_t)rt_part_rows(p) * (size_t)p->width * (size_t)elem_bytes;
    if (part_bytes == 0) return rt_set_error(RT_ERR_INVALID, "rt_gather_bands: empty frame");
    if (c->broken) return rt_set_error(RT_ERR_HIP, "rt_gather_bands: the communicator is unusable after a failed gather");
    // Everything that can be checked is checked BEFORE the group is opened: a gather that is posted on some ranks of the
    // communicator and not on the others leaves the posted ranks' kernels waiting on their streams for ever.
    for (int r = 0; r < n; ++r)
        if (!d_parts[r]) return rt_set_error(RT_ERR_INVALID, "rt_gather_bands: d_parts[%d] is NULL", r);
    int current = 0;
    HIP_TRY_C(hipGetDevice(&current));
    for (int r = 0; r < n; ++r) {
        const hipError_t e = hipSetDevice(c->devices[r]);
        if (e != hipSuccess) {
            (void)hipSetDevice(current);
            return rt_set_error(RT_ERR_HIP, "rt_gather_bands: cannot select device %d: %s", c->devices[r], hipGetErrorString(e));
        }
    }
    // one gather, every device's call inside one group: each peer writes its bands over its own xGMI link into the root
    {
        const ncclResult_t gs = rccl().GroupStart();
        if (gs != ncclSuccess) { (void)hipSetDevice(current); return rt_set_error(RT_ERR_HIP, "ncclGroupStart failed: %s", rccl().GetErrorString(gs)); }
    }
    ncclResult_t first_err = ncclSuccess;
    int posted = 0;
    for (int r = 0; r < n; ++r) {
        if (hipSetDevice(c->devices[r]) != hipSuccess) { first_err = ncclUnhandledCudaError; break; }
        const ncclResult_t e = rccl().Gather(d_parts[r], r == 0 ? d_stacked : nullptr, part_bytes, ncclUint8, 0, c->comms[r],
                                             (hipStream_t)(streams ? streams[r] : nullptr));
        if (e != ncclSuccess) { first_err = e; break; }
        ++posted;
    }
    if (first_err != ncclSuccess) {
        // Some ranks are posted, the rest are not: nothing may be left waiting for them. The group is closed (it has to be,
        // for the thread's sake), then the communicators are aborted, which releases the kernels of the ranks that were
        // posted, and the rt_comm refuses further use.
        c->broken = true;
        (void)rccl().GroupEnd();
        if (rccl().CommAbort)
            for (ncclComm_t k : c->comms)
                if (k) (void)rccl().CommAbort(k);
        (void)hipSetDevice(current);
        return rt_set_error(RT_ERR_HIP, "ncclGather failed on rank %d of %d (%s); the communicator has been aborted", posted, n, rccl().GetErrorString(first_err));
    }
    const ncclResult_t ge = rccl().GroupEnd();
    (void)hipSetDevice(c->devices[0]);
    int rc = RT_OK;
    if (ge != ncclSuccess) { c->broken = true; rc = rt_set_error(RT_ERR_HIP, "ncclGroupEnd failed: %s", rccl().GetErrorString(ge)); }
    // frame order on the root, on the root's stream (behind its part of the gather)
    if (rc == RT_OK) rc = rt_deinterleave_device(p, d_stacked, d_frame, elem_bytes, streams ? streams[0] : nullptr);
    (void)hipSetDevice(current);
    return rc;
}

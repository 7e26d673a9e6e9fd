// jcdf_group.hpp — multi-device group of the C ABI (include/jcdf.h, "multi-device group"): all devices of one process
// behind one call, the partial Fock matrices summed ON THE DEVICES.  Included at the end of jcdf_api.hip (one
// translation unit: it drives the members through the same internals as the per-handle entry points).
//
// Reference: one Julia task per device (GPUDF.jl:188-193), D2H of every device's F and a host axpy! over the devices
// (GPUDF.jl:267-277), MPI.Allreduce! across ranks (DensityFitting.jl:68-71).  Here ONE host thread enqueues every member
// (all HIP calls of a build are asynchronous), C_occ goes up once, and F comes down once:
//
//   member 0:   H2D C ──ev_c──► build ─► [reduce slice 0] ─► D2H slice 0 ─┐
//   member i:   wait ev_c, peer copy of C ─► build ─► [reduce slice i] ─► D2H slice i ─┴─► F_out (host), N*N doubles in all
//
// [reduce slice i], transport "peer": member i waits for every member's build (events) and sums ITS slice of the N*N
// elements from all members' buffers, read through peer-mapped pointers over xGMI, in fixed member order — a
// reduce-scatter with one writer per element and no atomics, in place (member j reads slice j of member i's buffer
// while member i writes slice i: disjoint).  Transport "rccl": ncclReduceScatter (in place: recvbuff = sendbuff +
// rank * chunk) on every member's stream inside one ncclGroupStart/End, librccl.so.1 bound with dlopen so that the
// library has no link-time dependency on it.
#pragma once
#include <rccl/rccl.h>          // types and enums only: every entry point is bound with dlsym
#include <dlfcn.h>
#include <chrono>

namespace {

struct GroupSrc {
    const double *p[JCDF_GROUP_MAX_DEVICES];
};

// dst[off + e] = sum_{j < n} src.p[j][off + e] (j ascending), e < len.  dst IS src.p[self] (in place: an element is read and
// written by the same thread, so no __restrict__ on it).  off is a multiple of 256 elements: 16-byte loads.
__global__ __launch_bounds__(256) void k_group_reduce_slice(GroupSrc src, int n, double *dst, int64_t off, int64_t len)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; e < len; e += stride) {
        if (e + 1 < len) {
            double2 acc = *reinterpret_cast<const double2 *>(src.p[0] + off + e);
            for (int j = 1; j < n; ++j) {
                const double2 v = *reinterpret_cast<const double2 *>(src.p[j] + off + e);
                acc.x += v.x;
                acc.y += v.y;
            }
            *reinterpret_cast<double2 *>(dst + off + e) = acc;
        } else {
            double acc = src.p[0][off + e];
            for (int j = 1; j < n; ++j) acc += src.p[j][off + e];
            dst[off + e] = acc;
        }
    }
}

// F[p * ldf + q] = src.p[owner(e)][e], e = q + N p, owner = e / chunk: the reduced slices gathered where the
// device-resident caller wants the matrix (leading dimension ldf); n = 1 with chunk >= N*N: a plain strided copy.
__global__ __launch_bounds__(256) void k_group_gather_ld(GroupSrc src, int64_t chunk, int N, double *__restrict__ F, int64_t ldf)
{
    const int64_t count = (int64_t)N * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
        const int64_t p = e / N, q = e - p * N;
        F[p * ldf + q] = src.p[e / chunk][e];
    }
}

struct RcclApi {
    void *lib = nullptr;
    std::string where;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

// librccl.so.1 as the process already has it (PyTorch carries its own copy, and with it its own HIP runtime: a second
// RCCL bound to another libamdhip64 must not be mixed in), else by soname through the usual search path, else ROCm's.
bool rccl_load(RcclApi &api, std::string &why)
{
    if (api.lib) return true;
    struct Try { const char *name; int flags; } tries[] = {
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
        {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
        {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
    };
    void *lib = nullptr;
    for (auto &t : tries) {
        lib = dlopen(t.name, t.flags);
        if (lib) {
            api.where = std::string(t.name) + ((t.flags & RTLD_NOLOAD) ? " (already in the process)" : "");
            break;
        }
    }
    if (!lib) {
        const char *e = dlerror();
        why = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "not found");
        return false;
    }
#define JCDF_RCCL_SYM(field, sym)                                                             \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(lib, sym));                       \
    if (!api.field) { why = std::string("librccl: missing symbol ") + sym; dlclose(lib); return false; }
    JCDF_RCCL_SYM(GetVersion, "ncclGetVersion")
    JCDF_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    JCDF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    JCDF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    JCDF_RCCL_SYM(ReduceScatter, "ncclReduceScatter")
    JCDF_RCCL_SYM(Reduce, "ncclReduce")
    JCDF_RCCL_SYM(GroupStart, "ncclGroupStart")
    JCDF_RCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef JCDF_RCCL_SYM
    api.lib = lib;
    return true;
}

std::string g_group_create_error;

enum GroupTransport { GT_AUTO = 0, GT_RCCL = 1, GT_PEER = 2 };

}  // namespace

struct jcdf_group {
    int n = 0;
    std::vector<jcdf_handle *> m;          // members (owned)
    std::vector<int> dev;
    bool shared_device = false;            // two members on one device
    bool peer_ok = false;                  // every pair of distinct devices has peer access enabled (both directions)
    std::string peer_why;
    std::string err, transport_desc;
    int want = GT_AUTO, eff = GT_AUTO;     // eff is decided by group_resolve_transport
    bool resolved = false;
    bool configured = false;
    int64_t N = 0, count = 0, chunk = 0;
    std::vector<int64_t> off;              // n + 1 slice offsets
    std::vector<double *> gF;              // per member: n * chunk doubles (the build's output; slice i reduced in place)
    std::vector<double *> gT;              // per member: staging of a pushed three-centre block (freed at the first build)
    std::vector<int64_t> gT_doubles;
    double *root_red = nullptr;            // "rccl" + device entry: ncclReduce target on member 0 (N*N)
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    // events per member: 0 start of the C copy, 1 C on the device, 2 reduce start (peers' builds done), 3 reduce done, 4 D2H / gather done
    static constexpr int GEV = 5;
    std::vector<hipEvent_t> ev;            // n * GEV
    hipEvent_t ev_order = nullptr;         // device entry: caller stream <-> member 0's stream
    bool have_red_events = false;          // a previous build recorded ev[.][3]: the next build waits for the peers' reads
    bool pending = false, pending_host = false;
    double host_t0 = 0.0, host_total = 0.0;
    hipEvent_t &E(int i, int k) { return ev[(size_t)i * GEV + k]; }
};

namespace {

// A group call visits every member's device; the caller's current device (PyTorch and Julia keep their own idea of it) is put
// back on every way out.
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int32_t gfail(jcdf_group *g, int32_t code, const std::string &msg)
{
    if (g) g->err = msg;
    return code;
}

// a member's failure, reported on the group
int32_t gmember(jcdf_group *g, int i, int32_t rc, const char *what)
{
    if (rc == JCDF_OK) return rc;
    return gfail(g, rc, std::string(what) + " (member " + std::to_string(i) + ", device " + std::to_string(g->dev[(size_t)i]) + "): " + g->m[(size_t)i]->err);
}

#define JCDF_GHIP(g, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return gfail((g), JCDF_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

void group_free_buffers(jcdf_group *g)
{
    for (int i = 0; i < g->n; ++i) {
        if (!g->m[(size_t)i]) continue;
        (void)hipSetDevice(g->dev[(size_t)i]);
        if ((size_t)i < g->gF.size() && g->gF[(size_t)i]) { (void)hipFree(g->gF[(size_t)i]); g->gF[(size_t)i] = nullptr; }
        if ((size_t)i < g->gT.size() && g->gT[(size_t)i]) { (void)hipFree(g->gT[(size_t)i]); g->gT[(size_t)i] = nullptr; g->gT_doubles[(size_t)i] = 0; }
    }
    if (g->root_red) { (void)hipSetDevice(g->dev[0]); (void)hipFree(g->root_red); g->root_red = nullptr; }
    (void)hipGetLastError();
}

void group_release_staging(jcdf_group *g)
{
    for (int i = 0; i < g->n; ++i)
        if (g->gT[(size_t)i]) {
            (void)hipSetDevice(g->dev[(size_t)i]);
            (void)hipFree(g->gT[(size_t)i]);
            g->gT[(size_t)i] = nullptr;
            g->gT_doubles[(size_t)i] = 0;
        }
}

// peer access between every ordered pair of distinct member devices
void group_enable_peer_access(jcdf_group *g)
{
    g->peer_ok = true;
    for (int i = 0; i < g->n && g->peer_ok; ++i)
        for (int j = 0; j < g->n && g->peer_ok; ++j) {
            const int di = g->dev[(size_t)i], dj = g->dev[(size_t)j];
            if (di == dj) continue;
            int can = 0;
            if (hipSetDevice(di) != hipSuccess || hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) {
                g->peer_ok = false;
                g->peer_why = "device " + std::to_string(di) + " cannot map the memory of device " + std::to_string(dj);
                break;
            }
            const hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();
            } else if (e != hipSuccess) {
                g->peer_ok = false;
                g->peer_why = std::string("hipDeviceEnablePeerAccess(") + std::to_string(dj) + ") on device " + std::to_string(di) + ": " + hipGetErrorString(e);
            }
        }
}

void group_destroy_comms(jcdf_group *g)
{
    if (g->rccl.lib)
        for (auto c : g->comms)
            if (c) (void)g->rccl.CommDestroy(c);
    g->comms.clear();
}

int32_t group_init_rccl(jcdf_group *g)
{
    if (g->shared_device) return gfail(g, JCDF_ERR_INVALID, "transport rccl: two members share a device (RCCL needs distinct devices; use \"peer\")");
    std::string why;
    if (!rccl_load(g->rccl, why)) return gfail(g, JCDF_ERR_HIP, "transport rccl: " + why);
    if (g->comms.empty()) {
        g->comms.assign((size_t)g->n, nullptr);
        const ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), g->n, g->dev.data());
        if (r != ncclSuccess) {
            g->comms.clear();
            return gfail(g, JCDF_ERR_HIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r));
        }
    }
    int v = 0;
    (void)g->rccl.GetVersion(&v);
    char buf[160];
    // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch (>= 2.9)
    std::snprintf(buf, sizeof(buf), "rccl %d.%d.%d (%s, %d ranks, ncclReduceScatter in place)", v / 10000, (v / 100) % 100, v % 100,
                  g->rccl.where.c_str(), g->n);
    g->transport_desc = buf;
    g->eff = GT_RCCL;
    return JCDF_OK;
}

int32_t group_init_peer(jcdf_group *g)
{
    if (!g->peer_ok) return gfail(g, JCDF_ERR_HIP, "transport peer: " + g->peer_why);
    g->transport_desc = "peer (" + std::to_string(g->n) + " members" + (g->shared_device ? ", shared devices" : "") +
                        ", fixed-order slice sums over peer-mapped buffers)";
    g->eff = GT_PEER;
    return JCDF_OK;
}

int32_t group_resolve_transport(jcdf_group *g)
{
    if (g->resolved) return JCDF_OK;
    int32_t rc;
    if (g->want == GT_RCCL) {
        rc = group_init_rccl(g);
    } else if (g->want == GT_PEER) {
        rc = group_init_peer(g);
    } else {
        // auto: RCCL for n > 1 distinct devices when it can be set up, else the peer kernel — both reduce on the devices
        rc = JCDF_ERR_INVALID;
        std::string rccl_err;
        if (g->n > 1 && !g->shared_device) {
            rc = group_init_rccl(g);
            if (rc) rccl_err = g->err;
        }
        if (rc) {
            rc = group_init_peer(g);
            if (rc && !rccl_err.empty()) g->err += "; " + rccl_err;
            if (!rc && !rccl_err.empty()) g->transport_desc += " [auto: " + rccl_err + "]";
        }
    }
    if (rc) return rc;
    g->resolved = true;
    return JCDF_OK;
}

// every member's stream waits for what a previous group build still reads from its buffers
void group_wait_previous_readers(jcdf_group *g, HipAcc &ok)
{
    if (!g->have_red_events || g->n == 1) return;
    for (int i = 0; i < g->n; ++i) {
        ok(hipSetDevice(g->dev[(size_t)i]));
        for (int j = 0; j < g->n; ++j)
            if (j != i) ok(hipStreamWaitEvent(g->m[(size_t)i]->stream, g->E(j, 4), 0));
    }
}

// Members 1.. take C from member 0's packed copy (m[0]->dC, ready at event E(0,1)), all members build into gF.
int32_t group_enqueue_builds(jcdf_group *g, const double *dC0, int64_t ldc0)
{
    const size_t cbytes = (size_t)(g->N * g->m[0]->o) * 8;
    HipAcc ok;
    for (int i = 1; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        ok(hipSetDevice(h->device));
        ok(hipStreamWaitEvent(h->stream, g->E(0, 1), 0));
        ok(hipEventRecord(g->E(i, 0), h->stream));
        if (h->device == g->dev[0])
            ok(hipMemcpyAsync(h->dC, g->m[0]->dC, cbytes, hipMemcpyDeviceToDevice, h->stream));
        else
            ok(hipMemcpyPeerAsync(h->dC, h->device, g->m[0]->dC, g->dev[0], cbytes, h->stream));
        ok(hipEventRecord(g->E(i, 1), h->stream));
    }
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("group build, C broadcast: ") + hipGetErrorString(ok.first));
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        JCDF_GHIP(g, hipSetDevice(h->device));
        if (h->stage_doubles) release_stage(h);
        h->timed_host_copy = false;
        const int32_t rc = (i == 0) ? enqueue_fock(h, dC0, ldc0, g->gF[0], g->N, h->stream)
                                    : enqueue_fock(h, h->dC, g->N, g->gF[(size_t)i], g->N, h->stream);
        if (rc) return gmember(g, i, rc, "Fock build");
    }
    return JCDF_OK;
}

// reduce-scatter of the members' gF: afterwards slice i of the sum is at gF[i] + off[i]; E(i,2) / E(i,3) bracket it
int32_t group_enqueue_reduce_scatter(jcdf_group *g)
{
    HipAcc ok;
    if (g->eff == GT_RCCL) {
        for (int i = 0; i < g->n; ++i) {
            ok(hipSetDevice(g->dev[(size_t)i]));
            ok(hipEventRecord(g->E(i, 2), g->m[(size_t)i]->stream));
        }
        ncclResult_t r = g->rccl.GroupStart();
        for (int i = 0; i < g->n && r == ncclSuccess; ++i)
            r = g->rccl.ReduceScatter(g->gF[(size_t)i], g->gF[(size_t)i] + (int64_t)i * g->chunk, (size_t)g->chunk, ncclDouble, ncclSum,
                                      g->comms[(size_t)i], g->m[(size_t)i]->stream);
        const ncclResult_t r2 = g->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return gfail(g, JCDF_ERR_HIP, std::string("ncclReduceScatter: ") + g->rccl.GetErrorString(r));
        for (int i = 0; i < g->n; ++i) {
            ok(hipSetDevice(g->dev[(size_t)i]));
            ok(hipEventRecord(g->E(i, 3), g->m[(size_t)i]->stream));
        }
    } else {
        GroupSrc src{};
        for (int j = 0; j < g->n; ++j) src.p[j] = g->gF[(size_t)j];
        for (int i = 0; i < g->n; ++i) {
            jcdf_handle *h = g->m[(size_t)i];
            ok(hipSetDevice(h->device));
            for (int j = 0; j < g->n; ++j)
                if (j != i) ok(hipStreamWaitEvent(h->stream, g->m[(size_t)j]->ev_end, 0));     // member j's build is complete
            ok(hipEventRecord(g->E(i, 2), h->stream));
            const int64_t len = g->off[(size_t)i + 1] - g->off[(size_t)i];
            if (len > 0 && g->n > 1) {
                const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((len / 2 + 255) / 256, 4 * (int64_t)h->num_cu));
                hipLaunchKernelGGL(k_group_reduce_slice, dim3(grid), dim3(256), 0, h->stream, src, g->n, g->gF[(size_t)i], g->off[(size_t)i], len);
            }
            ok(hipEventRecord(g->E(i, 3), h->stream));
        }
        ok(hipGetLastError());
    }
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("group build, reduce: ") + hipGetErrorString(ok.first));
    return JCDF_OK;
}

int32_t group_check_ready(jcdf_group *g, const char *who)
{
    if (!g) return JCDF_ERR_INVALID;
    if (!g->configured) return gfail(g, JCDF_ERR_INVALID, std::string(who) + ": jcdf_group_configure first");
    for (int i = 0; i < g->n; ++i)
        if (!g->m[(size_t)i]->have_B) return gfail(g, JCDF_ERR_INVALID, std::string(who) + ": B not set on member " + std::to_string(i));
    return group_resolve_transport(g);
}

void group_fill_timings(jcdf_group *g, jcdf_group_timings *gt, bool host_entry)
{
    if (!gt) return;
    std::memset(gt, 0, sizeof(*gt));
    double peer_copy = 0.0;
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        (void)hipSetDevice(h->device);
        if (i > 0) peer_copy = std::max(peer_copy, elapsed_s(g->E(i, 0), g->E(i, 1)));
        gt->build_time = std::max(gt->build_time, elapsed_s(h->ev_begin, h->ev_end));
        gt->reduce_time = std::max(gt->reduce_time, elapsed_s(g->E(i, 2), g->E(i, 3)));
        if (host_entry || i == 0) gt->d2h_time = std::max(gt->d2h_time, elapsed_s(g->E(i, 3), g->E(i, 4)));
    }
    (void)hipSetDevice(g->dev[0]);
    gt->bcast_time = elapsed_s(g->E(0, 0), g->E(0, 1)) + peer_copy;      // H2D (device entry: the packed copy) + the longest peer copy
    gt->total_time = g->host_total;
}

}  // namespace

extern "C" {

int64_t jcdf_group_reduce_plan(int64_t count, int32_t n, int64_t *offsets)
{
    if (count <= 0 || n < 1 || n > JCDF_GROUP_MAX_DEVICES || !offsets) return -1;
    const int64_t chunk = roundup((count + n - 1) / n, 256);
    for (int i = 0; i <= n; ++i) offsets[i] = std::min<int64_t>((int64_t)i * chunk, count);
    return chunk;
}

const char *jcdf_group_last_error(const jcdf_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int32_t jcdf_group_size(const jcdf_group *g) { return g ? g->n : 0; }

jcdf_handle *jcdf_group_handle(jcdf_group *g, int32_t i) { return (g && i >= 0 && i < g->n) ? g->m[(size_t)i] : nullptr; }

const char *jcdf_group_transport(const jcdf_group *g)
{
    if (!g) return "";
    return g->resolved ? g->transport_desc.c_str() : (g->want == GT_RCCL ? "rccl (not initialised yet)" : g->want == GT_PEER ? "peer (not initialised yet)" : "auto (not resolved yet)");
}

int32_t jcdf_group_destroy(jcdf_group *g)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_OK;
    for (auto h : g->m)
        if (h) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
    group_destroy_comms(g);
    group_free_buffers(g);
    for (int i = 0; i < g->n; ++i) {
        if (!g->m[(size_t)i]) continue;                  // a member that was never created (jcdf_group_create failed there): its device id may not exist
        (void)hipSetDevice(g->dev[(size_t)i]);
        for (int k = 0; k < jcdf_group::GEV; ++k)
            if ((size_t)(i * jcdf_group::GEV + k) < g->ev.size() && g->E(i, k)) (void)hipEventDestroy(g->E(i, k));
    }
    if (g->ev_order) { (void)hipSetDevice(g->dev[0]); (void)hipEventDestroy(g->ev_order); }
    (void)hipGetLastError();                             // nothing of a teardown stays behind as the thread's "last error"
    for (auto h : g->m)
        if (h) (void)jcdf_destroy(h);
    delete g;
    return JCDF_OK;
}

int32_t jcdf_group_create(jcdf_group **out, int32_t n_devices, const int32_t *device_ids)
{
    DeviceRestore restore_device_;
    if (!out) { g_group_create_error = "jcdf_group_create: out == NULL"; return JCDF_ERR_INVALID; }
    *out = nullptr;
    if (n_devices < 1 || n_devices > JCDF_GROUP_MAX_DEVICES || !device_ids) {
        g_group_create_error = "jcdf_group_create: 1 .. " + std::to_string(JCDF_GROUP_MAX_DEVICES) + " devices and a device list expected";
        return JCDF_ERR_INVALID;
    }
    jcdf_group *g = new (std::nothrow) jcdf_group();
    if (!g) { g_group_create_error = "jcdf_group_create: out of host memory"; return JCDF_ERR_ALLOC; }
    g->n = n_devices;
    g->dev.assign(device_ids, device_ids + n_devices);
    g->m.assign((size_t)n_devices, nullptr);
    g->gF.assign((size_t)n_devices, nullptr);
    g->gT.assign((size_t)n_devices, nullptr);
    g->gT_doubles.assign((size_t)n_devices, 0);
    g->ev.assign((size_t)n_devices * jcdf_group::GEV, nullptr);
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) g->shared_device = true;
    for (int i = 0; i < n_devices; ++i) {
        const int32_t rc = jcdf_create(&g->m[(size_t)i], device_ids[i]);
        if (rc) {
            g_group_create_error = "jcdf_group_create, member " + std::to_string(i) + ": " + g_create_error;
            jcdf_group_destroy(g);
            return rc;
        }
        hipError_t e = hipSuccess;
        for (int k = 0; k < jcdf_group::GEV && e == hipSuccess; ++k) e = hipEventCreate(&g->E(i, k));
        if (e != hipSuccess) {
            g_group_create_error = std::string("jcdf_group_create: hipEventCreate: ") + hipGetErrorString(e);
            jcdf_group_destroy(g);
            return JCDF_ERR_HIP;
        }
    }
    (void)hipSetDevice(g->dev[0]);
    if (hipEventCreateWithFlags(&g->ev_order, hipEventDisableTiming) != hipSuccess) {
        g_group_create_error = "jcdf_group_create: hipEventCreate failed";
        jcdf_group_destroy(g);
        return JCDF_ERR_HIP;
    }
    group_enable_peer_access(g);          // a failure is reported when a transport that needs it is selected
    *out = g;
    return JCDF_OK;
}

int32_t jcdf_group_set_transport(jcdf_group *g, const char *name)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    if (!name) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_set_transport: NULL name");
    const std::string s(name);
    int want;
    if (s == "auto") want = GT_AUTO;
    else if (s == "rccl") want = GT_RCCL;
    else if (s == "peer") want = GT_PEER;
    else return gfail(g, JCDF_ERR_INVALID, "jcdf_group_set_transport: unknown transport '" + s + "' (auto | rccl | peer)");
    for (auto h : g->m) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
    g->want = want;
    g->resolved = false;
    return group_resolve_transport(g);
}

int32_t jcdf_group_set_exchange_screening(jcdf_group *g, int64_t n_blocks)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    for (int i = 0; i < g->n; ++i) {
        const int32_t rc = gmember(g, i, jcdf_set_exchange_screening(g->m[(size_t)i], n_blocks), "jcdf_set_exchange_screening");
        if (rc) return rc;
    }
    return JCDF_OK;
}

int32_t jcdf_group_configure(jcdf_group *g, int64_t N, int64_t Q_total, const int64_t *shard_q0, int64_t n_occ, int64_t P,
                             const int64_t *pq_p, const int64_t *pq_q)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    if (!shard_q0 || N <= 0) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_configure: NULL shard list / N <= 0");
    if (shard_q0[0] < 0 || shard_q0[g->n] > Q_total) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_configure: shard list outside [0, Q_total]");
    for (int i = 0; i < g->n; ++i)
        if (shard_q0[i + 1] <= shard_q0[i])
            return gfail(g, JCDF_ERR_INVALID, "jcdf_group_configure: empty or descending aux shard for member " + std::to_string(i) +
                                                  " (more devices than auxiliary shells?)");
    for (auto h : g->m) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
    g->configured = false;
    g->have_red_events = false;
    group_free_buffers(g);
    for (int i = 0; i < g->n; ++i) {
        const int32_t rc = gmember(g, i, jcdf_configure(g->m[(size_t)i], N, Q_total, shard_q0[i], shard_q0[i + 1], n_occ, P, pq_p, pq_q), "jcdf_configure");
        if (rc) return rc;
    }
    g->N = N;
    g->count = N * N;
    g->off.assign((size_t)g->n + 1, 0);
    g->chunk = jcdf_group_reduce_plan(g->count, g->n, g->off.data());
    for (int i = 0; i < g->n; ++i) {
        JCDF_GHIP(g, hipSetDevice(g->dev[(size_t)i]));
        const size_t bytes = (size_t)(g->chunk * g->n) * 8;
        if (hipMalloc((void **)&g->gF[(size_t)i], bytes) != hipSuccess) return gfail(g, JCDF_ERR_ALLOC, "jcdf_group_configure: out of device memory for the Fock buffers");
        JCDF_GHIP(g, hipMemset(g->gF[(size_t)i], 0, bytes));       // the tail beyond N*N takes part in the RCCL reduce-scatter: zeros
    }
    g->configured = true;
    return group_resolve_transport(g);     // decided (and reported by jcdf_group_transport) before the first build
}

int32_t jcdf_group_set_metric(jcdf_group *g, const double *J2c)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    if (!g->configured || !J2c) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_set_metric: configure first / NULL");
    jcdf_handle *h0 = g->m[0];
    if (h0->tune_host_cholesky) {                     // the host routine was asked for: once per member, as the per-handle call does it
        for (int i = 0; i < g->n; ++i) {
            g->m[(size_t)i]->tune_host_cholesky = 1;
            const int32_t rc = gmember(g, i, jcdf_set_metric(g->m[(size_t)i], J2c), "jcdf_set_metric");
            if (rc) return rc;
        }
        return JCDF_OK;
    }
    JCDF_GHIP(g, hipSetDevice(h0->device));
    CholBuffers w;                                    // potrf + trtri once, on member 0's device
    int info = 0;
    const hipError_t e = chol_inverse_device(h0->stream, J2c, h0->Qtot, w, &info);
    if (e == hipErrorOutOfMemory) return gfail(g, JCDF_ERR_ALLOC, "jcdf_group_set_metric: out of device memory for the factorisation");
    if (e != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_set_metric: ") + hipGetErrorString(e));
    if (info != 0) return gfail(g, JCDF_ERR_NOT_SPD, "jcdf_group_set_metric: (P|Q) not positive definite at pivot " + std::to_string(info));
    const size_t vbytes = (size_t)(w.ld * w.ld) * 8;
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        JCDF_GHIP(g, hipSetDevice(h->device));
        if (h->device == h0->device) {
            const int32_t rc = gmember(g, i, upload_linv_from_device(h, w.V, w.ld), "L^-1 rows");
            if (rc) return rc;
            continue;
        }
        double *Vi = nullptr;                         // V = L^-T travels device-to-device; member i keeps its rows of L^-1
        if (hipMalloc((void **)&Vi, vbytes) != hipSuccess) return gfail(g, JCDF_ERR_ALLOC, "jcdf_group_set_metric: out of device memory for the copy of L^-1");
        hipError_t ce = hipMemcpyPeer(Vi, h->device, w.V, h0->device, vbytes);
        int32_t rc = JCDF_OK;
        if (ce != hipSuccess) rc = gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_set_metric: hipMemcpyPeer: ") + hipGetErrorString(ce));
        if (!rc) rc = gmember(g, i, upload_linv_from_device(h, Vi, w.ld), "L^-1 rows");
        (void)hipFree(Vi);
        if (rc) return rc;
    }
    return JCDF_OK;
}

int32_t jcdf_group_push_three_center(jcdf_group *g, int64_t s0, int64_t s1, const double *T)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    if (!g->configured) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_push_three_center: configure first");
    jcdf_handle *h0 = g->m[0];
    if (!T || s0 < 0 || s1 <= s0 || s1 > h0->Qtot) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_push_three_center: bad row range / NULL");
    const int64_t doubles = (s1 - s0) * h0->P;
    int first = -1;                                   // the member that received the block from the host
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        if (!h->have_metric) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_push_three_center: set the metric first");
        JCDF_GHIP(g, hipSetDevice(h->device));
        if (!h->pushed_any) {
            JCDF_GHIP(g, hipMemsetAsync(h->dB, 0, (size_t)((h->P + KC) * h->ldq + 4 * TILE_Q) * 8, h->stream));
            h->pushed_any = true;
        }
        if (s0 >= h->q1) continue;                    // L^-1[q0:q1, s0:s1] == 0
        if (g->gT_doubles[(size_t)i] < doubles) {
            if (g->gT[(size_t)i]) (void)hipFree(g->gT[(size_t)i]);
            g->gT[(size_t)i] = nullptr;
            g->gT_doubles[(size_t)i] = 0;
            if (hipMalloc((void **)&g->gT[(size_t)i], (size_t)doubles * 8) != hipSuccess)
                return gfail(g, JCDF_ERR_ALLOC, "jcdf_group_push_three_center: out of device memory for the staged block");
            g->gT_doubles[(size_t)i] = doubles;
        }
        if (first < 0) {
            JCDF_GHIP(g, hipMemcpyAsync(g->gT[(size_t)i], T, (size_t)doubles * 8, hipMemcpyDefault, h->stream));    // T: host memory, or device memory of any device of the process
            JCDF_GHIP(g, hipStreamSynchronize(h->stream));         // the caller's buffer is free again; the block is on the device
            first = i;
        } else if (h->device == g->dev[(size_t)first]) {
            JCDF_GHIP(g, hipMemcpyAsync(g->gT[(size_t)i], g->gT[(size_t)first], (size_t)doubles * 8, hipMemcpyDeviceToDevice, h->stream));
        } else {
            JCDF_GHIP(g, hipMemcpyPeerAsync(g->gT[(size_t)i], h->device, g->gT[(size_t)first], g->dev[(size_t)first], (size_t)doubles * 8, h->stream));
        }
        const int32_t rc = gmember(g, i, push_block(h, s0, s1, g->gT[(size_t)i], true), "push_three_center");
        if (rc) return rc;
        h->have_B = true;
    }
    for (int i = 0; i < g->n; ++i) {                  // the staging buffers are reused by the next push
        JCDF_GHIP(g, hipSetDevice(g->dev[(size_t)i]));
        JCDF_GHIP(g, hipStreamSynchronize(g->m[(size_t)i]->stream));
    }
    return JCDF_OK;
}

int32_t jcdf_group_set_core_hamiltonian(jcdf_group *g, const double *H)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    if (!g->configured) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_set_core_hamiltonian: configure first");
    for (int i = 0; i < g->n; ++i) {
        const int32_t rc = gmember(g, i, jcdf_set_core_hamiltonian(g->m[(size_t)i], i == 0 ? H : nullptr), "jcdf_set_core_hamiltonian");
        if (rc) return rc;
    }
    return JCDF_OK;
}

int32_t jcdf_group_synchronize(jcdf_group *g, jcdf_timings *t, jcdf_group_timings *gt)
{
    DeviceRestore restore_device_;
    if (!g) return JCDF_ERR_INVALID;
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        JCDF_GHIP(g, hipSetDevice(h->device));
        JCDF_GHIP(g, hipStreamSynchronize(h->stream));
    }
    if (g->pending) {
        g->host_total = now_s() - g->host_t0;
        g->pending = false;
    }
    for (int i = 0; i < g->n; ++i) {
        const int32_t rc = gmember(g, i, jcdf_synchronize(g->m[(size_t)i], t ? &t[i] : nullptr), "jcdf_synchronize");
        if (rc) return rc;
    }
    if (g->configured && g->have_red_events) group_fill_timings(g, gt, g->pending_host);
    else if (gt) std::memset(gt, 0, sizeof(*gt));
    return JCDF_OK;
}

int32_t jcdf_group_fock_build(jcdf_group *g, const double *C_occ, double *F_out, jcdf_timings *t, jcdf_group_timings *gt)
{
    DeviceRestore restore_device_;
    int32_t rc = group_check_ready(g, "jcdf_group_fock_build");
    if (rc) return rc;
    if (!C_occ || !F_out) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_fock_build: NULL pointer");
    group_release_staging(g);
    g->host_t0 = now_s();
    g->pending = true;
    g->pending_host = true;
    jcdf_handle *h0 = g->m[0];
    HipAcc ok;
    group_wait_previous_readers(g, ok);
    // C_occ: ONE H2D (pageable host memory: the copy has left the caller's buffer when the call returns)
    ok(hipSetDevice(h0->device));
    ok(hipEventRecord(g->E(0, 0), h0->stream));
    ok(hipMemcpyAsync(h0->dC, C_occ, (size_t)(g->N * h0->o) * 8, hipMemcpyHostToDevice, h0->stream));
    ok(hipEventRecord(g->E(0, 1), h0->stream));
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_fock_build: ") + hipGetErrorString(ok.first));
    if ((rc = group_enqueue_builds(g, h0->dC, g->N))) return rc;
    if ((rc = group_enqueue_reduce_scatter(g))) return rc;
    // the reduced slices leave their devices: N*N doubles in total, slice i over device i's own link
    for (int i = 0; i < g->n; ++i) {
        jcdf_handle *h = g->m[(size_t)i];
        const int64_t len = g->off[(size_t)i + 1] - g->off[(size_t)i];
        ok(hipSetDevice(h->device));
        if (len > 0)
            ok(hipMemcpyAsync(F_out + g->off[(size_t)i], g->gF[(size_t)i] + g->off[(size_t)i], (size_t)len * 8, hipMemcpyDeviceToHost, h->stream));
        ok(hipEventRecord(g->E(i, 4), h->stream));
    }
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_fock_build, D2H: ") + hipGetErrorString(ok.first));
    g->have_red_events = true;
    return jcdf_group_synchronize(g, t, gt);
}

int32_t jcdf_group_fock_build_device_ld(jcdf_group *g, const double *d_C_occ, int64_t ldc, double *d_F, int64_t ldf, void *stream)
{
    DeviceRestore restore_device_;
    int32_t rc = group_check_ready(g, "jcdf_group_fock_build_device_ld");
    if (rc) return rc;
    if (!d_C_occ || !d_F || ldc < g->N || ldf < g->N) return gfail(g, JCDF_ERR_INVALID, "jcdf_group_fock_build_device_ld: NULL pointer / leading dimension < N");
    group_release_staging(g);
    g->host_t0 = now_s();
    g->pending = true;
    g->pending_host = false;
    jcdf_handle *h0 = g->m[0];
    hipStream_t user = stream ? (hipStream_t)stream : h0->stream;
    HipAcc ok;
    group_wait_previous_readers(g, ok);
    ok(hipSetDevice(h0->device));
    if (user != h0->stream) {                          // member 0's stream runs behind the caller's
        ok(hipEventRecord(g->ev_order, user));
        ok(hipStreamWaitEvent(h0->stream, g->ev_order, 0));
    }
    // a packed copy of C on member 0 for the other members to fetch (member 0 itself reads the caller's matrix as it is)
    ok(hipEventRecord(g->E(0, 0), h0->stream));
    if (g->n > 1)
        ok(hipMemcpy2DAsync(h0->dC, (size_t)g->N * 8, d_C_occ, (size_t)ldc * 8, (size_t)g->N * 8, (size_t)h0->o, hipMemcpyDeviceToDevice, h0->stream));
    ok(hipEventRecord(g->E(0, 1), h0->stream));
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_fock_build_device_ld: ") + hipGetErrorString(ok.first));
    if ((rc = group_enqueue_builds(g, d_C_occ, ldc))) return rc;
    const unsigned ggrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((g->count + 255) / 256, 8 * (int64_t)h0->num_cu));
    if (g->eff == GT_RCCL && g->n > 1) {
        // the whole sum on member 0 (ncclReduce), then the strided copy into the caller's matrix
        if (!g->root_red) {
            ok(hipSetDevice(h0->device));
            if (hipMalloc((void **)&g->root_red, (size_t)g->count * 8) != hipSuccess) return gfail(g, JCDF_ERR_ALLOC, "jcdf_group_fock_build_device_ld: out of device memory");
        }
        for (int i = 0; i < g->n; ++i) {
            ok(hipSetDevice(g->dev[(size_t)i]));
            ok(hipEventRecord(g->E(i, 2), g->m[(size_t)i]->stream));
        }
        ncclResult_t r = g->rccl.GroupStart();
        for (int i = 0; i < g->n && r == ncclSuccess; ++i)
            r = g->rccl.Reduce(g->gF[(size_t)i], i == 0 ? g->root_red : nullptr, (size_t)g->count, ncclDouble, ncclSum, 0, g->comms[(size_t)i],
                               g->m[(size_t)i]->stream);
        const ncclResult_t r2 = g->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return gfail(g, JCDF_ERR_HIP, std::string("ncclReduce: ") + g->rccl.GetErrorString(r));
        for (int i = 0; i < g->n; ++i) {
            ok(hipSetDevice(g->dev[(size_t)i]));
            ok(hipEventRecord(g->E(i, 3), g->m[(size_t)i]->stream));
            if (i > 0) ok(hipEventRecord(g->E(i, 4), g->m[(size_t)i]->stream));
        }
        ok(hipSetDevice(h0->device));
        GroupSrc one{};
        one.p[0] = g->root_red;
        hipLaunchKernelGGL(k_group_gather_ld, dim3(ggrid), dim3(256), 0, h0->stream, one, g->count, (int)g->N, d_F, ldf);
    } else {
        if ((rc = group_enqueue_reduce_scatter(g))) return rc;
        // member 0 gathers the reduced slices through the peer mappings, straight into the caller's matrix
        ok(hipSetDevice(h0->device));
        for (int j = 1; j < g->n; ++j) ok(hipStreamWaitEvent(h0->stream, g->E(j, 3), 0));
        GroupSrc src{};
        for (int j = 0; j < g->n; ++j) src.p[j] = g->gF[(size_t)j];
        hipLaunchKernelGGL(k_group_gather_ld, dim3(ggrid), dim3(256), 0, h0->stream, src, g->chunk, (int)g->N, d_F, ldf);
    }
    ok(hipGetLastError());
    ok(hipEventRecord(g->E(0, 4), h0->stream));
    // the next build may overwrite gF[j] only after member 0 has read slice j: E(j,4) marks "member j's buffer is free"
    for (int j = 1; j < g->n; ++j) {
        ok(hipSetDevice(g->dev[(size_t)j]));
        ok(hipStreamWaitEvent(g->m[(size_t)j]->stream, g->E(0, 4), 0));
        ok(hipEventRecord(g->E(j, 4), g->m[(size_t)j]->stream));
    }
    ok(hipSetDevice(h0->device));
    if (user != h0->stream) {                          // the caller's stream continues behind the gather
        ok(hipEventRecord(g->ev_order, h0->stream));
        ok(hipStreamWaitEvent(user, g->ev_order, 0));
    }
    if (ok.first != hipSuccess) return gfail(g, JCDF_ERR_HIP, std::string("jcdf_group_fock_build_device_ld: ") + hipGetErrorString(ok.first));
    g->have_red_events = true;
    return JCDF_OK;
}

}  // extern "C"

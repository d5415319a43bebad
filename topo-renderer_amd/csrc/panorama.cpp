// panorama.cpp -- the 360-degree strip and the viewpoint batch behind the C ABI (include/topo_hip.h: topo_comm_*,
// topo_render_panorama, topo_render_batch), so that a host in any language -- the reference's is Rust -- reaches the
// multi-GPU path without Python: one process per GPU, each with its own topo_ctx over the replicated DEM; rank g renders
// sectors [8g/N, 8(g+1)/N) in ONE submission straight into its slice of the sector-major strip and an in-place RCCL
// all-gather on the same stream assembles the strip on every rank.  No reference counterpart: the
// reference renders one perspective view on one GPU.
//
// RCCL is bound at run time (dlopen), not at link time: a process that never creates a communicator does not need the
// library, and a host that already carries one (PyTorch bundles its own librccl.so) gets that same copy.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "terrain_renderer.hpp"

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>      // prototypes and enum values only: the library is still bound at run time
#define TOPO_HAVE_RCCL_H 1
#endif

namespace topo {

#define TOPO_HIP_TRY(expr)                                   \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);    \
    } while (0)

namespace {

struct RcclApi {
#ifdef TOPO_HAVE_RCCL_H
    // the pointers have the header's own types, so a prototype that drifts from rccl.h fails to compile here
    using UniqueId = ncclUniqueId;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    using Comm_t = ncclComm_t;
    using Result = ncclResult_t;
    static constexpr ncclDataType_t kChar = ncclChar, kFloat = ncclFloat;
#else
    // rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260, ncclGetErrorString :339, ncclCommCuDevice :389,
    // ncclAllGather :678, ncclSend :700, ncclRecv :722; ncclDataType_t :459-466
    struct UniqueId { char internal[128]; };
    using Comm_t = void*;
    using Result = int;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm_t* comm, int nranks, UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(Comm_t comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, Comm_t comm, hipStream_t stream) = nullptr;
    int (*Send)(const void* send, size_t count, int dtype, int peer, Comm_t comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* recv, size_t count, int dtype, int peer, Comm_t comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCuDevice)(const Comm_t comm, int* device) = nullptr;
    static constexpr int kChar = 0, kFloat = 7;
#endif
    void* handle = nullptr;
    std::string error;
};
static_assert(sizeof(RcclApi::UniqueId) == TOPO_COMM_ID_BYTES, "topo_comm_unique_id hands out an ncclUniqueId");

void rccl_load(RcclApi& api) {
    // an already loaded copy first (RTLD_NOLOAD), then the usual names; TOPO_RCCL_LIB overrides them (tests: a missing library)
    std::vector<std::string> names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (const char* e = getenv("TOPO_RCCL_LIB")) names = {e};
    for (const auto& n : names)
        if ((api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD))) break;
    const char* why = nullptr;
    if (!api.handle)
        for (const auto& n : names) {
            if ((api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
            why = dlerror();      // (one call: it returns the message AND clears it)
        }
    if (!api.handle) {
        api.error = std::string("librccl.so not found: ") + (why ? why : "no loader message");
        return;
    }
    auto sym = [&](const char* s) {
        void* p = dlsym(api.handle, s);
        if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + s;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.CommCuDevice = (decltype(api.CommCuDevice))sym("ncclCommCuDevice");
}

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(api); });
    return api;
}

}  // namespace

struct Comm {
    RcclApi::Comm_t nccl = nullptr;
    int rank = 0, world = 1;
    int device = -1;           // the communicator's device (ncclCommCuDevice), -1 = unknown
    bool owned = false;        // created by topo_comm_init (destroyed with the Comm) or borrowed (topo_comm_from_nccl)
    // the exchange runs on a stream of its own, behind the part of the frame it ships and beside the rest of it
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ready;      // one per exchange slot of a panorama: "this part of the strip is final"
    hipEvent_t done = nullptr;
};

int comm_unique_id(uint8_t out[128], std::string* err) {
    RcclApi& a = rccl();
    if (!a.error.empty() || !a.GetUniqueId) { *err = a.error.empty() ? "RCCL unavailable" : a.error; return TOPO_ERR_HIP; }
    RcclApi::UniqueId id;
    if (auto rc = a.GetUniqueId(&id)) { *err = std::string("ncclGetUniqueId: ") + a.GetErrorString(rc); return TOPO_ERR_HIP; }
    memcpy(out, &id, 128);
    return TOPO_OK;
}

static int comm_streams(Comm* c, int device, std::string* err) {
    if (c->world <= 1) return TOPO_OK;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        *err = "HIP stream/event creation for the exchange failed";
        return TOPO_ERR_HIP;
    }
    RcclApi& a = rccl();
    if (a.CommCuDevice && c->nccl) {
        int d = -1;
        if (a.CommCuDevice(c->nccl, &d) == 0) c->device = d;
    }
    return TOPO_OK;
}

int comm_init(Comm** out, int device, const uint8_t id128[128], int rank, int world, std::string* err) {
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { *err = "rank/world out of range"; return TOPO_ERR_INVALID; }
    if (kPanoramaSectors % world != 0) { *err = "the sector count (8) must be divisible by the number of ranks"; return TOPO_ERR_INVALID; }
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    if (world > 1) {
        RcclApi& a = rccl();
        if (!a.error.empty()) { *err = a.error; delete c; return TOPO_ERR_HIP; }
        if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; delete c; return TOPO_ERR_HIP; }
        RcclApi::UniqueId id;
        memcpy(&id, id128, 128);
        if (auto rc = a.CommInitRank(&c->nccl, world, id, rank)) { *err = std::string("ncclCommInitRank: ") + a.GetErrorString(rc); delete c; return TOPO_ERR_HIP; }
        c->owned = true;
        if (int rc = comm_streams(c, device, err)) { comm_destroy(c); return rc; }
    }
    *out = c;
    return TOPO_OK;
}

int comm_from_nccl(Comm** out, void* nccl_comm, int rank, int world, std::string* err) {
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !nccl_comm)) { *err = "bad communicator arguments"; return TOPO_ERR_INVALID; }
    if (kPanoramaSectors % world != 0) { *err = "the sector count (8) must be divisible by the number of ranks"; return TOPO_ERR_INVALID; }
    if (world > 1 && !rccl().error.empty()) { *err = rccl().error; return TOPO_ERR_HIP; }
    Comm* c = new Comm();
    c->nccl = (RcclApi::Comm_t)nccl_comm;
    c->rank = rank;
    c->world = world;
    if (world > 1) {
        int device = 0;
        RcclApi& a = rccl();
        if (!a.CommCuDevice || a.CommCuDevice(c->nccl, &device) != 0) (void)hipGetDevice(&device);
        if (int rc = comm_streams(c, device, err)) { comm_destroy(c); return rc; }
    }
    *out = c;
    return TOPO_OK;
}

void comm_destroy(Comm* c) {
    if (!c) return;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (hipEvent_t e : c->ready) (void)hipEventDestroy(e);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->owned && c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
    delete c;
}

void comm_rank(const Comm* c, int* rank, int* world) {
    *rank = c ? c->rank : 0;
    *world = c ? c->world : 1;
}

// The sectors this rank renders: [first, first + count) of the kPanoramaSectors fixed sectors.
void panorama_sector_range(int rank, int world, uint32_t* first, uint32_t* count) {
    *count = kPanoramaSectors / (uint32_t)world;
    *first = (uint32_t)rank * *count;
}

// The exchange plan of one panorama (the same on every rank): a rank's share of the strip is resolved and shipped in SLOTS
// -- (sector of the rank's range, band of rows) -- so that the exchange of slot i runs under the resolve of slot i + 1.
// Bands are whole rows of k_resolve's blocks and about 8 MiB of RGBA (xGMI moves 8 MiB in ~55 us per link; a slot per
// 1 MiB would be launch-bound, one slot per sector leaves nothing to overlap at N = 8 where a rank has ONE sector).
// World of one: one slot per sector, nothing to ship.
uint32_t panorama_slots(int world, uint32_t sector_w, uint32_t sector_h, topo_panorama_slot* out, uint32_t cap) {
    const uint32_t count = kPanoramaSectors / (uint32_t)(world < 1 ? 1 : world);
    const uint32_t block_rows = (sector_h + kResolveBlockH - 1) / kResolveBlockH;
    const size_t sector_bytes = (size_t)sector_w * sector_h * 4;
    size_t band_bytes = 8u << 20;
    if (const char* e = getenv("TOPO_PANORAMA_BAND_BYTES"))      // tuning knob (and the CPU test's way to get bands out of tiny sectors)
        if (atoll(e) > 0) band_bytes = (size_t)atoll(e);
    uint32_t bands = world > 1 ? (uint32_t)((sector_bytes + band_bytes - 1) / band_bytes) : 1u;
    if (bands > 8u) bands = 8u;
    if (bands > block_rows) bands = block_rows;
    if (bands < 1u) bands = 1u;
    const uint32_t rows_per_band = (block_rows + bands - 1) / bands;
    uint32_t n = 0;
    for (uint32_t s = 0; s < count; ++s)
        for (uint32_t b = 0; b * rows_per_band < block_rows; ++b) {
            const uint32_t r0 = b * rows_per_band * kResolveBlockH;
            uint32_t r1 = (b + 1) * rows_per_band * kResolveBlockH;
            if (r1 > sector_h) r1 = sector_h;
            if (n < cap && out) out[n] = topo_panorama_slot{s, r0, r1 - r0};
            ++n;
        }
    return n;
}

int TerrainRenderer::render_panorama(Comm* comm, const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h,
                                     float sun_theta_deg, float sun_phi_deg, int32_t view_mode, uint8_t* strip_dev, float* depth_dev) {
    if (!eye || !strip_dev) return fail(TOPO_ERR_INVALID, "null argument");
    if (sector_w == 0 || sector_h == 0) return fail(TOPO_ERR_INVALID, "sector size must be non-zero");
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    if (world > 1 && comm->device >= 0 && comm->device != device_) return fail(TOPO_ERR_INVALID, "the communicator lives on another device than the context");
    uint32_t first = 0, count = 0;
    panorama_sector_range(rank, world, &first, &count);
    topo_uniforms views[kPanoramaSectors];
    panorama_uniforms(eye, yaw0, pitch, sector_w, sector_h, sun_theta_deg, sun_phi_deg, view_mode, kPanoramaSectors, views);
    const size_t sector_px = (size_t)sector_w * sector_h;
    OutputParams o{};
    o.rgba = strip_dev + (size_t)first * sector_px * 4;
    o.rgba_view_stride = sector_px * 4;
    o.rgba_pitch = (size_t)sector_w * 4;
    o.depth = depth_dev ? depth_dev + (size_t)first * sector_px : nullptr;
    o.depth_view_stride = sector_px * 4;
    o.depth_pitch = (size_t)sector_w * 4;
    // (TOPO_PANORAMA_FORCE_SLOTS: the slot-by-slot resolve without an exchange, so that a one-GPU box can test it)
    const bool force_slots = world == 1 && getenv("TOPO_PANORAMA_FORCE_SLOTS") != nullptr;
    if (world == 1 && !force_slots) return render_views_device(count, views + first, sector_w, sector_h, o);      // nothing to ship: the ordinary path, frames in flight and all
    // ---- N > 1.  All of this rank's sectors in ONE cull / raster submission on the context's stream (frames in flight are
    // joined first: the exchange below is ordered against stream_), resolved slot by slot; behind each slot's k_resolve an
    // event releases the slot to the exchange stream, where every rank sends its part to every other rank and receives
    // theirs straight into place (grouped ncclSend / ncclRecv: on a fully connected xGMI node all seven links carry a
    // slot at once, where a ring all-gather is bound by one link).  The caller's stream -- stream_ -- waits for the last
    // exchange at the end, so "after topo_synchronize every rank holds the whole strip" stands.
    RcclApi& a = rccl();
    if (world > 1 && (!comm->nccl || !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd)) return fail(TOPO_ERR_HIP, a.error.empty() ? "no RCCL communicator" : a.error);
    if (int rc = join()) return rc;
    if (int rc = upload_tile_table()) return rc;
    topo_panorama_slot plan[kPanoramaSectors * 8];
    const uint32_t n_slots = panorama_slots(force_slots ? 2 : world, sector_w, sector_h, plan, kPanoramaSectors * 8) * (force_slots ? 2u : 1u);
    if (force_slots)      // (the two-rank plan covers four sectors: repeat it for the other four)
        for (uint32_t i = n_slots / 2; i < n_slots; ++i) plan[i] = topo_panorama_slot{plan[i - n_slots / 2].sector + 4u, plan[i - n_slots / 2].row0, plan[i - n_slots / 2].rows};
    while (world > 1 && comm->ready.size() < n_slots) {
        hipEvent_t e = nullptr;
        TOPO_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        comm->ready.push_back(e);
    }
    const uint32_t rblocks_x = (sector_w + kResolveBlockW - 1) / kResolveBlockW;
    const uint32_t rblocks_view = rblocks_x * ((sector_h + kResolveBlockH - 1) / kResolveBlockH);
    std::vector<ResolveSlot> slots(n_slots);
    for (uint32_t i = 0; i < n_slots; ++i) {
        slots[i].block_first = plan[i].sector * rblocks_view + plan[i].row0 / kResolveBlockH * rblocks_x;
        slots[i].block_count = (plan[i].rows + kResolveBlockH - 1) / kResolveBlockH * rblocks_x;
    }
    int comm_rc = TOPO_OK;
    std::string comm_err;
    const std::function<int(uint32_t, hipStream_t)> ship = [&](uint32_t i, hipStream_t frame_stream) -> int {
        TOPO_HIP_TRY(hipEventRecord(comm->ready[i], frame_stream));
        TOPO_HIP_TRY(hipStreamWaitEvent(comm->stream, comm->ready[i], 0));
        const size_t row_bytes = (size_t)sector_w * 4, off = (size_t)plan[i].row0 * row_bytes, bytes = (size_t)plan[i].rows * row_bytes;
        auto fail_nccl = [&](const char* what, RcclApi::Result rc) { comm_rc = TOPO_ERR_HIP; comm_err = std::string(what) + ": " + a.GetErrorString(rc); return TOPO_ERR_HIP; };
        if (auto rc = a.GroupStart()) return fail_nccl("ncclGroupStart", rc);
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            const size_t mine = ((size_t)first + plan[i].sector) * sector_px * 4 + off, theirs = ((size_t)p * count + plan[i].sector) * sector_px * 4 + off;
            if (auto rc = a.Send(strip_dev + mine, bytes, RcclApi::kChar, p, comm->nccl, comm->stream)) { (void)a.GroupEnd(); return fail_nccl("ncclSend (rgba)", rc); }
            if (auto rc = a.Recv(strip_dev + theirs, bytes, RcclApi::kChar, p, comm->nccl, comm->stream)) { (void)a.GroupEnd(); return fail_nccl("ncclRecv (rgba)", rc); }
            if (depth_dev) {
                if (auto rc = a.Send(depth_dev + mine / 4, bytes / 4, RcclApi::kFloat, p, comm->nccl, comm->stream)) { (void)a.GroupEnd(); return fail_nccl("ncclSend (depth)", rc); }
                if (auto rc = a.Recv(depth_dev + theirs / 4, bytes / 4, RcclApi::kFloat, p, comm->nccl, comm->stream)) { (void)a.GroupEnd(); return fail_nccl("ncclRecv (depth)", rc); }
            }
        }
        if (auto rc = a.GroupEnd()) return fail_nccl("ncclGroupEnd", rc);
        return TOPO_OK;
    };
    last_ctx_ = 0;
    const int rc = render_frame(ctx_[0], stream_, count, views + first, sector_w, sector_h, o, slots.data(), n_slots, world > 1 ? &ship : nullptr);
    if (world == 1) return rc;
    // whatever was queued on the exchange stream is waited for by the caller's stream, also on the error paths
    (void)hipEventRecord(comm->done, comm->stream);
    (void)hipStreamWaitEvent(stream_, comm->done, 0);
    if (comm_rc != TOPO_OK) return fail(comm_rc, comm_err);
    return rc;
}

// BASELINE config 5: a batch of viewpoints, each a full panorama of kPanoramaSectors sectors; viewpoints are independent,
// so a host shards them across its GPUs by calling this with its own share (no collective).  Eight viewpoints (64 views)
// go into one submission; with topo_set_pipeline_depth(d) d submissions are in flight.
int TerrainRenderer::render_batch(uint32_t n_viewpoints, const float* eyes, const float* yaw0s, const float* sun_theta_phi_deg, float pitch,
                                  uint32_t sector_w, uint32_t sector_h, int32_t view_mode, uint8_t* rgba_dev, float* depth_dev) {
    if (!eyes || !yaw0s || !sun_theta_phi_deg || !rgba_dev) return fail(TOPO_ERR_INVALID, "null argument");
    if (sector_w == 0 || sector_h == 0) return fail(TOPO_ERR_INVALID, "sector size must be non-zero");
    constexpr uint32_t kGroup = 8;
    const size_t sector_px = (size_t)sector_w * sector_h;
    std::vector<topo_uniforms> views(kGroup * kPanoramaSectors);
    for (uint32_t v0 = 0; v0 < n_viewpoints; v0 += kGroup) {
        const uint32_t nv = n_viewpoints - v0 < kGroup ? n_viewpoints - v0 : kGroup;
        for (uint32_t v = 0; v < nv; ++v)
            panorama_uniforms(eyes + 3 * (v0 + v), yaw0s[v0 + v], pitch, sector_w, sector_h, sun_theta_phi_deg[2 * (v0 + v)],
                              sun_theta_phi_deg[2 * (v0 + v) + 1], view_mode, kPanoramaSectors, views.data() + (size_t)v * kPanoramaSectors);
        OutputParams o{};
        o.rgba = rgba_dev + (size_t)v0 * kPanoramaSectors * sector_px * 4;
        o.rgba_view_stride = sector_px * 4;
        o.rgba_pitch = (size_t)sector_w * 4;
        o.depth = depth_dev ? depth_dev + (size_t)v0 * kPanoramaSectors * sector_px : nullptr;
        o.depth_view_stride = sector_px * 4;
        o.depth_pitch = (size_t)sector_w * 4;
        if (int rc = render_views_device(nv * kPanoramaSectors, views.data(), sector_w, sector_h, o)) return rc;
    }
    return TOPO_OK;
}

}  // namespace topo

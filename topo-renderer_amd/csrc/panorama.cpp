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
#include <string>
#include <vector>

#include "terrain_renderer.hpp"

namespace topo {

namespace {

struct RcclApi {
    // rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260, ncclGetErrorString :339, ncclAllGather :678
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, UniqueId id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    void* handle = nullptr;
    std::string error;
};
constexpr int kNcclChar = 0, kNcclFloat = 7;      // ncclDataType_t (rccl.h:459-466)

RcclApi& rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    // an already loaded copy first (RTLD_NOLOAD), then the usual names
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
        if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.handle)
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.handle) { api.error = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : ""); return api; }
    auto sym = [&](const char* s) { void* p = dlsym(api.handle, s); if (!p) api.error = std::string("librccl.so lacks ") + s; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    return api;
}

}  // namespace

struct Comm {
    void* nccl = nullptr;      // ncclComm_t
    int rank = 0, world = 1;
    bool owned = false;        // created by topo_comm_init (destroyed with the Comm) or borrowed (topo_comm_from_nccl)
};

int comm_unique_id(uint8_t out[128], std::string* err) {
    RcclApi& a = rccl();
    if (!a.error.empty() || !a.GetUniqueId) { *err = a.error.empty() ? "RCCL unavailable" : a.error; return TOPO_ERR_HIP; }
    RcclApi::UniqueId id;
    if (int rc = a.GetUniqueId(&id)) { *err = std::string("ncclGetUniqueId: ") + a.GetErrorString(rc); return TOPO_ERR_HIP; }
    memcpy(out, id.internal, 128);
    return TOPO_OK;
}

int comm_init(Comm** out, int device, const uint8_t id128[128], int rank, int world, std::string* err) {
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { *err = "rank/world out of range"; return TOPO_ERR_INVALID; }
    if (kPanoramaSectors % world != 0) { *err = "the sector count (8) must be divisible by the number of ranks"; return TOPO_ERR_INVALID; }
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    if (world > 1) {
        RcclApi& a = rccl();
        if (!a.error.empty()) { *err = a.error; delete c; return TOPO_ERR_HIP; }
        if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; delete c; return TOPO_ERR_HIP; }
        RcclApi::UniqueId id;
        memcpy(id.internal, id128, 128);
        if (int rc = a.CommInitRank(&c->nccl, world, id, rank)) { *err = std::string("ncclCommInitRank: ") + a.GetErrorString(rc); delete c; return TOPO_ERR_HIP; }
        c->owned = true;
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
    c->nccl = nccl_comm;
    c->rank = rank;
    c->world = world;
    *out = c;
    return TOPO_OK;
}

void comm_destroy(Comm* c) {
    if (!c) return;
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

int TerrainRenderer::render_panorama(const Comm* comm, const float eye[3], float yaw0, float pitch, uint32_t sector_w, uint32_t sector_h,
                                     float sun_theta_deg, float sun_phi_deg, int32_t view_mode, uint8_t* strip_dev, float* depth_dev) {
    if (!eye || !strip_dev) return fail(TOPO_ERR_INVALID, "null argument");
    if (sector_w == 0 || sector_h == 0) return fail(TOPO_ERR_INVALID, "sector size must be non-zero");
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
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
    // all of this rank's sectors in one submission (one set of kernel launches), on the context's stream
    const int saved_depth = pipeline_depth_;
    if (saved_depth != 1)
        if (int rc = set_pipeline_depth(1)) return rc;      // the gather below is ordered after the frame by the stream
    if (int rc = render_views_device(count, views + first, sector_w, sector_h, o)) return rc;
    if (world > 1) {
        RcclApi& a = rccl();
        if (!comm->nccl || !a.AllGather) return fail(TOPO_ERR_HIP, "no RCCL communicator");
        // in place: this rank's slice already sits at offset rank * slice of the receive buffer
        const size_t slice = (size_t)count * sector_px * 4;
        if (int rc = a.AllGather(strip_dev + (size_t)rank * slice, strip_dev, slice, kNcclChar, comm->nccl, stream_))
            return fail(TOPO_ERR_HIP, std::string("ncclAllGather (rgba): ") + a.GetErrorString(rc));
        if (depth_dev)
            if (int rc = a.AllGather(depth_dev + (size_t)rank * count * sector_px, depth_dev, (size_t)count * sector_px, kNcclFloat, comm->nccl, stream_))
                return fail(TOPO_ERR_HIP, std::string("ncclAllGather (depth): ") + a.GetErrorString(rc));
    }
    if (saved_depth != 1)
        if (int rc = set_pipeline_depth(saved_depth)) return rc;
    return TOPO_OK;
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

/* TEST-ONLY: include/topo_hip.h compiled as plain C99 and linked against libtopo_hip.so -- what a cgo / Rust-bindgen /
 * C host sees.  Exercises the host-side entry points (no GPU needed) and the error path of topo_create on a box without a
 * HIP device.  Built and run by tests/test_abi_cpu.py::test_header_is_plain_c. */
#include <stdio.h>
#include <string.h>

#include "topo_hip.h"

int main(void) {
    if (sizeof(topo_uniforms) != 160 || sizeof(topo_post_uniforms) != 16) return 10;
    if (topo_pad_256(8192) != 8192 || topo_pad_256(512) != 512 || topo_pad_256(513) != 768) return 11;
    float eye[3];
    topo_geometry_transform(1000.0f, 15.5f, 45.5f, eye);
    topo_uniforms u[8];
    memset(u, 0, sizeof u);
    topo_panorama_uniforms(eye, 0.0f, 0.0f, 2048, 4096, 15.5f, 45.5f, 0, 8, u);
    if (u[0].camera_proj[0] == 0.0f || u[7].view_mode != 0) return 12;
    const double ps[3] = {1.0 / 1200, 1.0 / 1200, 0.0}, tp[6] = {0, 0, 0, 15.0, 46.0, 0};
    float rp[2], mp[2], sc[2];
    if (topo_coordinate_transform(ps, 3, tp, 6, NULL, rp, mp, sc) != TOPO_OK || mp[0] != 15.0f) return 13;
    if (topo_coordinate_transform(ps, 3, tp, 6, ps, rp, mp, sc) != TOPO_ERR_UNSUPPORTED) return 14;
    int32_t locs[64];
    if (topo_locations_range(45.5f, 15.5f, 100000.0f, locs, 32) == 0) return 15;
    /* the multi-GPU entry points, degenerate world of one (needs neither RCCL nor a GPU) */
    topo_comm* comm = NULL;
    uint32_t first = 99, count = 99;
    if (topo_comm_init(&comm, 0, NULL, 0, 1) != TOPO_OK || comm == NULL) return 17;
    topo_panorama_sector_range(0, 1, &first, &count);
    if (first != 0 || count != TOPO_PANORAMA_SECTORS) return 18;
    topo_panorama_sector_range(3, 4, &first, &count);
    if (first != 6 || count != 2) return 19;
    topo_comm* bad = NULL;
    if (topo_comm_init(&bad, 0, NULL, 0, 3) != TOPO_ERR_INVALID || bad != NULL) return 20;      /* 8 sectors do not divide by 3 */
    if (topo_comm_init(&bad, 0, NULL, 2, 2) != TOPO_ERR_INVALID) return 21;
    topo_ctx* ctx = NULL;
    const int rc = topo_create(&ctx, 0, 64, 64, TOPO_FORMAT_RGBA8_UNORM_SRGB);
    if (rc == TOPO_OK) {                 /* a GPU is present: the context works and goes away again */
        if (topo_render_panorama(ctx, comm, eye, 0.0f, 0.0f, 64, 64, 15.5f, 45.5f, 0, NULL, NULL) != TOPO_ERR_INVALID) return 22;
        topo_destroy(ctx);
    } else if (rc != TOPO_ERR_HIP || ctx != NULL) {
        return 16;                       /* without a device: a clean error, no CPU fallback */
    }
    topo_comm_destroy(comm);
    printf("abi harness ok (topo_create rc %d)\n", rc);
    return 0;
}

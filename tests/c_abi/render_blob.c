/* Pure-C client of include/vello_hip.h (no Python, no C++): proves that the drop-in boundary is a plain C ABI.
 * Usage: render_blob <scene.bin> <out.rgba>
 * scene.bin = 10 x u32 layout, u32 width, u32 height, u32 base_color, u32 aa, u32 n_ramps, u64 scene_len,
 *             scene bytes, n_ramps * 512 ramp texels.
 * Exercises the blocking one-shot call and the animation form on a second context. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vello_hip.h"

static int fail(const char *what, vello_hip_ctx *ctx, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, vello_hip_last_error(ctx));
    return 1;
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    vello_hip_layout layout;
    uint32_t hdr[5];
    uint64_t scene_len;
    if (fread(&layout, sizeof layout, 1, f) != 1 || fread(hdr, sizeof hdr, 1, f) != 1 || fread(&scene_len, 8, 1, f) != 1) return 2;
    uint8_t *scene = (uint8_t *)malloc(scene_len ? scene_len : 1);
    if (scene_len && fread(scene, 1, scene_len, f) != scene_len) return 2;
    uint32_t n_ramps = hdr[4];
    uint32_t *ramps = n_ramps ? (uint32_t *)malloc((size_t)n_ramps * 512u * 4u) : NULL;
    if (n_ramps && fread(ramps, 4, (size_t)n_ramps * 512u, f) != (size_t)n_ramps * 512u) return 2;
    fclose(f);

    vello_hip_render_params params = {hdr[0], hdr[1], hdr[2], hdr[3]};
    size_t out_bytes = (size_t)params.width * params.height * 4u;
    uint8_t *out = (uint8_t *)calloc(out_bytes, 1);
    uint8_t *out2 = (uint8_t *)calloc(out_bytes, 1);

    vello_hip_ctx *ctx = NULL;
    int rc = vello_hip_create(0, VELLO_HIP_AA_MASK_ALL, NULL, &ctx);
    if (rc != VELLO_HIP_OK) return fail("vello_hip_create", NULL, rc);
    vello_hip_bump bump;
    rc = vello_hip_render(ctx, scene, (size_t)scene_len, &layout, &params, ramps, n_ramps, out, 0, 0, &bump);
    if (rc != VELLO_HIP_OK) return fail("vello_hip_render", ctx, rc);

    /* animation form: two frames in flight, each call brings the scene again; read the frame back through the test seam */
    if ((rc = vello_hip_set_frames_in_flight(ctx, 2)) != VELLO_HIP_OK) return fail("set_frames_in_flight", ctx, rc);
    for (int i = 0; i < 3; i++) {
        rc = vello_hip_render_frame(ctx, scene, (size_t)scene_len, &layout, &params, ramps, n_ramps, NULL, 0);
        if (rc != VELLO_HIP_OK) return fail("vello_hip_render_frame", ctx, rc);
    }
    if ((rc = vello_hip_sync(ctx)) != VELLO_HIP_OK) return fail("vello_hip_sync", ctx, rc);
    if ((rc = vello_hip_read_buffer(ctx, VELLO_HIP_BUF_OUTPUT, out2, 0, out_bytes)) != VELLO_HIP_OK) return fail("read_buffer", ctx, rc);
    if (memcmp(out, out2, out_bytes) != 0) {
        fprintf(stderr, "one-shot and pipelined frames differ\n");
        return 1;
    }
    vello_hip_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 1, out_bytes, f) != out_bytes) return 2;
    fclose(f);
    printf("ok lines=%u segments=%u failed=%u\n", bump.lines, bump.segments, bump.failed);
    return 0;
}

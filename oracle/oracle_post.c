/*
 * oracle_post.c -- CPU restatement of the reference's second post-processing mode, temporal
 * reprojection.  TEST INFRASTRUCTURE ONLY (see jpt_oracle.h).  PARITY UNPINNED: the reference holds no
 * test or golden image for it, and GLSL cannot run here.
 *
 * Citations: "T:n" = project/addons/jar_path_tracing/src/shaders/temporal_reprojection.glsl line n,
 * "M:n" = .../main.glsl, "H:n" = src/path_tracing/post_processing/temporal_reprojection.h.
 */
#include "jpt_oracle.h"
#include "oracle_pins.h"

/* imageStore(outputImage rgba8, vec4(radiance, 1)) of M:434 for a whole frame */
void jpto_screen_rgba8(const float *radiance_rgba, size_t n_pixels, uint8_t *screen_rgba8)
{
    for (size_t i = 0; i < n_pixels; i++) {
        for (int k = 0; k < 3; k++) screen_rgba8[i * 4 + k] = p_unorm8(radiance_rgba[i * 4 + k]);
        screen_rgba8[i * 4 + 3] = 255;
    }
}

/* T:19-27 (same curve as progressive_rendering.glsl:19-26) */
static float aces1(float v)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return p_clamp((v * (a * v + b)) / (v * (c * v + d) + e), 0.0f, 1.0f);
}

/* One dispatch of temporal_reprojection.glsl (T:30-72).  screen: rgba8 in/out; depth: the r32f image
 * main.glsl wrote for this frame (read-only); fb1/fb2: the two rgba32f history images. */
void jpto_temporal_reproject(const jpto_temporal_params *p, uint8_t *screen_rgba8, const float *depth, float *fb1, float *fb2)
{
    const int32_t W = p->width, H = p->height;
    const float *m = p->deltaMatrix;                    /* mat4 reprojectionMatrix, column-major (T:5) */
    const int use_first = (p->frame_count % 2u) == 0u;  /* T:46 */
    const float *fb_prev = use_first ? fb1 : fb2;       /* T:61 */
    float *fb_next = use_first ? fb2 : fb1;             /* T:67 */
    const float fw = (float)(uint32_t)W, fh = (float)(uint32_t)H;
    for (int32_t y = 0; y < H; y++)
        for (int32_t x = 0; x < W; x++) {
            const size_t i = (size_t)y * (size_t)W + (size_t)x;
            float cur[3], rep[3];
            for (int k = 0; k < 3; k++) rep[k] = cur[k] = p_from_unorm8(screen_rgba8[i * 4 + k]); /* T:35,48 */
            const float d = depth[i];                                                          /* T:36 */
            /* T:38-43 */
            const float nx = ((float)x + 0.5f) / fw * 2.0f - 1.0f;
            const float ny = ((float)y + 0.5f) / fh * -2.0f + 1.0f;
            if (p->frame_count > 0u) { /* T:49 */
                /* T:50: mat4 * vec4 = c0*x + c1*y + c2*z + c3*w, left to right (w = 1) */
                float cx = m[0] * nx + m[4] * ny + m[8] * d + m[12] * 1.0f;
                float cy = m[1] * nx + m[5] * ny + m[9] * d + m[13] * 1.0f;
                float cz = m[2] * nx + m[6] * ny + m[10] * d + m[14] * 1.0f;
                const float cw = m[3] * nx + m[7] * ny + m[11] * d + m[15] * 1.0f;
                cx = cx / cw; cy = cy / cw; cz = cz / cw; /* T:51 */
                const float u = (cx + 1.0f) * 0.5f;       /* T:53-56 */
                const float v = (1.0f - cy) * 0.5f;
                const int32_t px = p_f2i(u * fw), py = p_f2i(v * fh); /* T:57 */
                if (px >= 0 && px < W && py >= 0 && py < H) {         /* T:59 */
                    const size_t j = (size_t)py * (size_t)W + (size_t)px;
                    if (p_abs(depth[j] - cz) < 0.1f)
                        for (int k = 0; k < 3; k++) rep[k] = fb_prev[j * 4 + k]; /* T:60 */
                }
            }
            for (int k = 0; k < 3; k++) {
                const float blended = p_mix(cur[k], rep[k], 0.75f); /* T:64: the literal, not blendFactor */
                fb_next[i * 4 + k] = blended;                       /* T:66 */
                screen_rgba8[i * 4 + k] = p_unorm8(aces1(blended)); /* T:68-70 */
            }
            fb_next[i * 4 + 3] = 1.0f;
            screen_rgba8[i * 4 + 3] = 255;
        }
}

/* Orthographic front/back attribute rasteriser -- CPU oracle, TEST INFRASTRUCTURE ONLY.
 *
 * Restates what the reference obtains from OpenGL in utils/visualize_util.py:11-52 (render_cano_mesh)
 * with the 'vertex_attribute' shader (utils/renderer.py:10-29), gl_orthographic_projection_matrix
 * (:316-323) and Renderer.render (:432-451): both views look along the canonical z axis at the mesh
 * translated by -center; x,y in [-1,1] map onto the size x size image, row 0 at y = +1 (the reference
 * flips the GL read-back); the front view keeps the fragment with the LARGEST z, the back view (rotation
 * by pi about y, image flipped horizontally afterwards so that both maps are pixel-aligned) the one
 * with the SMALLEST z; back faces are culled (GL_CULL_FACE, counter-clockwise = front); the output is
 * the linearly interpolated per-vertex attribute (the canonical normal, NOT rotated), 0 where nothing
 * is drawn.
 *
 * PARITY UNPINNED against an actual OpenGL driver: sample positions (pixel centres), the top-left fill
 * rule and 8 sub-pixel bits are the GL specification's; depth ties and the last bits of the
 * interpolation are implementation-defined there.  Pinned: bit-exact agreement of the HIP kernel with
 * this file (tests/test_gpu_raster.py) and analytic checks (tests/test_raster_oracle.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC raster_oracle.c -o _build/libraster_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#define SUB 256 /* sub-pixel resolution */

static inline int64_t edge(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
/* top-left rule for a counter-clockwise triangle in a y-down raster: an edge is "top" if it is
 * horizontal and goes right-to-left... expressed on the fixed-point edge vector (dx, dy) */
static inline int is_top_left(int64_t dx, int64_t dy) { return (dy < 0) || (dy == 0 && dx < 0); }

/* view 0 = front, 1 = back.  out: size*size*3 floats (zeroed here). */
void raster_oracle(const float *verts, const float *attrs, int64_t nv, const int32_t *faces, int64_t nf,
                   const float center[3], int size, int view, float *out)
{
    const int64_t npix = (int64_t)size * size;
    float *zbuf = (float *)malloc(sizeof(float) * npix);
    int32_t *tbuf = (int32_t *)malloc(sizeof(int32_t) * npix);
    int64_t i;
    const float half = 0.5f * (float)size;
    for (i = 0; i < npix; ++i) { zbuf[i] = -INFINITY; tbuf[i] = -1; }
    for (i = 0; i < 3 * npix; ++i) out[i] = 0.0f;
    for (int64_t t = 0; t < nf; ++t) {
        int64_t fx[3], fy[3]; float zz[3];
        for (int k = 0; k < 3; ++k) {
            const float *p = verts + 3 * (int64_t)faces[3 * t + k];
            const float x = p[0] - center[0], y = p[1] - center[1], z = p[2] - center[2];
            /* column = (x+1)*size/2, row = (1-y)*size/2 (y up -> row down); depth key: larger = closer */
            fx[k] = (int64_t)floorf((x + 1.0f) * half * (float)SUB + 0.5f);
            fy[k] = (int64_t)floorf((1.0f - y) * half * (float)SUB + 0.5f);
            zz[k] = view == 0 ? z : -z;
        }
        /* Winding: a triangle that is counter-clockwise in GL window space (y up) seen from the front is,
         * in this y-down raster, clockwise: area < 0.  Front view keeps those; the back view (x mirrored
         * by the rotation, mirrored again by the flip -> same raster, opposite facing) keeps area > 0. */
        int64_t area = edge(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2]);
        if (view == 0) { if (area >= 0) continue; } else { if (area <= 0) continue; }
        int a = 0, b = 1, c = 2;
        if (area < 0) { b = 2; c = 1; area = -area; }          /* make (a,b,c) positively oriented in the raster */
        int64_t minx = fx[0], maxx = fx[0], miny = fy[0], maxy = fy[0];
        for (int k = 1; k < 3; ++k) { if (fx[k] < minx) minx = fx[k]; if (fx[k] > maxx) maxx = fx[k]; if (fy[k] < miny) miny = fy[k]; if (fy[k] > maxy) maxy = fy[k]; }
        int64_t x0 = (minx - SUB / 2 + SUB - 1) / SUB, x1 = (maxx - SUB / 2) / SUB;   /* pixel centres at (i+0.5)*SUB */
        int64_t y0 = (miny - SUB / 2 + SUB - 1) / SUB, y1 = (maxy - SUB / 2) / SUB;
        if (minx - SUB / 2 < 0) x0 = 0; if (miny - SUB / 2 < 0) y0 = 0;
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > size - 1) x1 = size - 1; if (y1 > size - 1) y1 = size - 1;
        const int tl0 = is_top_left(fx[c] - fx[b], fy[c] - fy[b]);      /* edge opposite a */
        const int tl1 = is_top_left(fx[a] - fx[c], fy[a] - fy[c]);
        const int tl2 = is_top_left(fx[b] - fx[a], fy[b] - fy[a]);
        for (int64_t py = y0; py <= y1; ++py)
            for (int64_t px = x0; px <= x1; ++px) {
                const int64_t sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
                const int64_t w0 = edge(fx[b], fy[b], fx[c], fy[c], sx, sy);
                const int64_t w1 = edge(fx[c], fy[c], fx[a], fy[a], sx, sy);
                const int64_t w2 = edge(fx[a], fy[a], fx[b], fy[b], sx, sy);
                if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
                const float inv = 1.0f / (float)area;
                const float l0 = (float)w0 * inv, l1 = (float)w1 * inv, l2 = (float)w2 * inv;
                const float z = (l0 * zz[a] + l1 * zz[b]) + l2 * zz[c];
                const int64_t pi = py * size + px;
                if (z > zbuf[pi] || (z == zbuf[pi] && (int32_t)t < tbuf[pi])) { zbuf[pi] = z; tbuf[pi] = (int32_t)t; }
            }
    }
    /* resolve */
    for (int64_t py = 0; py < size; ++py)
        for (int64_t px = 0; px < size; ++px) {
            const int64_t pi = py * size + px;
            const int32_t t = tbuf[pi];
            if (t < 0) continue;
            int64_t fx[3], fy[3];
            for (int k = 0; k < 3; ++k) {
                const float *p = verts + 3 * (int64_t)faces[3 * (int64_t)t + k];
                const float x = p[0] - center[0], y = p[1] - center[1];
                fx[k] = (int64_t)floorf((x + 1.0f) * half * (float)SUB + 0.5f);
                fy[k] = (int64_t)floorf((1.0f - y) * half * (float)SUB + 0.5f);
            }
            int a = 0, b = 1, c = 2;
            int64_t area = edge(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2]);
            if (area < 0) { b = 2; c = 1; area = -area; }
            const int64_t sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
            const float inv = 1.0f / (float)area;
            const float l0 = (float)edge(fx[b], fy[b], fx[c], fy[c], sx, sy) * inv;
            const float l1 = (float)edge(fx[c], fy[c], fx[a], fy[a], sx, sy) * inv;
            const float l2 = (float)edge(fx[a], fy[a], fx[b], fy[b], sx, sy) * inv;
            const float *A = attrs + 3 * (int64_t)faces[3 * (int64_t)t + a], *B = attrs + 3 * (int64_t)faces[3 * (int64_t)t + b],
                        *Cc = attrs + 3 * (int64_t)faces[3 * (int64_t)t + c];
            for (int k = 0; k < 3; ++k) out[3 * pi + k] = (l0 * A[k] + l1 * B[k]) + l2 * Cc[k];
        }
    free(zbuf); free(tbuf);
    (void)nv;
}

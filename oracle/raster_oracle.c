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

/* ---- general model-view-projection view (utils/renderer.py: vs_position / vs_vertex_attribute :10-51, Renderer.render
 * :432-451), as normal_fusion.canonicalize_normal_map uses it with gl_perspective_projection_matrix (:297-312) to get the
 * live mesh's position map (normal_fusion.py:14-20).
 *   clip = mvp * (v, 1); ndc = clip.xyz / clip.w; column = (ndc.x + 1) * W / 2, row = (1 - ndc.y) * H / 2 (the reference
 *   flips the GL read-back, renderer.py:449); counter-clockwise-in-GL-window = front, back faces culled; GL_LESS depth
 *   test on ndc.z, fragments with ndc.z outside [-1, 1] dropped; attributes interpolated perspective-correctly
 *   (sum l_i a_i / w_i) / (sum l_i / w_i); output RGBA = (attribute, 1), background 0.
 * Deviation from GL: a triangle with a vertex at w <= 0 is dropped instead of clipped against the near plane.
 * attrs == NULL renders the positions (the 'position' shader).  mvp row-major (the reference uploads with GL_TRUE). */
static inline void xform(const float m[16], const float *p, float c[4])
{
    for (int r = 0; r < 4; ++r) c[r] = ((m[4 * r] * p[0] + m[4 * r + 1] * p[1]) + m[4 * r + 2] * p[2]) + m[4 * r + 3];
}

static int setup_mvp(const float *verts, const int32_t *faces, int64_t t, const float m[16], int W, int H,
                     int64_t fx[3], int64_t fy[3], float zn[3], float iw[3])
{
    for (int k = 0; k < 3; ++k) {
        float c[4];
        xform(m, verts + 3 * (int64_t)faces[3 * t + k], c);
        if (!(c[3] > 0.0f)) return 0;
        iw[k] = 1.0f / c[3];
        const float nx = c[0] * iw[k], ny = c[1] * iw[k];
        zn[k] = c[2] * iw[k];
        const float px = (nx + 1.0f) * (0.5f * (float)W) * (float)SUB + 0.5f, py = (1.0f - ny) * (0.5f * (float)H) * (float)SUB + 0.5f;
        if (!(fabsf(px) < 1.0e12f) || !(fabsf(py) < 1.0e12f)) return 0;          /* far off screen / non-finite */
        fx[k] = (int64_t)floorf(px); fy[k] = (int64_t)floorf(py);
    }
    return 1;
}

void raster_mvp_oracle(const float *verts, const float *attrs, int64_t nv, const int32_t *faces, int64_t nf,
                       const float mvp[16], int W, int H, float *out /* H*W*4 */)
{
    const int64_t npix = (int64_t)W * H;
    float *zbuf = (float *)malloc(sizeof(float) * npix);
    int32_t *tbuf = (int32_t *)malloc(sizeof(int32_t) * npix);
    if (!attrs) attrs = verts;
    for (int64_t i = 0; i < npix; ++i) { zbuf[i] = INFINITY; tbuf[i] = -1; }
    for (int64_t i = 0; i < 4 * npix; ++i) out[i] = 0.0f;
    for (int64_t t = 0; t < nf; ++t) {
        int64_t fx[3], fy[3]; float zn[3], iw[3];
        if (!setup_mvp(verts, faces, t, mvp, W, H, fx, fy, zn, iw)) continue;
        int64_t area = edge(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2]);
        if (area >= 0) continue;                                   /* back-facing or degenerate */
        const int a = 0, b = 2, c = 1; area = -area;
        int64_t minx = fx[0], maxx = fx[0], miny = fy[0], maxy = fy[0];
        for (int k = 1; k < 3; ++k) { if (fx[k] < minx) minx = fx[k]; if (fx[k] > maxx) maxx = fx[k]; if (fy[k] < miny) miny = fy[k]; if (fy[k] > maxy) maxy = fy[k]; }
        if (maxx < 0 || maxy < 0 || minx > (int64_t)W * SUB || miny > (int64_t)H * SUB) continue;
        int64_t x0 = (minx - SUB / 2 + SUB - 1) / SUB, x1 = (maxx - SUB / 2) / SUB;
        int64_t y0 = (miny - SUB / 2 + SUB - 1) / SUB, y1 = (maxy - SUB / 2) / SUB;
        if (minx - SUB / 2 < 0) x0 = 0; if (miny - SUB / 2 < 0) y0 = 0;
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
        const int tl0 = is_top_left(fx[c] - fx[b], fy[c] - fy[b]);
        const int tl1 = is_top_left(fx[a] - fx[c], fy[a] - fy[c]);
        const int tl2 = is_top_left(fx[b] - fx[a], fy[b] - fy[a]);
        const float inv = 1.0f / (float)area;
        for (int64_t py = y0; py <= y1; ++py)
            for (int64_t px = x0; px <= x1; ++px) {
                const int64_t sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
                const int64_t w0 = edge(fx[b], fy[b], fx[c], fy[c], sx, sy);
                const int64_t w1 = edge(fx[c], fy[c], fx[a], fy[a], sx, sy);
                const int64_t w2 = edge(fx[a], fy[a], fx[b], fy[b], sx, sy);
                if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
                const float l0 = (float)w0 * inv, l1 = (float)w1 * inv, l2 = (float)w2 * inv;
                const float z = (l0 * zn[a] + l1 * zn[b]) + l2 * zn[c];
                if (!(z >= -1.0f && z <= 1.0f)) continue;
                const int64_t pi = py * W + px;
                if (z < zbuf[pi] || (z == zbuf[pi] && (int32_t)t < tbuf[pi])) { zbuf[pi] = z; tbuf[pi] = (int32_t)t; }
            }
    }
    for (int64_t py = 0; py < H; ++py)
        for (int64_t px = 0; px < W; ++px) {
            const int64_t pi = py * W + px;
            const int32_t t = tbuf[pi];
            if (t < 0) continue;
            int64_t fx[3], fy[3]; float zn[3], iw[3];
            setup_mvp(verts, faces, t, mvp, W, H, fx, fy, zn, iw);
            const int a = 0, b = 2, c = 1;
            const int64_t area = -edge(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2]);
            const int64_t sx = px * SUB + SUB / 2, sy = py * SUB + SUB / 2;
            const float inv = 1.0f / (float)area;
            const float u0 = (float)edge(fx[b], fy[b], fx[c], fy[c], sx, sy) * inv * iw[a];
            const float u1 = (float)edge(fx[c], fy[c], fx[a], fy[a], sx, sy) * inv * iw[b];
            const float u2 = (float)edge(fx[a], fy[a], fx[b], fy[b], sx, sy) * inv * iw[c];
            const float den = 1.0f / ((u0 + u1) + u2);
            const float *A = attrs + 3 * (int64_t)faces[3 * (int64_t)t + a], *B = attrs + 3 * (int64_t)faces[3 * (int64_t)t + b],
                        *Cc = attrs + 3 * (int64_t)faces[3 * (int64_t)t + c];
            for (int k = 0; k < 3; ++k) out[4 * pi + k] = ((u0 * A[k] + u1 * B[k]) + u2 * Cc[k]) * den;
            out[4 * pi + 3] = 1.0f;
        }
    free(zbuf); free(tbuf);
    (void)nv;
}

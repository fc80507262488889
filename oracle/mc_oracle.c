/* Marching-cubes CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/avatarcap_oracle.py header).
 *
 * PARITY UNPINNED.  The reference calls skimage.measure.marching_cubes(vol, iso, spacing=voxel)
 * (utils/recon_util.py:64; scikit-image==0.17.2, requirements.txt:8, default method = Lewiner).
 * scikit-image is neither vendored under /root/reference nor installable here, and the
 * reference holds no test or golden vector for it, so this file restates the *published*
 * algorithm family instead of the library:
 *
 *   - marching cubes over cells of 8 samples; a corner is "above" when value > iso;
 *   - one vertex per sign-changing grid edge, shared by the (up to 4) cells around it, at the
 *     linear root t = (iso - v0) / (v1 - v0), position (index + t) * spacing  (what Lewiner's
 *     1/|v| corner weighting reduces to, up to FLT_EPSILON);
 *   - ambiguous faces (diagonal corners alike) are resolved with the asymptotic decider on the
 *     four face values -- Lewiner's face test -- so neighbouring cells always agree and the
 *     surface is watertight;
 *   - NOT restated: Lewiner's interior (tunnel) tests and his hand-tuned tilings; each boundary
 *     loop of a cell is triangulated without extra vertices instead, avoiding chords that lie inside a cube face
 *     wherever the loop allows it (triangulate_loop).  The surface topology can therefore differ from
 *     scikit-image inside cells of cases 4/6/7/10/12/13, and triangle diagonals can differ anywhere.
 *   - degenerate triangles are kept (allow_degenerate=True is the library default).
 *
 * What IS pinned (tests/test_mc_*.py): closed 2-manifoldness on analytic fields, Euler
 * characteristic, every vertex on a grid edge at the linear root, area/volume vs analytic,
 * determinism, and bit-exact agreement of the HIP kernel with this file.
 *
 * Canonical output order (shared with the HIP kernel, include/avcap.h):
 *   vertices: ascending (voxel linear index i = x*Y*Z + y*Z + z, then axis 0,1,2) of the owning
 *             edge (the edge from voxel i towards +axis);
 *   faces:    ascending cell linear index, then loop order (by smallest cube-edge id), then triangulate_loop's order.
 *   winding:  right-hand normal points towards HIGHER values (the reference then flips,
 *             recon_util.py:69).
 *
 * Cube conventions: corner c = dx + 2*dy + 4*dz.  Cube edge e = 4*axis + j with
 *   axis 0: j = dy + 2*dz,   axis 1: j = dx + 2*dz,   axis 2: j = dx + 2*dy.
 * Faces f = 2*axis + side, corners listed counter-clockwise seen from outside the cube.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC mc_oracle.c -o _build/libmc_oracle.so
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static const int FACE_CORNERS[6][4] = {
    /* -x */ {0, 4, 6, 2}, /* +x */ {1, 3, 7, 5},
    /* -y */ {0, 1, 5, 4}, /* +y */ {2, 6, 7, 3},
    /* -z */ {0, 2, 3, 1}, /* +z */ {4, 5, 7, 6}};

/* cube edge id joining two corners that differ in exactly one bit */
static int edge_between(int a, int b)
{
    int lo = a < b ? a : b, d = a ^ b;
    int dx = lo & 1, dy = (lo >> 1) & 1, dz = (lo >> 2) & 1;
    if (d == 1) return 0 + dy + 2 * dz;
    if (d == 2) return 4 + dx + 2 * dz;
    return 8 + dx + 2 * dy;
}

/* the two cube faces a cube edge lies on, as a 6-bit mask */
static int edge_face_mask(int e)
{
    const int axis = e >> 2, j = e & 3;
    const int a = axis == 0 ? 2 * (j & 1) + 4 * (j >> 1) : (axis == 1 ? (j & 1) + 4 * (j >> 1) : (j & 1) + 2 * (j >> 1));
    const int b = a | (1 << axis);
    int f, k, m = 0;
    for (f = 0; f < 6; ++f) {
        int ha = 0, hb = 0;
        for (k = 0; k < 4; ++k) { ha |= FACE_CORNERS[f][k] == a; hb |= FACE_CORNERS[f][k] == b; }
        if (ha && hb) m |= 1 << f;
    }
    return m;
}

/* Triangulate the polygon loop[0..n-1] without new vertices, with as few chords inside a cube face as possible (such a
 * chord can coincide with one drawn by the neighbouring cell in the same face, which would make an edge shared by four
 * triangles); ties -> the largest split index, i.e. a fan from loop[0] when nothing else matters. */
static int tri_cost[12][12], tri_split[12][12];
static int emit_tris(const int *loop, int i, int j, int *tri, int nt)
{
    int k;
    if (j - i < 2) return nt;
    k = tri_split[i][j];
    nt = emit_tris(loop, i, k, tri, nt);
    tri[3 * nt] = loop[i]; tri[3 * nt + 1] = loop[k]; tri[3 * nt + 2] = loop[j]; ++nt;
    return emit_tris(loop, k, j, tri, nt);
}
static int triangulate_loop(const int *loop, int n, int *tri, int nt)
{
    int span, i, j, k, fm[12];
    for (i = 0; i < n; ++i) fm[i] = edge_face_mask(loop[i]);
    for (i = 0; i < n; ++i) for (j = 0; j < n; ++j) { tri_cost[i][j] = 0; tri_split[i][j] = -1; }
    for (span = 2; span < n; ++span)
        for (i = 0; i + span < n; ++i) {
            int best = -1;
            j = i + span;
            for (k = j - 1; k > i; --k) {
                /* a face is the + side of one cell and the - side of its neighbour: chords in + faces (mask 0x2a) cost 1000,
                 * so the two cells can only ever draw the same chord when one of them has no other choice */
                const int cik = (k - i != 1 && k - i != n - 1) ? (fm[i] & fm[k]) : 0, ckj = (j - k != 1 && j - k != n - 1) ? (fm[k] & fm[j]) : 0;
                const int wik = !cik ? 0 : ((cik & 0x2a) ? 1000 : 1), wkj = !ckj ? 0 : ((ckj & 0x2a) ? 1000 : 1);
                const int c = tri_cost[i][k] + tri_cost[k][j] + wik + wkj;
                if (best < 0 || c < best) { best = c; tri_split[i][j] = k; }
            }
            tri_cost[i][j] = best;
        }
    return emit_tris(loop, 0, n - 1, tri, nt);
}

/* trace the oriented boundary loops of one cell.  val[c] = sample - iso.  Returns the number
 * of triangles; tri[3*k..] are cube-edge ids. */
static int cell_triangles(const float val[8], int tri[36])
{
    int above[8], c, f, i;
    int nxt[12];
    for (i = 0; i < 12; ++i) nxt[i] = -1;
    for (c = 0; c < 8; ++c) above[c] = val[c] > 0.0f;
    for (f = 0; f < 6; ++f) {
        const int *P = FACE_CORNERS[f];
        int s[4], E[4], ncut = 0;
        for (i = 0; i < 4; ++i) { s[i] = above[P[i]]; E[i] = edge_between(P[i], P[(i + 1) & 3]); }
        for (i = 0; i < 4; ++i) ncut += s[i] != s[(i + 1) & 3];
        if (ncut == 0) continue;
        if (ncut == 2) {
            int st = -1, en = -1;
            for (i = 0; i < 4; ++i) {
                if (s[i] && !s[(i + 1) & 3]) st = E[i];      /* above -> below: segment start */
                if (!s[i] && s[(i + 1) & 3]) en = E[i];      /* below -> above: segment end   */
            }
            nxt[st] = en;
        } else {
            /* ambiguous: asymptotic decider, products rounded separately (no fma) */
            float a = val[P[0]], b = val[P[1]], cc = val[P[2]], d = val[P[3]];
            volatile float p = a * cc, q = b * d;
            int connected = s[0] ? (p > q) : (q > p);        /* "above" corners joined through the face */
            int k0 = s[0] ? 0 : 1;                            /* index of an above corner */
            /* above corners at k0 and k0+2; starts are E[k0], E[k0+2]; ends are E[k0-1], E[k0+1] */
            int sA = E[k0], sB = E[(k0 + 2) & 3], eA = E[(k0 + 3) & 3], eB = E[(k0 + 1) & 3];
            if (connected) { nxt[sA] = eB; nxt[sB] = eA; }    /* cut around the below corners */
            else           { nxt[sA] = eA; nxt[sB] = eB; }    /* cut around the above corners */
        }
    }
    {
        int visited[12] = {0}, nt = 0, e;
        for (e = 0; e < 12; ++e) {
            int loop[12], n = 0, cur;
            if (nxt[e] < 0 || visited[e]) continue;
            cur = e;
            do { loop[n++] = cur; visited[cur] = 1; cur = nxt[cur]; } while (cur != e && n < 12);
            nt = triangulate_loop(loop, n, tri, nt);
        }
        return nt;
    }
}

/* exported so tests can compare single cells with the generated GPU tables */
int mc_oracle_cell(const float val[8], int tri[36]) { return cell_triangles(val, tri); }

void mc_oracle_free(void *p) { free(p); }

/* returns 0 on success.  verts: nv*3 floats, faces: nf*3 ints (malloc'ed, free with mc_oracle_free) */
int mc_oracle(const float *vol, int X, int Y, int Z, float iso, const float spacing[3],
              float **verts_out, int64_t *nv_out, int32_t **faces_out, int64_t *nf_out)
{
    const int64_t N = (int64_t)X * Y * Z, sY = Z, sX = (int64_t)Y * Z;
    const int dim[3] = {X, Y, Z};
    const int64_t stride[3] = {sX, sY, 1};
    int32_t *vid = (int32_t *)malloc(sizeof(int32_t) * 3 * N);
    int64_t nv = 0, nf = 0, capf = 1024, i;
    float *verts; int32_t *faces;
    int x, y, z, a;
    if (!vid) return -1;
    /* pass 1: vertex ids + count */
    for (x = 0; x < X; ++x) for (y = 0; y < Y; ++y) for (z = 0; z < Z; ++z) {
        int64_t li = x * sX + y * sY + z;
        int p[3] = {x, y, z};
        int s0 = (vol[li] - iso) > 0.0f;
        for (a = 0; a < 3; ++a) {
            int32_t id = -1;
            if (p[a] + 1 < dim[a]) { int s1 = (vol[li + stride[a]] - iso) > 0.0f; if (s0 != s1) id = (int32_t)nv++; }
            vid[3 * li + a] = id;
        }
    }
    verts = (float *)malloc(sizeof(float) * 3 * (nv > 0 ? nv : 1));
    for (x = 0; x < X; ++x) for (y = 0; y < Y; ++y) for (z = 0; z < Z; ++z) {
        int64_t li = x * sX + y * sY + z;
        int p[3] = {x, y, z};
        for (a = 0; a < 3; ++a) {
            int32_t id = vid[3 * li + a];
            if (id >= 0) {
                float v0 = vol[li] - iso, v1 = vol[li + stride[a]] - iso;
                volatile float num = 0.0f - v0, den = v1 - v0;
                float t = num / den;
                int b;
                for (b = 0; b < 3; ++b) {
                    volatile float idx = (float)p[b] + (b == a ? t : 0.0f);
                    verts[3 * (int64_t)id + b] = idx * spacing[b];
                }
            }
        }
    }
    /* pass 2: faces */
    faces = (int32_t *)malloc(sizeof(int32_t) * 3 * capf);
    for (x = 0; x + 1 < X; ++x) for (y = 0; y + 1 < Y; ++y) for (z = 0; z + 1 < Z; ++z) {
        int64_t li = x * sX + y * sY + z;
        float val[8]; int tri[36], nt, c, k;
        int any = 0, all = 1;
        for (c = 0; c < 8; ++c) {
            val[c] = vol[li + (c & 1) * sX + ((c >> 1) & 1) * sY + ((c >> 2) & 1)] - iso;
            if (val[c] > 0.0f) any = 1; else all = 0;
        }
        if (!any || all) continue;
        nt = cell_triangles(val, tri);
        if (nf + nt > capf) { while (nf + nt > capf) capf *= 2; faces = (int32_t *)realloc(faces, sizeof(int32_t) * 3 * capf); }
        for (k = 0; k < 3 * nt; ++k) {
            int e = tri[k], axis = e >> 2, j = e & 3;
            int o[3];
            if (axis == 0) { o[0] = 0; o[1] = j & 1; o[2] = j >> 1; }
            else if (axis == 1) { o[0] = j & 1; o[1] = 0; o[2] = j >> 1; }
            else { o[0] = j & 1; o[1] = j >> 1; o[2] = 0; }
            faces[3 * nf + k] = vid[3 * (li + o[0] * sX + o[1] * sY + o[2]) + axis];
        }
        nf += nt;
    }
    free(vid);
    *verts_out = verts; *nv_out = nv; *faces_out = faces; *nf_out = nf;
    (void)i;
    return 0;
}

#!/usr/bin/env python3
"""Generates avatarcap_amd/csrc/mc_tables.h: the marching-cubes case tables of the HIP kernel.

Independent (Python) implementation of the cell rule that oracle/mc_oracle.c states in C:
oriented boundary-loop tracing on the 6 cube faces, ambiguous faces resolved by a per-face
"above corners connected" bit (the kernel evaluates the asymptotic decider at run time), every loop
triangulated without new vertices so that as few triangle edges as possible lie inside a cube face (such a chord can
coincide with one the neighbouring cell draws in the same face -> an edge shared by four triangles); among the
triangulations with the fewest such chords the one closest to a fan from the loop's smallest cube-edge id is taken.  tests/test_mc_tables.py checks the generated
tables against the C oracle cell by cell, for all 256 configurations and all decider outcomes.

Table layout (all uint32 words, copied into LDS by the kernel):
  cfg_info[256]: bits 0..15 first row of the configuration, bits 16..18 number of ambiguous
                 faces n, bits 19..36 -> second word: the ambiguous face ids (3 bits each)
                 => two words per configuration: {row | n << 16, faces packed 3 bits each}
  rows[R][4]   : 16 bytes per (configuration, variant): byte 0 = triangle count T, then 30 nibbles
                 = cube-edge ids of the T triangles (3 per triangle), low nibble first.
  variant bit i = decider outcome of the i-th ambiguous face (ascending face id).
"""
import os
import sys

FACE_CORNERS = [(0, 4, 6, 2), (1, 3, 7, 5), (0, 1, 5, 4), (2, 6, 7, 3), (0, 2, 3, 1), (4, 5, 7, 6)]


def edge_between(a, b):
    lo, d = min(a, b), a ^ b
    dx, dy, dz = lo & 1, (lo >> 1) & 1, (lo >> 2) & 1
    if d == 1:
        return 0 + dy + 2 * dz
    if d == 2:
        return 4 + dx + 2 * dz
    return 8 + dx + 2 * dy


def face_state(cfg, f):
    P = FACE_CORNERS[f]
    s = [(cfg >> c) & 1 for c in P]
    ncut = sum(s[i] != s[(i + 1) & 3] for i in range(4))
    return P, s, ncut


def ambiguous_faces(cfg):
    return [f for f in range(6) if face_state(cfg, f)[2] == 4]


def edge_corners(e):
    axis, j = divmod(e, 4)
    a = (2 * (j & 1) + 4 * (j >> 1), (j & 1) + 4 * (j >> 1), (j & 1) + 2 * (j >> 1))[axis]
    return a, a | (1 << axis)


EDGE_FACES = [frozenset(f for f in range(6) if all(c in FACE_CORNERS[f] for c in edge_corners(e))) for e in range(12)]


PLUS_FACE_COST = 1000


def triangulate(loop):
    """Triangles (index triples into `loop`) of the polygon 0..n-1: dynamic programme over (i, j) chains, cost = number of
    chords whose two cube edges share a cube face; ties -> the largest split index (a fan from vertex 0 when nothing else matters)."""
    n = len(loop)

    def w(i, j):
        common = EDGE_FACES[loop[i]] & EDGE_FACES[loop[j]] if (j - i) not in (1, n - 1) else ()
        # a face is the + side of one cell and the - side of its neighbour: chords in + faces are (all but) forbidden, so the
        # two cells can only ever draw the same chord when one of them has no other choice
        return 0 if not common else (PLUS_FACE_COST if any(f & 1 for f in common) else 1)
    cost = [[0] * n for _ in range(n)]
    split = [[-1] * n for _ in range(n)]
    for span in range(2, n):
        for i in range(0, n - span):
            j = i + span
            best = None
            for k in range(j - 1, i, -1):
                c = cost[i][k] + cost[k][j] + w(i, k) + w(k, j)
                if best is None or c < best:
                    best, split[i][j] = c, k
            cost[i][j] = best
    out = []

    def emit(i, j):
        if j - i < 2:
            return
        k = split[i][j]
        emit(i, k)
        out.append((i, k, j))
        emit(k, j)
    emit(0, n - 1)
    return out


def triangles(cfg, connected_by_face):
    nxt = {}
    for f in range(6):
        P, s, ncut = face_state(cfg, f)
        E = [edge_between(P[i], P[(i + 1) & 3]) for i in range(4)]
        if ncut == 2:
            st = [E[i] for i in range(4) if s[i] and not s[(i + 1) & 3]][0]
            en = [E[i] for i in range(4) if not s[i] and s[(i + 1) & 3]][0]
            nxt[st] = en
        elif ncut == 4:
            k0 = 0 if s[0] else 1
            sA, sB, eA, eB = E[k0], E[(k0 + 2) & 3], E[(k0 + 3) & 3], E[(k0 + 1) & 3]
            if connected_by_face[f]:
                nxt[sA], nxt[sB] = eB, eA
            else:
                nxt[sA], nxt[sB] = eA, eB
    tris, seen = [], set()
    for e in range(12):
        if e in nxt and e not in seen:
            loop, cur = [], e
            while True:
                loop.append(cur); seen.add(cur); cur = nxt[cur]
                if cur == e:
                    break
            for (a, b, c) in triangulate(loop):
                tris.append((loop[a], loop[b], loop[c]))
    return tris


def build():
    info, rows = [], []
    for cfg in range(256):
        amb = ambiguous_faces(cfg)
        base = len(rows)
        for variant in range(1 << len(amb)):
            conn = {f: (variant >> i) & 1 for i, f in enumerate(amb)}
            t = triangles(cfg, conn)
            assert len(t) <= 10
            nib = [e for tri in t for e in tri] + [0] * (30 - 3 * len(t))
            b = [len(t)] + [nib[2 * i] | (nib[2 * i + 1] << 4) for i in range(15)]
            rows.append([b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24) for w in range(4)])
        faces = 0
        for i, f in enumerate(amb):
            faces |= f << (3 * i)
        info.append((base | (len(amb) << 16), faces))
    return info, rows


def main(out_path):
    info, rows = build()
    with open(out_path, 'w') as fh:
        fh.write('// GENERATED by tools/gen_mc_tables.py -- do not edit.  Layout documented there.\n#pragma once\n#include <stdint.h>\n')
        fh.write(f'namespace avc {{ namespace mc {{\nconstexpr int N_ROWS = {len(rows)};\n')
        fh.write('static const uint32_t CFG_INFO[256][2] = {\n')
        for a, b in info:
            fh.write(f'  {{0x{a:08x}u, 0x{b:08x}u}},\n')
        fh.write('};\nstatic const uint32_t ROWS[N_ROWS][4] = {\n')
        for r in rows:
            fh.write('  {' + ', '.join(f'0x{w:08x}u' for w in r) + '},\n')
        fh.write('};\n}}  // namespace avc::mc\n')
    print(f'{out_path}: {len(rows)} rows')


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, '..', 'avatarcap_amd', 'csrc', 'mc_tables.h'))

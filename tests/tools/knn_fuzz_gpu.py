"""Fuzz of the HIP KNN (csrc/knn_lbs.hip): every search path -- the default mix, per-lane rings, the wave-cooperative scan -- and the bound form with candidate
lists (avc_lbs_prepare at a random reach) against the exhaustive LDS-tiled scan, bit for bit (squared distances and indices; ties -> lower index), on reference
sets the grid was not tuned for: 4 .. 20,000 points, clustered, collinear, coplanar, on a lattice (masses of exact ties), duplicated, with queries inside, far outside
and exactly on reference points.  Small cases are also held to the fp32 oracle (oracle/avatarcap_oracle.knn).
    python tests/tools/knn_fuzz_gpu.py [cases] [seed]          (needs an MI355X; a checker script, not part of the product path)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avatarcap_amd import _lib, config                                  # noqa: E402
config.cfg = config.default_cfg(); config.device = torch.device('cuda')
from avatarcap_amd.utils.smpl_util import SmplUtil                      # noqa: E402
from oracle import avatarcap_oracle as orc                              # noqa: E402


def points(rs, n, kind):
    if kind == 'uniform':
        return rs.uniform(-1, 1, (n, 3))
    if kind == 'clusters':
        c = rs.uniform(-1, 1, (rs.randint(1, 6), 3))
        return c[rs.randint(0, c.shape[0], n)] + rs.normal(0, 10.0 ** rs.uniform(-4, -1), (n, 3))
    if kind == 'line':
        return np.outer(rs.uniform(-1, 1, n), rs.normal(0, 1, 3)) + rs.normal(0, 1e-6, (n, 3)) * rs.randint(0, 2)
    if kind == 'plane':
        p = rs.uniform(-1, 1, (n, 3)); p[:, rs.randint(0, 3)] = rs.uniform(-1, 1)
        return p
    if kind == 'lattice':                                               # exact ties everywhere
        return rs.randint(-4, 5, (n, 3)) * 0.125
    raise ValueError(kind)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 4711)
    su = SmplUtil()
    bad, t0, nq_total = 0, time.time(), 0
    kinds = ['uniform', 'clusters', 'line', 'plane', 'lattice']
    try:
        for k in range(cases):
            nr = int(rs.choice([4, 5, 17, 63, 64, 65, 300, 1000, 6890, 20000]))
            nq = int(rs.choice([1, 63, 64, 65, 1000, 5000]))
            K = int(rs.choice([1, 4, 8]))
            if K > nr:
                K = 4 if nr >= 4 else 1
            rk, qk = kinds[rs.randint(0, 5)], kinds[rs.randint(0, 5)]
            ref = points(rs, nr, rk).astype(np.float32)
            if rs.rand() < 0.4 and nr > 8:
                ref[rs.randint(0, nr, nr // 8)] = ref[rs.randint(0, nr, nr // 8)]           # duplicated reference points
            q = points(rs, nq, qk).astype(np.float32) * np.float32(rs.choice([0.5, 1.0, 3.0]))
            if rs.rand() < 0.5:
                m = min(nq, nr) // 2
                q[:m] = ref[rs.randint(0, nr, m)]                                           # queries exactly on reference points
            if rs.rand() < 0.3:
                q[-1] = 50.0
            qt, rt = torch.from_numpy(q[None]).cuda(), torch.from_numpy(ref[None]).cuda()
            _lib.set_option('knn_search', 3)
            d_b, i_b = su.knn_points(qt, rt, K=K)
            why = []
            for path in (0, 1, 2):
                _lib.set_option('knn_search', path)
                d_p, i_p = su.knn_points(qt, rt, K=K)
                if not (torch.equal(i_p, i_b) and torch.equal(d_p, d_b)):
                    why.append(f'search path {path}')
            _lib.set_option('knn_search', 0)
            if nq * nr <= 2_000_000:
                od2, oidx = orc.knn(q, ref, K)
                if not (np.array_equal(i_b[0].cpu().numpy(), oidx) and np.array_equal(d_b[0].cpu().numpy(), od2)):
                    why.append('exhaustive scan vs oracle')
            if nr >= 4:                                                                     # the bound form: lists at a random reach against the scan of the same vertices
                reach = int(rs.choice([0, 20, 140, 1000]))
                _lib.set_option('lbs_reach_mm', reach)
                sw = rs.rand(nr, 24).astype(np.float32)
                sb = SmplUtil(sw)
                sb.set_cano_smpl_vertices(rt[0])
                got = sb.calculate_lbs(qt)
                _lib.set_option('knn_search', 3)
                want = sb._lbs(qt, sb.cano_smpl_vertices)
                _lib.set_option('knn_search', 0)
                if not torch.equal(got, want):
                    why.append(f'bound LBS, reach {reach}')
            nq_total += nq
            if why:
                bad += 1
                print(f'case {k}: nr {nr} ({rk}) nq {nq} ({qk}) K {K}: MISMATCH in {why}')
    finally:
        _lib.set_option('knn_search', 0)
        _lib.set_option('lbs_reach_mm', 140)
    print(f'{cases} cases, {nq_total} queries, {bad} mismatches, {time.time() - t0:.0f} s')
    return 1 if bad else 0


if __name__ == '__main__':
    raise SystemExit(main())

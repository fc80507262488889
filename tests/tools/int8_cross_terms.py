"""Could the two cross terms of the split-fp16 product run on the I8 matrix pipe?  (VERDICT round 2, next #2.)

The fused query evaluates every product as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on three fp16 MFMA passes.  gfx950's I8 MFMA has twice the fp16
rate, so if BOTH operands of the two cross terms could be quantised to int8 -- weights under a per-row power-of-two scale fixed at pack time,
activations under a per-point power-of-two scale found in the epilogue -- a k-step would cost 1 + 1/2 + 1/2 = 2 fp16-equivalents instead of 3
(and with s_wlo = 2^-11 s_whi, s_xlo = 2^-11 s_xhi the two cross terms share one i32 accumulator and one K-concatenated 32x32x32 instruction).

This tool answers the numerical half on the CPU: the fp64 oracle with EVERY affine layer's product replaced by a model of the candidate
arithmetic (operand rounding only; accumulation stays fp64, so the numbers are lower bounds of what a kernel would show), on the golden network
and the three other seeds / gains of tests/test_gpu_query.py.  Output: occupancy and offset L_inf against the unperturbed fp64 oracle.

    python tests/tools/int8_cross_terms.py [n_points]          (CPU, ~1 min)
"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import synthetic as syn
import golden_inputs as gi
from common import geotex_sd, geotex_shapes
from oracle import avatarcap_oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048


def f16(a):
    with np.errstate(over='ignore'):
        return a.astype(np.float16).astype(np.float64)


def pow2_ceil(a):
    """smallest power of two >= a (a > 0)"""
    return 2.0 ** np.ceil(np.log2(np.maximum(a, 1e-300)))


def q_int(a, amax, bits, pow2=True):
    """symmetric integer quantisation of `a` to `bits` bits (incl. sign) under the scale given by amax (broadcastable); returns the dequantised
    values.  pow2: the scale is the power of two that just covers amax (what a kernel can undo exactly), else amax itself."""
    lim = 2 ** (bits - 1) - 1
    top = pow2_ceil(amax) if pow2 else np.maximum(amax, 1e-300)
    step = top / (lim + (1 if pow2 else 0))          # pow2: grid of 2^bits points over [-top, top)
    return np.clip(np.rint(a / step), -lim - 1, lim) * step


def q_fp8_e4m3(a):
    """round to OCP e4m3 (4 significant bits, normal range only is enough here: values are pre-scaled per row / per point into range)"""
    m, e = np.frexp(a)
    return np.ldexp(np.rint(m * 16.0) / 16.0, e)


def split(a):
    hi = f16(a)
    return hi, f16(a - hi)


def make_hook(kind):
    def hook(inp, W, tag):
        x_hi, x_lo = split(inp)
        w_hi, w_lo = split(W)
        if kind == 'fp16x3':                       # the shipped arithmetic (operand rounding only)
            return x_hi @ w_hi.T + x_lo @ w_hi.T + x_hi @ w_lo.T
        if kind == 'fp16x2':                       # both cross terms dropped: the floor of "what the cross terms are worth"
            return x_hi @ w_hi.T
        rw = np.abs(W).max(1, keepdims=True)       # per output row (pack time)
        rx = np.abs(inp).max(1, keepdims=True)     # per point (epilogue: a max over the lane's 128 values and its partner lane)
        if kind.startswith('i8') or kind.startswith('i'):
            bits = int(kind[1:kind.index('_')])
            mode = kind[kind.index('_') + 1:]
            if mode == 'indep':                    # each of the four operands under the best power-of-two scale of its own
                qwh = q_int(w_hi, rw, bits); qxl = q_int(x_lo, np.abs(x_lo).max(1, keepdims=True), bits)
                qwl = q_int(w_lo, np.abs(w_lo).max(1, keepdims=True), bits); qxh = q_int(x_hi, rx, bits)
            elif mode == 'shared':                 # lo scales tied to the hi scales by 2^-11: one accumulator, one K-concatenated instruction
                qwh = q_int(w_hi, rw, bits); qxh = q_int(x_hi, rx, bits)
                qwl = q_int(w_lo, pow2_ceil(rw) * 2.0 ** -11, bits); qxl = q_int(x_lo, pow2_ceil(rx) * 2.0 ** -11, bits)
            elif mode == 'exact':                  # non-power-of-two scales (max exactly at the top code): the best any int grid can do
                qwh = q_int(w_hi, rw, bits, False); qxl = q_int(x_lo, np.abs(x_lo).max(1, keepdims=True), bits, False)
                qwl = q_int(w_lo, np.abs(w_lo).max(1, keepdims=True), bits, False); qxh = q_int(x_hi, rx, bits, False)
            else:
                raise ValueError(kind)
            return x_hi @ w_hi.T + qxl @ qwh.T + qxh @ qwl.T
        if kind == 'fp8':                          # cross terms on e4m3 operands (no scale search needed: floating point)
            return x_hi @ w_hi.T + q_fp8_e4m3(x_lo) @ q_fp8_e4m3(w_hi).T + q_fp8_e4m3(x_hi) @ q_fp8_e4m3(w_lo).T
        raise ValueError(kind)
    return hook


def plain(inp, W, tag):
    return inp @ W.T


def run(pts, fmap, sd, hook):
    orc._matmul = hook
    try:
        r = orc.occupancy_query(pts, fmap, gi.center(), sd)
    finally:
        orc._matmul = plain
    return r['cano_pts_ov'], r['nonrigid_offset']


KINDS = [('fp16x3', 'shipped: three fp16 passes'),
         ('fp16x2', 'hi*hi only (cross terms dropped)'),
         ('i8_exact', 'int8 cross terms, exact-max scales (bound of any int8 grid)'),
         ('i8_indep', 'int8 cross terms, power-of-two scale per operand'),
         ('i8_shared', 'int8 cross terms, lo scales = 2^-11 hi scales (one K-concatenated MFMA)'),
         ('fp8', 'fp8 e4m3 cross terms'),
         ('i10_indep', '(10-bit integers, for the slope)'),
         ('i12_indep', '(12-bit integers, for the slope)')]


def main():
    fmap = gi.pose_feat_map()
    nets = [('golden', geotex_sd(), gi.query_points(11, N))]
    for seed, gain in ((7, 1.6), (123, 1.0), (2024, 2.2)):
        nets.append((f'seed {seed} gain {gain}', syn.synth_state_dict(geotex_shapes(), seed, gain=gain), gi.query_points(300 + seed, N)))
    base = [run(pts, fmap, sd, plain) for _, sd, pts in nets]
    print('| arithmetic of every layer | MFMA cost per k-step (fp16 = 1) | ' + ' | '.join(f'{n}: occ / offset L_inf' for n, _, _ in nets) + ' |')
    print('|---|---:|' + '---|' * len(nets))
    cost = {'fp16x3': 3, 'fp16x2': 1, 'fp8': 2}
    for kind, label in KINDS:
        cells = []
        for (name, sd, pts), (occ0, off0) in zip(nets, base):
            occ, off = run(pts, fmap, sd, make_hook(kind))
            scale = max(1.0, float(np.abs(occ0).max()))
            cells.append('%.1e / %.1e' % (float(np.abs(occ - occ0).max()) / scale, float(np.abs(off - off0).max())))
        print(f'| {label} | {cost.get(kind, 2)} | ' + ' | '.join(cells) + ' |', flush=True)
    print('\n(occupancy relative to max(1, |occ|max) of the net; %d points per net; operand rounding only, fp64 accumulation)' % N)


if __name__ == '__main__':
    main()

"""The dense 256^3 query under the part's power-managed clock (profiles/r03_power_wall.md).

For each regime: ~4 s of back-to-back launches; per launch the device time (HIP events inside the library), the shader cycles of the same launches
(s_memtime of workgroup 0: avc_timing_read_cycles) -> the clock the chip held; rocm-smi power / sclk sampled while they run.
  real     the bench workload (synthetic network, random pose map)
  zero-w   the same launch with ALL weights and biases zero: identical instruction stream and cycle count, no toggling in the multipliers
  blocks-N the real workload on N of the 256 CUs (avc_set_option mlp_blocks)
"""
import ctypes as C
import subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from avatarcap_amd import _lib, config, synthetic as syn
config.cfg = config.default_cfg()
from avatarcap_amd.network.arch_avatar import GeoTexAvatar, OccupancyNet
from avatarcap_amd.grid import volume_axes

RES = 256


def smi_sample():
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
    pw = sclk = None
    for l in out.splitlines():
        low = l.lower()
        if 'power' in low and '(w)' in low and pw is None:
            try: pw = float(l.split(':')[-1].strip())
            except ValueError: pass
        if 'sclk' in low and '(' in l and sclk is None:
            try: sclk = float(l.split('(')[-1].split('Mhz')[0].split('MHz')[0])
            except ValueError: pass
    return pw, sclk


def run(net, label, seconds=4.0, blocks=0):
    axes = volume_axes(syn.CANO_BOUNDS, (RES,) * 3, 'cuda')
    batch = {'cano_smpl_center': torch.zeros(1, 3, device='cuda')}
    ctx = _lib.ctx(torch.device('cuda', 0))
    _lib.set_option('mlp_blocks', blocks)
    q = OccupancyNet(net)
    q.query_grid(batch, axes, (RES,) * 3); torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi_sample()); time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    _lib.check(_lib.lib().avc_timing_enable(ctx, 1))
    n = max(8, int(seconds / 0.06))
    for _ in range(n):
        q.query_grid(batch, axes, (RES,) * 3)
    torch.cuda.synchronize()
    stop.set(); th.join()
    ms, nl, cyc = C.c_double(), C.c_int64(), C.c_double()
    _lib.check(_lib.lib().avc_timing_read(ctx, 0, C.byref(ms), C.byref(nl), 1))
    _lib.check(_lib.lib().avc_timing_read_cycles(ctx, 0, C.byref(cyc), C.byref(nl)))
    _lib.check(_lib.lib().avc_timing_enable(ctx, 0))
    _lib.set_option('mlp_blocks', 0)
    pw = [s[0] for s in samples[2:] if s[0] is not None]; sc = [s[1] for s in samples[2:] if s[1] is not None]
    mfma = 4728 * (RES ** 3 // 128) * 4 * 32768 / (ms.value * 1e-3) / 1e12
    print(f'| {label} | {n} | {ms.value:.2f} | {cyc.value:.4e} | {cyc.value / ms.value / 1e3:.0f} | {mfma:.0f} | '
          f'{(np.mean(pw) if pw else float("nan")):.0f} | {(np.mean(sc) if sc else float("nan")):.0f} |', flush=True)


def main():
    net = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
    sd = syn.synth_state_dict(syn.module_shapes(net), syn.SEED)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.warping_field.pose_feat_map = torch.randn(1, 64, 256, 256, device='cuda')
    print('| regime | launches | device ms / launch | shader cycles / launch (s_memtime) | clock held, MHz | MFMA issued, TFLOP/s | rocm-smi power, W | rocm-smi sclk, MHz |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    pw, sc = smi_sample()
    print(f'| idle | | | | | | {pw} | {sc} |')
    run(net, 'real workload, 256 CUs')
    for b in (248, 224, 192, 128):
        run(net, f'real workload, {b} CUs', blocks=b)
    zero = GeoTexAvatar(base_weight_volume=np.zeros((2, 2, 2, 24), np.float32)).to('cuda').eval()
    zsd = {k: (np.zeros_like(v) if ('weight' in k or 'bias' in k) and 'running' not in k and 'unet' not in k else v) for k, v in sd.items()}
    zero.load_state_dict({k: torch.from_numpy(v) for k, v in zsd.items()})
    zero.warping_field.pose_feat_map = torch.zeros(1, 64, 256, 256, device='cuda')
    run(zero, 'ALL weights / biases / features zero, 256 CUs')
    run(net, 'real workload again, 256 CUs')


if __name__ == '__main__':
    main()

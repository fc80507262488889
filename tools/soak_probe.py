"""Does anything grow?  `main.py -m test --synthetic` (the asynchronous loop with PLY output) for many frames at a small grid, the device's free memory and the host's
resident set sampled by a watcher thread; prints first / last samples.  A leak of a block per frame shows as a slope."""
import os, sys, threading, time, subprocess, tempfile
import torch
sys.path.insert(0, '.')
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 400
d = tempfile.mkdtemp()
cfg = os.path.join(d, 'cfg.yaml')
open(cfg, 'w').write("training: {training_data_dir: null}\ntesting: {vol_res: [96, 96, 48], recon_net_ckpt: null, net_ckpt: null, net_ckpt_finetuned: null, testing_data_dir: null, output_dir: null}\n"
                     "model: {cano_template: {pos_encoding: 10}, warping_field: {pos_encoding: 0}}\n")
p = subprocess.Popen([sys.executable, 'main.py', '-c', cfg, '-m', 'test', '--synthetic', '--frames', str(frames), '--save-ply', '--no-npz', '--output-dir', os.path.join(d, 'out')],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
samples = []


def watch():
    while p.poll() is None:
        try:
            free, total = torch.cuda.mem_get_info(0)
            rss = int(open(f'/proc/{p.pid}/statm').read().split()[1]) * 4096
            samples.append((time.time(), (total - free) / 1e6, rss / 1e6))
        except Exception:
            pass
        time.sleep(0.5)


th = threading.Thread(target=watch, daemon=True); th.start()
out = p.communicate()[0]
done = [l for l in out.splitlines() if 'frames done' in l]
print(done[-1] if done else out[-800:])
s = samples[len(samples) // 4:]          # past start-up
k = max(1, len(s) // 6)
print('device memory in use (MB) over the steady part:', [round(x[1]) for x in s[::k]])
print('host resident set (MB):', [round(x[2]) for x in s[::k]])
import shutil; shutil.rmtree(d, ignore_errors=True)

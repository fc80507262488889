#!/usr/bin/env python3
"""Per-kernel split of the LAST frame in a rocprofv3 --kernel-trace CSV (steady state: MIOpen's first-call searches and
warm-up launches are excluded).  usage: frame_split.py <kernel_trace.csv> <out.md> <title> [first-kernel-of-a-frame regex]"""
import collections, csv, re, sys


def main():
    src, dst, title = sys.argv[1:4]
    first = sys.argv[4] if len(sys.argv) > 4 else 'avatar_kernel'
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if re.search(first, r['Kernel_Name'])]
    # a frame begins a little before its fused query (U-Net launches): cut at the previous frame's last skinning kernel
    prev_end = max(i for i, r in enumerate(rows[:starts[-1]]) if 'skinning_kernel' in r['Kernel_Name'])
    fr = rows[prev_end + 1:]
    t0, t1 = int(fr[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in fr)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in fr:
        n = re.sub(r'avc::\(anonymous namespace\)::', 'avc::', r['Kernel_Name']).replace('void ', '')
        n = n.split('(')[0][:110]
        agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    tot = sum(v[1] for v in agg.values())
    with open(dst, 'w') as f:
        f.write(f'# {title}\n\nsource: `{src}` (rocprofv3 --kernel-trace), last frame only: wall {(t1 - t0) / 1e6:.2f} ms, '
                f'kernel time {tot:.2f} ms, {len(fr)} launches\n\n| kernel | launches | ms | % |\n|---|---:|---:|---:|\n')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            f.write(f'| `{k}` | {v[0]} | {v[1]:.3f} | {100 * v[1] / tot:.1f} |\n')
    print('wrote', dst)


if __name__ == '__main__':
    main()

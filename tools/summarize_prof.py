#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` run (…_kernel_stats.csv) into a small markdown table
for profiles/ (gpurun_out/ is scratch).  usage: summarize_prof.py <kernel_stats.csv> <out.md> [title]"""
import csv
import re
import sys


def short(name, n=110):
    name = re.sub(r'\s+', ' ', name)
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    with open(dst, 'w') as f:
        f.write(f'# {title}\n\nsource: `{src}` (rocprofv3 --kernel-trace --stats --output-format csv); total kernel time {tot/1e6:.2f} ms\n\n')
        f.write('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n')
        for r in rows[:25]:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                    f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    print('wrote', dst)


if __name__ == '__main__':
    main()

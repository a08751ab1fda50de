#!/usr/bin/env python3
"""Copy the summaries tools/prof_r2.sh left under gpurun_out/prof_r2/ into profiles/r02_*: the leading '#' comment
lines of each committed file are kept (edit them by hand when a statement changes), the body is replaced."""
import os
import sys

SRC = 'gpurun_out/prof_r2'
MAP = {   # committed file: (tail file or None, table file)
    'r02_fullrank_headline_kernel_stats.txt': ('headline_tail.txt', 'headline_kernel_stats.txt'),
    'r02_fullrank_fr512_kernel_stats.txt': ('fr512_tail.txt', 'fr512_kernel_stats.txt'),
    'r02_c3_kernel_stats.txt': ('c3_tail.txt', 'c3_kernel_stats.txt'),
    'r02_c4_kernel_stats.txt': ('c4_tail.txt', 'c4_kernel_stats.txt'),
    'r02_fit_loop_kernel_stats.txt': ('fit_tail.txt', 'fit_kernel_stats.txt'),
    'r02_fullrank_gemm_pmc.txt': (None, 'fr1024_pmc.txt'),
    'r02_fullrank_gemm_hbm.txt': (None, 'fr1024_hbm.txt'),
}


def main():
    for dst, (tail, table) in MAP.items():
        path = os.path.join('profiles', dst)
        header = []
        if os.path.exists(path):
            for line in open(path):
                if not line.startswith('#'):
                    break
                header.append(line)
        body = []
        if tail and os.path.exists(os.path.join(SRC, tail)):
            body += [l for l in open(os.path.join(SRC, tail)) if l.strip()]
        tpath = os.path.join(SRC, table)
        if not os.path.exists(tpath):
            print('missing', tpath, file=sys.stderr)
            continue
        body += list(open(tpath))
        open(path, 'w').write(''.join(header + body))
        print('wrote', path, len(body), 'lines')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Copy the summaries tools/prof_r3.sh left under gpurun_out/prof_r3/ into profiles/r03_*: the leading '#' comment
lines of each committed file are kept (edit them by hand when a statement changes), the body is replaced."""
import os
import sys

SRC = 'gpurun_out/prof_r3'
MAP = {   # committed file: (tail file or None, table file)
    'r03_fullrank_headline_kernel_stats.txt': ('headline_tail.txt', 'headline_kernel_stats.txt'),
    'r03_fullrank_fr512_kernel_stats.txt': ('fr512_tail.txt', 'fr512_kernel_stats.txt'),
    'r03_fullrank_funnel_kernel_stats.txt': ('funnel_tail.txt', 'funnel_kernel_stats.txt'),
    'r03_fullrank_path_deriv_kernel_stats.txt': ('pathderiv_tail.txt', 'pathderiv_kernel_stats.txt'),
    'r03_fullrank_fit_kernel_stats.txt': ('frfit_tail.txt', 'frfit_kernel_stats.txt'),
    'r03_c3_kernel_stats.txt': ('c3_tail.txt', 'c3_kernel_stats.txt'),
    'r03_mvt_ekl_kernel_stats.txt': ('mvtekl_tail.txt', 'mvtekl_kernel_stats.txt'),
    'r03_c4_kernel_stats.txt': ('c4_tail.txt', 'c4_kernel_stats.txt'),
    'r03_fit_loop_kernel_stats.txt': ('fit_tail.txt', 'fit_kernel_stats.txt'),
    'r03_lr8_kernel_stats.txt': ('lr8_tail.txt', 'lr8_kernel_stats.txt'),
    'r03_lr32_kernel_stats.txt': ('lr32_tail.txt', 'lr32_kernel_stats.txt'),
    'r03_lr64_kernel_stats.txt': ('lr64_tail.txt', 'lr64_kernel_stats.txt'),
    'r03_fullrank_gemm_pmc.txt': (None, 'fr1024_pmc.txt'),
    'r03_meanfield_c1_pmc_hbm.txt': (None, 'meanfield_c1_hbm.txt'),
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

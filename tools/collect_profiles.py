#!/usr/bin/env python3
"""Copy the summaries tools/prof_r4.sh left under gpurun_out/prof_r4/ into profiles/r04_*: the leading '#' comment
lines of each committed file are kept (edit them by hand when a statement changes), the body is replaced."""
import os
import sys

SRC = 'gpurun_out/prof_r4'
MAP = {   # committed file: (tail file or None, table file)
    'r04_fullrank_headline_kernel_stats.txt': ('headline_tail.txt', 'headline_kernel_stats.txt'),
    'r04_fullrank_fr512_kernel_stats.txt': ('fr512_tail.txt', 'fr512_kernel_stats.txt'),
    'r04_fullrank_funnel_kernel_stats.txt': ('funnel_tail.txt', 'funnel_kernel_stats.txt'),
    'r04_fullrank_path_deriv_kernel_stats.txt': ('pathderiv_tail.txt', 'pathderiv_kernel_stats.txt'),
    'r04_fullrank_fit_kernel_stats.txt': ('frfit_tail.txt', 'frfit_kernel_stats.txt'),
    'r04_c3_kernel_stats.txt': ('c3_tail.txt', 'c3_kernel_stats.txt'),
    'r04_mvt_ekl_kernel_stats.txt': ('mvtekl_tail.txt', 'mvtekl_kernel_stats.txt'),
    'r04_c4_kernel_stats.txt': ('c4_tail.txt', 'c4_kernel_stats.txt'),
    'r04_fit_loop_kernel_stats.txt': ('fit_tail.txt', 'fit_kernel_stats.txt'),
    'r04_lr8_kernel_stats.txt': ('lr8_tail.txt', 'lr8_kernel_stats.txt'),
    'r04_lr32_kernel_stats.txt': ('lr32_tail.txt', 'lr32_kernel_stats.txt'),
    'r04_lr64_kernel_stats.txt': ('lr64_tail.txt', 'lr64_kernel_stats.txt'),
    'r04_fullrank_gemm_pmc.txt': (None, 'fr1024_pmc.txt'),
    'r04_meanfield_c1_pmc_hbm.txt': (None, 'meanfield_c1_hbm.txt'),
    'r04_alpha_kernel_stats.txt': ('alpha_tail.txt', 'alpha_kernel_stats.txt'),
    'r04_legacy_dev_kernel_stats.txt': ('legacydev_tail.txt', 'legacydev_kernel_stats.txt'),
    'r04_api_call_kernel_stats.txt': ('apicall_tail.txt', 'apicall_kernel_stats.txt'),
    'r04_psis_kernel_stats.txt': ('psis_tail.txt', 'psis_kernel_stats.txt'),
    'r04_dis_bisect_kernel_stats.txt': ('disbisect_tail.txt', 'disbisect_kernel_stats.txt'),
}


def main():
    for dst, (tail, table) in MAP.items():
        path = os.path.join('profiles', dst)
        header = []
        if not os.path.exists(path) and os.path.exists(path.replace('r04_', 'r03_')):      # start from last round's header
            for line in open(path.replace('r04_', 'r03_')):
                if not line.startswith('#'):
                    break
                header.append(line.replace('round 3', 'round 4').replace('Round 3', 'Round 4'))
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

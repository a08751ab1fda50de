#!/usr/bin/env python3
"""Generate viabel_amd/csrc/vb_mt_jump.h: jump-ahead polynomials of MT19937 for the device restatement of numpy's
legacy normal stream (vb_legacy_dev.hip).

MT19937's output words satisfy a linear recurrence over GF(2) whose characteristic polynomial phi (degree 19937) is
found here with Berlekamp-Massey from one output bit stream.  For a jump of J words, g_J(x) = x^J mod phi(x) gives
    u[n + J] = XOR over { i : coefficient i of g_J is 1 } of u[n + i]          (all 32 bits, n >= 0)
for the sequence u of untempered state words from the first refreshed block onward (Haramoto, Matsumoto, Nishimura,
Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008) -- so the state 624 * B * 2^k
words ahead of a known block is a correlation of g with 20 560 words generated from that block, and 4^r streams become
4^(r+1) per round (three jumps per known stream).  The table holds g for J = 624 * BLOCKS_PER_STREAM * a * 4^r,
r = 0 .. R - 1, a = 1, 2, 3 (row 3 r + a - 1).

Integer work: checked here against numpy's own generator (RandomState.random_sample consumes two words per draw) for
the first rounds and by the squaring identity for all of them; tests/test_legacy_rng_cpu.py re-checks the committed
table.  Run in the build container:  python tools/make_mt_jump.py
"""
import os
import sys

import numpy as np

N, M, DEG = 624, 397, 19937
BLOCKS_PER_STREAM = 256
R = 5                     # rounds of the radix-4 ladder: up to 4^5 = 1024 streams
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'viabel_amd', 'csrc', 'vb_mt_jump.h')


def refresh(key):
    key = [int(x) for x in key]
    for k in range(N):
        y = (key[k] & 0x80000000) | (key[(k + 1) % N] & 0x7fffffff)
        key[k] = key[(k + M) % N] ^ (y >> 1) ^ ((-(y & 1)) & 0x9908b0df)
    return key


def seq_words(block, nwords):
    """u[0 .. nwords): `block` itself followed by its successors."""
    out, key = list(int(x) for x in block), list(block)
    while len(out) < nwords:
        key = refresh(key)
        out += key
    return out[:nwords]


def revbits(x, n):
    return int(bin(x)[2:].zfill(n)[::-1], 2)


def berlekamp_massey(bits):
    n = len(bits)
    sbits = 0
    for i, b in enumerate(bits):
        sbits |= b << i
    C, B, L, m = 1, 1, 0, 1
    for i in range(n):
        if i >= L:
            window = (sbits >> (i - L)) & ((1 << (L + 1)) - 1)
            d = bin(window & revbits(C, L + 1)).count('1') & 1
        else:
            d = 0
            for j in range(L + 1):
                if (C >> j) & 1 and i - j >= 0:
                    d ^= bits[i - j]
        if d:
            T = C
            C ^= B << m
            if 2 * L <= i:
                L, B, m = i + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def clmul(a, b):
    """Carry-less product of two GF(2) polynomials held as ints: coefficients spread into 16-bit slots (a product
    coefficient counts at most 19 937 < 65 536 terms), one big-integer multiplication, parities taken back."""
    def spread(x):
        nb = max(1, x.bit_length())
        bits = np.frombuffer(x.to_bytes((nb + 7) // 8, 'little'), dtype=np.uint8)
        bits = np.unpackbits(bits, bitorder='little')[:nb].astype('<u2')
        return int.from_bytes(bits.tobytes(), 'little')
    p = spread(a) * spread(b)
    raw = np.frombuffer(p.to_bytes((p.bit_length() + 15) // 16 * 2 or 2, 'little'), dtype='<u2')
    bits = (raw & 1).astype(np.uint8)
    return int.from_bytes(np.packbits(bits, bitorder='little').tobytes(), 'little')


def polymod(a, phi):
    dphi = phi.bit_length() - 1
    while a.bit_length() - 1 >= dphi:
        a ^= phi << (a.bit_length() - 1 - dphi)
    return a


def polypow_x(e, phi):
    """x^e mod phi by square and multiply."""
    result, base = 1, 2      # 1, x
    while e:
        if e & 1:
            result = polymod(clmul(result, base), phi)
        base = polymod(clmul(base, base), phi)
        e >>= 1
    return result


def jump_by_correlation(block, g):
    """Block 624 * J / 624 ... i.e. the 624 words J ahead of `block`, J the jump g was made for."""
    u = seq_words(block, DEG + N)
    idx = [i for i in range(DEG) if (g >> i) & 1]
    ua = np.array(u, dtype=np.uint32)
    ia = np.array(idx)
    return [int(np.bitwise_xor.reduce(ua[ia + j])) for j in range(N)]


def main():
    rs = np.random.RandomState(20260412)
    key0 = rs.get_state()[1]
    block1 = refresh(key0)
    w = seq_words(block1, 2 * DEG + 64)
    C, L = berlekamp_massey([(x >> 31) & 1 for x in w])
    assert L == DEG and C.bit_length() - 1 == DEG, (L, C.bit_length())
    phi = revbits(C, DEG + 1)                       # characteristic polynomial: x^L C(1 / x)
    assert phi.bit_length() - 1 == DEG and phi & 1
    # every bit lane satisfies the same recurrence (spot check: bit 0 and bit 17)
    for lane in (0, 17):
        s = [(x >> lane) & 1 for x in w]
        for n in (0, 5, 1000):
            acc = 0
            for j in range(1, DEG + 1):
                if (C >> j) & 1:
                    acc ^= s[n + DEG - j]
            assert acc == s[n + DEG], lane
    W = N * BLOCKS_PER_STREAM
    polys, jumps = [], []                          # row 3 r + a - 1: jump of a 4^r streams
    base = polypow_x(W, phi)
    for r in range(R):
        two = polymod(clmul(base, base), phi)
        three = polymod(clmul(two, base), phi)
        polys += [base, two, three]
        jumps += [BLOCKS_PER_STREAM * a * 4 ** r for a in (1, 2, 3)]
        base = polymod(clmul(two, two), phi)       # x^(W 4^(r+1))
    # against numpy: after consuming 624 m words from a fresh seed the state is block m with pos = 624
    for row in (0, 1, 2, 3, 5):
        m = jumps[row]
        rs2 = np.random.RandomState(99 + row)
        b1 = refresh(rs2.get_state()[1])
        rs2.random_sample(N * m // 2)               # two words per draw
        st = rs2.get_state()
        assert st[2] == N
        want = refresh(st[1])                       # block m + 1 = the block 624 m words after block 1
        got = jump_by_correlation(b1, polys[row])
        assert got == [int(x) for x in want], 'jump polynomial %d fails against numpy' % row
        print('jump 624 x %d words: correlation equals numpy\'s state' % m)
    with open(OUT, 'w') as f:
        f.write('// GENERATED by tools/make_mt_jump.py -- do not edit.  Jump-ahead polynomials of MT19937:\n'
                '// kMtJump[3 r + a - 1] = x^(624 * kMtBlocksPerStream * a * 4^r) mod phi(x), a = 1, 2, 3, phi the characteristic\n'
                '// polynomial of the recurrence (degree 19937), coefficient i in bit (i & 31) of word (i >> 5).\n#pragma once\n'
                '#include <cstdint>\nnamespace vb {\nconstexpr int kMtBlocksPerStream = %d;\nconstexpr int kMtJumpRounds = %d;\n'
                'constexpr int kMtJumpPolys = %d;\nconstexpr int kMtJumpWords = %d;\n' % (BLOCKS_PER_STREAM, R, 3 * R, N))
        f.write('static const uint32_t kMtJump[kMtJumpPolys][kMtJumpWords] = {\n')
        for g in polys:
            words = [(g >> (32 * i)) & 0xffffffff for i in range(N)]
            f.write('{' + ','.join('0x%08xu' % x for x in words) + '},\n')
        f.write('};\n}  // namespace vb\n')
    print('wrote %s (%d polynomials)' % (OUT, 3 * R))


if __name__ == '__main__':
    main()

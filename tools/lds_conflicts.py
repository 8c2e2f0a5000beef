#!/usr/bin/env python3
"""
Bank-conflict model of the corr_H B-operand read (16x16x4 MFMA, ds_read_b32: 32 banks, conflicts per 32-lane half).
Lane (j, kq) of column tile t reads word  -(a*XST + b) + kq  with column J = 16 t + j = a*Ax + b.
Prints the average LDS cycles per half for a row stride XST == r (mod 32); r == Ax + 1 is conflict-free.
"""
import sys


def cost(Ax, Ay, XST):
    J = Ax * Ay
    tot = n = 0
    for t in range((J + 15) // 16):
        for half in (0, 1):
            banks = {}
            for j in range(16):
                col = t * 16 + j
                if col >= J:
                    continue
                a, b = divmod(col, Ax)
                for kq in (2 * half, 2 * half + 1):
                    w = -(a * XST + b) + kq + 10 ** 6
                    banks.setdefault(w % 32, set()).add(w)
            tot += max(len(v) for v in banks.values())
            n += 1
    return tot / n


if __name__ == '__main__':
    Ax, Ay = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 12)
    for r in range(32):
        print(f'XST == {r:2d} (mod 32): {cost(Ax, Ay, 96 + r):.2f} LDS cycles per half-wave')

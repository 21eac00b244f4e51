#!/usr/bin/env python3
"""LDS bank-conflict model of gfx950 (MI355X_MICROARCH.md, LDS table) for the access patterns of the block kernels.

A wave64 DS instruction is served in fixed lane groups, one LDS cycle per group when conflict-free; within a group each
extra distinct dword address on a busy bank adds one cycle.  ds_read_b32 / ds_write_b32: 2 groups of 32 lanes, 32 banks;
ds_read_b64: 2 x 32, 64 banks; ds_read_b128: 4 groups of 16 (the guide's lane sets), 64 banks.

    python tools/lds_banks.py            # the table DESIGN.md quotes
"""
import sys

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    [x + 32 for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))],
    [x + 32 for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))],
]


def cycles(addr_of_lane, width=1, lanes=range(64)):
    """addr_of_lane(lane) -> first dword address (or None: lane inactive). Returns (cycles, ideal cycles)."""
    if width == 1:
        groups, nb = [list(range(0, 32)), list(range(32, 64))], 32
    elif width == 2:
        groups, nb = [list(range(0, 32)), list(range(32, 64))], 64
    else:
        groups, nb = B128_GROUPS, 64
    tot = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            if l not in lanes:
                continue
            a = addr_of_lane(l)
            if a is None:
                continue
            for w in range(width):
                per_bank.setdefault((a + w) % nb, set()).add(a + w)
        tot += max([len(s) for s in per_bank.values()] + [1])
    return tot, len(groups)


def valu_window(C, P, L, nthreads, row=0):
    """(channel, chunk) register-window read of one row offset: worst / mean cycles over the waves of the workgroup."""
    res = []
    for wave in range(nthreads // 64):
        def addr(l, wave=wave):
            tid = wave * 64 + l
            c, ch = tid % C, tid // C
            nch = nthreads // C
            if ch >= nch:
                ch = nch - 1
            return (ch * L + row) * P + c
        res.append(cycles(addr)[0])
    return sum(res) / len(res), 2


def mfma_rows(P, rowmap, col0=0):
    """operand whose lanes (r16 -> 16 consecutive columns, g -> a row): A of dW (u), B of dW (dp), B of du (W^T)."""
    return cycles(lambda l: rowmap(l >> 4) * P + col0 + (l & 15))


def mfma_cols_b32(P, kmap=lambda g: g):
    """operand whose lanes (r16 -> a row, g -> a column): A of du / of the forward pointwise, one k-step per read."""
    return cycles(lambda l: (l & 15) * P + kmap(l >> 4))


def mfma_cols_b128(P):
    """the same operand read as one float4 per lane (columns 4g..4g+3 = the lane's k of four k-steps)."""
    return cycles(lambda l: (l & 15) * P + 4 * (l >> 4), width=4)


def commit_b128(C, P, nthreads, wave=0):
    Q = C // 4
    def addr(l):
        i = wave * 64 + l
        return (i // Q) * P + (i % Q) * 4
    return cycles(addr, width=4)


def main():
    print("pattern                                   cycles/ideal")
    for C, nth, L in ((48, 256, 13), (48, 512, 7), (48, 384, 8), (32, 256, 8), (64, 256, 16), (64, 512, 8)):
        for P in sorted({C, C + 2, C + 4, C + 8, C + 16}):
            v, ideal = valu_window(C, P, L, nth)
            r1 = mfma_rows(P, lambda g: g)[0]
            r4 = mfma_rows(P, lambda g: 4 * g)[0]
            r8 = mfma_rows(P, lambda g: 8 * g)[0]
            c32 = mfma_cols_b32(P)[0]
            c128 = mfma_cols_b128(P)[0] if P % 4 == 0 else -1
            print(f"C={C} threads={nth} L={L} P={P}: window {v:.2f}/2  rows d=1 {r1}/2 d=4 {r4}/2 d=8 {r8}/2  cols b32 {c32}/2  cols b128 {c128}/4")
    return 0


if __name__ == "__main__":
    sys.exit(main())

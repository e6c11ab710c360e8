"""Symbolic check of the split-f16 GEMM kernels' LDS staging / fragment / epilogue index formulas
(csrc/conv_igemm.hip::conv_split_kernel and csrc/gemm_dma.hip::gemm_dma_kernel), transcribed into numpy for every tile
configuration the launchers instantiate.  It cannot prove the HIP source (a GPU run does that) but it pins the
formulas of the generalised kernel (wave grid WM x WN, K-slice width RB) to the layout of the GPU-validated
default: every LDS slot a fragment load reads holds exactly the (row, k, hi|lo) element the 32x32x16 MFMA
operand layout expects, and the epilogue writes every output element of the tile exactly once."""
import itertools

import numpy as np
import pytest

CONFIGS = [  # BM, BN, WM, WN, RB        (NW = WM * WN waves)
    (128, 128, 2, 4, 128),   # conv_split_kernel<128,128,8,...>
    (128, 64, 2, 2, 128),    # the 4-wave tiles
    (64, 64, 2, 2, 128),
]


def stage(BM, BN, WM, WN, RB):
    NW = WM * WN
    ROWB, VPR = RB + 16, RB // 16
    RPP = NW * 64 // VPR
    A_IT, B_IT = BM // RPP, BN // RPP
    assert BM % RPP == 0 and BN % RPP == 0
    lds = np.full(((BM + BN) * ROWB // 2, 4), -1, dtype=np.int64)      # one entry per f16 slot: (operand, row, k, part)
    for tid in range(NW * 64):
        vec, r0 = tid % VPR, tid // VPR
        a_wr = (vec >> 2) * 64 + (vec & 3) * 8
        for it in range(A_IT):
            row = r0 + it * RPP
            for e in range(4):                                          # split_f16x4: 4 consecutive k of this thread's u32x4
                k = vec * 4 + e
                lds[(row * ROWB + a_wr) // 2 + e] = (0, row, k, 0)       # hi halves
                lds[(row * ROWB + a_wr + 32) // 2 + e] = (0, row, k, 1)  # lo halves
        for it in range(B_IT):
            row = r0 + it * RPP
            for h in range(8):                                          # 16 bytes = 8 halves of the packed weight row
                byte = vec * 16 + h * 2                                  # [K/16][16 hi | 16 lo]
                blk, inb = byte // 64, byte % 64
                lds[((BM + row) * ROWB + vec * 16) // 2 + h] = (1, row, blk * 16 + (inb % 32) // 2, inb // 32)
    return lds, ROWB


@pytest.mark.parametrize("BM,BN,WM,WN,RB", CONFIGS)
def test_fragments_read_what_the_mfma_layout_expects(BM, BN, WM, WN, RB):
    lds, ROWB = stage(BM, BN, WM, WN, RB)
    TM, TN, KB16 = BM // (32 * WM), BN // (32 * WN), RB // 64
    assert TM >= 1 and TN >= 1 and BM // WM == 32 * TM and BN // WN == 32 * TN
    for wave, lane in itertools.product(range(WM * WN), range(64)):
        wm, wn = wave // WN, wave % WN
        a_rd = (wm * (BM // WM) + (lane & 31)) * ROWB + (lane >> 5) * 16
        b_rd = BM * ROWB + (wn * (BN // WN) + (lane & 31)) * ROWB + (lane >> 5) * 16
        for j16 in range(KB16):
            for i in range(TM):
                for part in (0, 1):
                    got = lds[(a_rd + i * 32 * ROWB + j16 * 64 + 32 * part) // 2:][:8]
                    exp_row = wm * (BM // WM) + i * 32 + (lane & 31)
                    exp = [(0, exp_row, j16 * 16 + (lane >> 5) * 8 + h, part) for h in range(8)]
                    assert got.tolist() == [list(x) for x in exp]
            for j in range(TN):
                for part in (0, 1):
                    got = lds[(b_rd + j * 32 * ROWB + j16 * 64 + 32 * part) // 2:][:8]
                    exp_row = wn * (BN // WN) + j * 32 + (lane & 31)
                    exp = [(1, exp_row, j16 * 16 + (lane >> 5) * 8 + h, part) for h in range(8)]
                    assert got.tolist() == [list(x) for x in exp]


@pytest.mark.parametrize("BM,BN,WM,WN,RB", CONFIGS)
def test_epilogue_covers_the_tile_once_and_reads_are_bank_conflict_free(BM, BN, WM, WN, RB):
    TM, TN = BM // (32 * WM), BN // (32 * WN)
    hits = np.zeros((BM, BN), dtype=np.int32)
    for wave, lane in itertools.product(range(WM * WN), range(64)):
        wm, wn = wave // WN, wave % WN
        for j, i, e in itertools.product(range(TN), range(TM), range(16)):
            n = wn * (BN // WN) + j * 32 + (lane & 31)
            m = wm * (BM // WM) + i * 32 + 4 * (lane >> 5) + (e & 3) + 8 * (e >> 2)      # 32x32 C/D layout
            hits[m, n] += 1
    assert (hits == 1).all()
    # ds_read_b128 is served in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS table); 64 banks of 4 bytes:
    ROWB = RB + 16
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for g in groups:
        banks = []
        for lane in g:
            addr = (lane & 31) * ROWB + (lane >> 5) * 16
            banks += [((addr // 4) + d) % 64 for d in range(4)]
        assert len(set(banks)) == 64, (RB, sorted(banks))


DMA_CONFIGS = [  # BM, BN, WM, WN   (csrc/gemm_dma.hip::launch_tile instantiations)
    (256, 256, 2, 4),
    (256, 128, 4, 2),
    (128, 128, 2, 2),
]
GROUPS16 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS16 += [[l + 32 for l in g] for g in GROUPS16]


@pytest.mark.parametrize("BM,BN,WM,WN", DMA_CONFIGS)
def test_gemm_dma_stage_layout(BM, BN, WM, WN):
    """gemm_dma_kernel: both operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds: LDS destination = wave-uniform
    base + lane * 16), rows are unpadded 128 B and the 16-byte chunk index is XORed with (row >> 1) & 7 on the per-lane
    SOURCE address; the fragment reads apply the same XOR.  Checks: the DMA pieces tile the stage exactly once, every
    fragment read returns the (row, k, hi|lo) elements the 32x32x16 MFMA operand layout expects, ds_read_b128 lane
    groups are bank-conflict free, and the transposed epilogue (lane = token, 4 consecutive channels per accumulator
    quad) covers the tile exactly once."""
    NW = WM * WN
    A_DMA, B_DMA = BM // (8 * NW), BN // (8 * NW)
    TM, TN = BM // (32 * WM), BN // (32 * WN)
    assert A_DMA >= 1 and B_DMA >= 1 and TM % 2 == 0 and TN >= 1
    lds = np.full(((BM + BN) * 128 // 2, 4), -1, dtype=np.int64)        # per f16 slot: (operand, row, k, part)
    for wave, lane in itertools.product(range(NW), range(64)):
        rsub, slot = lane >> 3, lane & 7
        for op, ndma, base in ((0, A_DMA, 0), (1, B_DMA, BM * 128)):
            for i in range(ndma):
                rl = (wave * ndma + i) * 8 + rsub                       # tile-local row this lane fetches
                chunk = slot ^ ((rl >> 1) & 7)                          # SOURCE chunk of the 128-byte K slice
                dst = base + (wave * ndma + i) * 1024 + lane * 16       # hardware: M0 base + lane * 16
                for h in range(8):
                    byte = chunk * 16 + h * 2                           # row = [K/16][16 hi | 16 lo]
                    blk, inb = byte // 64, byte % 64
                    assert lds[dst // 2 + h, 0] == -1
                    lds[dst // 2 + h] = (op, rl, blk * 16 + (inb % 32) // 2, inb // 32)
    assert (lds[:, 0] >= 0).all()
    for wave, lane in itertools.product(range(NW), range(64)):
        wm, wn = wave // WN, wave % WN
        swz, hsel = (lane >> 1) & 7, lane >> 5
        for g, part in itertools.product(range(2), range(2)):
            c = ((g * 4 + part * 2 + hsel) ^ swz) * 16
            offA = (wm * (BM // WM) + (lane & 31)) * 128 + c
            offW = BM * 128 + (wn * (BN // WN) + (lane & 31)) * 128 + c
            for i in range(TM):
                got = lds[(offA + i * 4096) // 2:][:8].tolist()
                row = wm * (BM // WM) + i * 32 + (lane & 31)
                assert got == [[0, row, g * 16 + hsel * 8 + h, part] for h in range(8)], (wave, lane, g, part, i)
            for j in range(TN):
                got = lds[(offW + j * 4096) // 2:][:8].tolist()
                row = wn * (BN // WN) + j * 32 + (lane & 31)
                assert got == [[1, row, g * 16 + hsel * 8 + h, part] for h in range(8)], (wave, lane, g, part, j)
    for grp, gp in itertools.product(GROUPS16, range(4)):
        banks = []
        for lane in grp:
            addr = (lane & 31) * 128 + ((((gp >> 1) * 4 + (gp & 1) * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16)
            banks += [((addr // 4) + d) % 64 for d in range(4)]
        assert len(set(banks)) == 64, sorted(banks)
    hits = np.zeros((BM, BN), dtype=np.int32)
    for wave, lane in itertools.product(range(NW), range(64)):
        wm, wn = wave // WN, wave % WN
        for i, j, q, c in itertools.product(range(TM), range(TN), range(4), range(4)):
            m = wm * (BM // WM) + i * 32 + (lane & 31)                  # D^T: column = token
            e = q * 4 + c                                               # accumulator register -> row of D^T = channel
            n = wn * (BN // WN) + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
            assert n == wn * (BN // WN) + 4 * (lane >> 5) + j * 32 + q * 8 + c
            hits[m, n] += 1
    assert (hits == 1).all()


def test_split_row_byte_offsets():
    """omni_split_off(c): channel c (c % 4 == 0) of a format-B row -> byte offset of its hi halves (lo = +32)."""
    seen = set()
    for c in range(0, 256, 4):
        off = (c >> 4) * 64 + (c & 15) * 2
        for part, h in itertools.product(range(2), range(4)):
            b = off + 32 * part + 2 * h
            assert b not in seen and b // 64 == (c + h) // 16 and (b % 64) // 32 == part and (b % 32) // 2 == (c + h) % 16
            seen.add(b)
    assert len(seen) == 512 and max(seen) == 1022                        # 256 channels x (hi + lo) halves = 1024 bytes

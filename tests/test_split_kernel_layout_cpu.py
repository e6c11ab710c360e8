"""Symbolic check of the split-f16 GEMM kernels' LDS staging / fragment / epilogue index formulas
(csrc/conv_igemm.hip::conv_split_kernel and conv_split2_kernel), transcribed into numpy for every tile
configuration the launchers instantiate.  It cannot prove the HIP source (a GPU run does that) but it pins the
formulas of the generalised kernel (wave grid WM x WN, K-slice width RB) to the layout of the GPU-validated
default: every LDS slot a fragment load reads holds exactly the (row, k, hi|lo) element the 32x32x16 MFMA
operand layout expects, and the epilogue writes every output element of the tile exactly once."""
import itertools

import numpy as np
import pytest

CONFIGS = [  # BM, BN, WM, WN, RB        (NW = WM * WN waves)
    (128, 128, 2, 4, 128),   # default (variant 2): conv_split_kernel<128,128,8,...>
    (128, 128, 2, 2, 128),   # variant 0 and the 4-wave fallbacks
    (128, 64, 2, 2, 128),
    (64, 64, 2, 2, 128),
    (256, 128, 4, 2, 128),   # variant 4
    (128, 128, 2, 4, 256),   # variant 5: 64-wide K slices
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


def test_lds_dma_weight_layout_variant6():
    """conv_split2_kernel<..., BDMA = true> (OMNI_SPLIT_VARIANT=6): weight slices arrive by global_load_lds_dwordx4,
    whose LDS destination is wave-uniform base + lane*16 — rows are unpadded (128 B) and an XOR swizzle on the per-lane
    SOURCE chunk, undone by the same XOR on the fragment read address, keeps ds_read_b128 conflict-free."""
    BM, BN, WM, WN, RB = 128, 128, 2, 4, 128
    NW, VPR = WM * WN, 8
    RPP, B_IT = NW * 8, BN // (NW * 8)
    lds = np.full((BN * RB // 2, 3), -1, dtype=np.int64)              # weight region only: (row, k, part) per f16 slot
    for wave, lane, it in itertools.product(range(NW), range(64), range(B_IT)):
        tid = wave * 64 + lane
        vec, r0 = tid % VPR, tid // VPR
        row = r0 + it * RPP                                            # global weight row of this lane
        chunk = vec ^ ((r0 >> 1) & 7)                                  # source chunk (16 B) of the 128-byte slice
        dst = (it * RPP + wave * 8) * RB + lane * 16                   # hardware: M0 base + lane * 16
        for h in range(8):
            byte = chunk * 16 + h * 2                                  # within [16 hi | 16 lo] x 2 blocks
            blk, inb = byte // 64, byte % 64
            lds[dst // 2 + h] = (row, blk * 16 + (inb % 32) // 2, inb // 32)
    assert (lds[:, 0] >= 0).all()                                      # the 16 DMA instructions fill the region exactly
    TN = BN // (32 * WN)
    for wave, lane in itertools.product(range(NW), range(64)):
        wn = wave % WN
        swz = (lane >> 1) & 7
        for j, j16, part in itertools.product(range(TN), range(2), range(2)):
            R = wn * (BN // WN) + j * 32 + (lane & 31)
            addr = R * RB + (((j16 * 4 + part * 2 + (lane >> 5)) ^ swz) * 16)
            got = lds[addr // 2:][:8].tolist()
            assert got == [[R, j16 * 16 + (lane >> 5) * 8 + h, part] for h in range(8)], (wave, lane, j16, part)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for g, c in itertools.product(groups, range(4)):                    # c = j16*2 + part: the chunk pair a read touches
        banks = []
        for lane in g:
            addr = (lane & 31) * RB + (((c // 2 * 4 + c % 2 * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) * 16)
            banks += [((addr // 4) + d) % 64 for d in range(4)]
        assert len(set(banks)) == 64, sorted(banks)

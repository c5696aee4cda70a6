"""The LDS images of the K-major GEMM operands against the bank model the counters gave (round 6, profiles/r06_gemm_pmc.json): gfx950's
LDS has 64 banks of 4 bytes; a ds_read_b128 is served in FOUR passes of 16 lanes, and SQ_LDS_BANK_CONFLICT showed which lanes share a
pass — those with equal (lane & 15) >> 1 & 3, i.e. rows {2p, 2p + 1, 2p + 8, 2p + 9} of all four 16-lane groups (the first 64-byte-row
image of gemm128w.hip, conflict-free for passes of equal lane >> 4, cost 4 extra cycles on every read). The formulas below restate the
kernels' address maps (csrc/gemm256p_kernel.h `kmaj_lane`, csrc/gemm128w.hip `Frags<true>::init`); the test pins that both are
conflict-free under BOTH groupings and that the plain-order image is not — a layout change has to pass here before it costs a GPU visit."""


def cycles(addr_of_lane, passes):
    total = 0
    for lanes in passes:
        load = {}
        for lane in lanes:
            a = addr_of_lane(lane)
            for b in range(4):  # 16 bytes = four consecutive banks
                bank = (a // 4 + b) % 64
                load[bank] = load.get(bank, 0) + 1
        total += max(load.values())
    return total


SAME_GROUP = [list(range(16 * p, 16 * p + 16)) for p in range(4)]
MEASURED = [[lane for lane in range(64) if ((lane & 15) >> 1) % 4 == p] for p in range(4)]


def kmajor_128(lane):  # gemm256p_kernel.h: 128-byte rows, 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
    l15, g4 = lane & 15, lane >> 4
    return l15 * 128 + ((g4 ^ (l15 >> 1)) & 7) * 16


def kmajor_64_plain(lane):  # gemm128w.hip's first image: 64-byte rows in order, chunk c of row r at c ^ ((r >> 2) & 3)
    l15, g4 = lane & 15, lane >> 4
    return l15 * 64 + ((g4 ^ (l15 >> 2)) & 3) * 16


def slot_of_row(l15):
    return (l15 & 1) | (((l15 >> 3) & 1) << 1) | (((l15 >> 1) & 3) << 2)


def kmajor_64_slots(lane):  # gemm128w.hip as shipped: row l15 in slot b0 | b3 << 1 | (b2 b1) << 2, chunk c at c ^ (slot >> 2)
    l15, g4 = lane & 15, lane >> 4
    s = slot_of_row(l15)
    return s * 64 + ((g4 ^ (s >> 2)) & 3) * 16


def test_shipped_kmajor_images_are_conflict_free_under_both_pass_groupings():
    for image in (kmajor_128, kmajor_64_slots):
        assert cycles(image, SAME_GROUP) == 4
        assert cycles(image, MEASURED) == 4


def test_plain_order_64_byte_rows_conflict_under_the_measured_grouping():
    assert cycles(kmajor_64_plain, SAME_GROUP) == 4   # why it looked right on paper
    assert cycles(kmajor_64_plain, MEASURED) == 8     # what the counters showed: + 4 cycles on every ds_read_b128


def test_slot_map_is_a_permutation_and_matches_the_dma_side():
    assert sorted(slot_of_row(r) for r in range(16)) == list(range(16))
    # piece_offs<true>: the lane that writes slot s fetches row (s & 1) | ((s >> 2) & 3) << 1 | ((s >> 1) & 1) << 3 — the inverse map
    for s in range(16):
        row = (s & 1) | (((s >> 2) & 3) << 1) | (((s >> 1) & 1) << 3)
        assert slot_of_row(row) == s

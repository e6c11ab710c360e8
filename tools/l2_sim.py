"""LRU simulation of ONE XCD's 4 MiB L2 under the split-GEMM kernel's operand stream, for the round-1 tile order
(row-block mapping, xcd_n = 1) and the N-partition order (xcd_n > 1).  No GPU: a model, used to sanity-check the
residency argument of DESIGN.md §5a before the change could be measured.

64 workgroup slots per XCD (32 CUs x 2 blocks); a slot walks the K slices of its tile (per slice: a 128 x 128-byte
activation slice and a 128 x 128-byte weight slice = 256 cache lines), then writes its 128x128 f32 output tile
(write-allocate) and takes the XCD's next tile in dispatch order.  `stagger`: slot j starts j*ktiles/64 slices late
(steady state after the first dispatch wave has drifted apart); 0 = perfect lockstep."""
import argparse
from collections import OrderedDict

LINE = 128


class L2:
    def __init__(self, size=4 << 20, ways=16):
        self.ways = ways
        self.nsets = size // LINE // ways
        self.sets = [OrderedDict() for _ in range(self.nsets)]
        self.miss = {"A": 0, "W": 0, "Y": 0}
        self.acc = {"A": 0, "W": 0, "Y": 0}

    def access(self, kind, line):
        h = (line * 0x9E3779B1) & 0xFFFFFFFF            # address hashing over channels / sets
        st = self.sets[(h >> 7) % self.nsets]
        self.acc[kind] += 1
        if line in st:
            st.move_to_end(line)
            return
        self.miss[kind] += 1
        st[line] = kind
        if len(st) > self.ways:
            st.popitem(last=False)


def tile_seq(mapping, ntiles, xcd_n, x=0):
    """Tiles of XCD x in dispatch order (infinite generator): mirrors conv_igemm.hip::tile_of_block."""
    s = 0
    gn = ntiles // xcd_n
    mper = 8 // xcd_n
    while True:
        ml, nl = divmod(s, gn)
        yield ml * mper + x // xcd_n, (x % xcd_n) * gn + nl
        s += 1


def simulate(N, K, xcd_n, tiles_per_slot=6, stagger=True, slots=64):
    ntiles, ktiles = N // 128, K // 32
    cache = L2()
    seq = tile_seq("x", ntiles, xcd_n)
    A_BASE, W_BASE, Y_BASE = 0, 1 << 36, 1 << 37
    state = []
    for j in range(slots):
        state.append({"tile": next(seq), "k": -(j * ktiles // slots) if stagger else 0, "done": 0})
    total_tiles = 0
    while any(s["done"] < tiles_per_slot for s in state):
        for s in state:
            if s["done"] >= tiles_per_slot:
                continue
            if s["k"] < 0:
                s["k"] += 1
                continue
            mt, nt = s["tile"]
            k = s["k"]
            for r in range(128):
                cache.access("A", (A_BASE + ((mt * 128 + r) * K * 4 + k * 128)) // LINE)
            for r in range(128):
                cache.access("W", (W_BASE + ((nt * 128 + r) * K * 4 + k * 128)) // LINE)
            s["k"] += 1
            if s["k"] == ktiles:
                for r in range(128):
                    for c in range(4):
                        cache.access("Y", (Y_BASE + ((mt * 128 + r) * N * 4 + nt * 512 + c * 128)) // LINE)
                s["done"] += 1
                s["k"] = 0
                s["tile"] = next(seq)
                total_tiles += 1
    per_tile = {k: v * LINE / total_tiles / 1024 for k, v in cache.miss.items()}
    ideal = (128 * K * 4) / 1024
    return total_tiles, per_tile, ideal


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=6)
    a = ap.parse_args()
    print("| GEMM (N x K) | order | phases | A miss KiB/tile | W miss KiB/tile | (A or W panel = KiB) | operand miss vs no-cache |")
    print("|---|---|---|---|---|---|---|")
    for name, N, K, xn in (("fc1 2048x512", 2048, 512, 4), ("qkv 1536x512", 1536, 512, 4), ("fc2 512x2048", 512, 2048, 4),
                           ("enc fc1 3072x768", 3072, 768, 8)):
        for order, x in (("row blocks (round 1)", 1), (f"N partition xcd_n={xn}", xn)):
            for stagger in (False, True):
                n, pt, ideal = simulate(N, K, x, a.tiles, stagger)
                frac = (pt["A"] + pt["W"]) / (2 * ideal)
                print(f"| {name} | {order} | {'staggered' if stagger else 'lockstep'} | {pt['A']:.0f} | {pt['W']:.0f} | {ideal:.0f} | {frac:.2f} |")

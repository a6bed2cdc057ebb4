"""Cache lines touched per wave instruction by the global accesses of the 16x16-tile kernels (CPU, design-time audit).

A vector-memory instruction of a wave64 is processed line by line by the texture addresser / L1 (128-byte lines on gfx950): an instruction
whose 64 lanes read 1 KB from 8 lines costs half the address-path time of one that reads the same 1 KB from 16 lines. Round 3 found this
to be worth 8 of 60 us in the bf16x3 gate and 1.5-2.3 of 57 us in the fp32 gate (weights repacked in fetch order);
this script tabulates the pattern of every global access of those kernels: lines touched and the fraction of each line used.

    python tools/line_touch.py
"""
LINE = 128


def touch(addr_of_lane, width):
    lines = {}
    for lane in range(64):
        a = addr_of_lane(lane)
        for b in range(0, width, 4):
            lines.setdefault((a + b) // LINE, set()).add((a + b) % LINE)
    used = sum(len(v) * 4 for v in lines.values())
    return len(lines), used / (len(lines) * LINE)


def row(name, n, frac, per_wave, note=""):
    print(f"{name:74s} {n:3d} lines  {100 * frac:5.1f} % of each line used   x {per_wave:3d} per wave {note}")


def main():
    Kp, ldw = 256, 6 * 256                      # mel gate: 6 Winograd components x 256 channels per packed weight row (floats)
    lde, ldc = 20 * 512, 20 * 256               # conditioner slab row (20 layers x 512 packed columns), gate-output row (20 layers x 256)
    lc = lambda l: l & 15
    kg = lambda l: l >> 4
    print("--- fp32 gate (wino43_gate16.hip), one K chunk of one component = two 16-byte fetches per lane")
    n, f = touch(lambda l: ((8 * 0 + (lc(l) & 7) + 32 * (lc(l) >> 3)) * ldw + kg(l) * 8) * 4, 16)
    row("weights, packed rows [Np][6][Kp] (round-3 first form): column pc, K floats 8 kg ..", n, f, 12, "per K chunk")
    n, f = touch(lambda l: l * 16, 16)
    row("weights, fetch order [tile][wave][chunk][comp][half][lane][4] (ss_pack_gate16_weights)", n, f, 12, "per K chunk")
    n, f = touch(lambda l: ((l >> 3) * Kp + (l & 7) * 4) * 4, 16)
    row("raw rows: 8 quad rows x 128 B per instruction", n, f, 7, "per K chunk")
    n, f = touch(lambda l: ((4 * kg(l)) * lde + (8 * 0 + (lc(l) & 7) + 32 * (lc(l) >> 3))) * 4, 4)
    row("epilogue: conditioner addend, 4 bytes per lane (column pc of frame 4 kg + r)", n, f, 32, "(MT = 2)")
    n, f = touch(lambda l: ((4 * kg(l) + 2 * (lc(l) >> 3)) * ldc + (lc(l) & 7)) * 4, 4)
    row("epilogue: gate output store, 4 bytes per lane (8 channels x 4 quads x 2 frame sets)", n, f, 16, "(MT = 2)")
    print("--- residual projection (gemm16.hip)")
    n, f = touch(lambda l: (lc(l) * Kp + kg(l) * 8) * 4, 16)
    row("weights, packed rows [Np][Kp]", n, f, 16, "per launch (K = 256)")
    n, f = touch(lambda l: l * 16, 16)
    row("weights, fetch order (ss_pack_gemm16_weights)", n, f, 16, "per launch")
    n, f = touch(lambda l: ((4 * kg(l)) * 256 + lc(l)) * 4, 4)
    row("epilogue: residual-stream load / store, 4 bytes per lane (16 columns x 4 rows)", n, f, 48, "(MT = 6: 24 + 24)")
    print("--- for comparison: 32x32-tile kernels (conv_gemm, wino43_gate, wino43_conv): lane = column, 32 lanes x 4 B = one full line per row")
    n, f = touch(lambda l: ((l >> 5) * 4 * 256 + (l & 31)) * 4, 4)
    row("epilogue load / store of a 32x32 accumulator row pair", n, f, 16)
    print("--- LDS-DMA pieces (gemm16 A ring, gate256, gemm_bf16): 8 rows x 128 B per instruction")
    n, f = touch(lambda l: ((l >> 3) * 5120 + (l & 7) * 4) * 4, 16)
    row("A piece of the skip GEMM (row stride 20 KB)", n, f, 6, "per K chunk (MT = 6)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Stream-K tail of the persistent 3x3 forward kernel at single layers: microseconds per launch without scratch (-1), with the
tail forced off (1), with runs of n - 1 quads forced (PUNET_X6_SPLITK(n)) and with the library's own plan (0) - what sk_plan's
cost model in csrc/conv_x6.hip was calibrated on (profiles/r05/streamk_tail_probe.txt)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shape_sweep as SS
B = 4
for (h, w, c) in [(88, 88, 512), (64, 64, 512), (65, 88, 512), (48, 48, 512), (80, 80, 512), (176, 176, 256), (96, 96, 256), (128,128,256)]:
    fl = 2.0 * B * h * w * c * c * 9
    row = []
    for sk in (-1, 1, 2, 3, 5, 9, 12, 0):
        r = SS.layer(B, h, w, c, 4, ["fwd"], sk)
        row.append(f"s={sk}: {fl / r['fwd'] / 1e9 * 1e3:7.1f} us")
    print(f"{c}->{c}@{h}x{w}: " + "  ".join(row), flush=True)

#!/usr/bin/env python3
"""Where do the deferred-epilogue kernel's accumulators / bytes differ from the oracle?  (bring-up aid)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import abi  # noqa: E402
import orc  # noqa: E402
import synth  # noqa: E402

GEOMS = {"conv2": (300, 96, 27, 27, 256, 5, 1, 2), "conv5": (270, 384, 13, 13, 256, 3, 1, 1), "conv3": (257, 256, 13, 13, 384, 3, 1, 1)}


def report(tag, got, want):
    bad = np.argwhere(got != want)
    print(tag, "mismatches", len(bad), "of", got.size)
    if len(bad) == 0:
        return
    for ax, nm in enumerate(("image", "pixel", "feature")[: bad.shape[1]] if bad.shape[1] == 3 else ("image", "feature", "y", "x")):
        vals, cnt = np.unique(bad[:, ax], return_counts=True)
        print("  by", nm, ": distinct", len(vals), "first", vals[:24].tolist(), "counts", cnt[:24].tolist())
    print("  first", bad[:5].tolist(), "got", [int(got[tuple(b)]) for b in bad[:5]], "want", [int(want[tuple(b)]) for b in bad[:5]])


def main():
    orc.lib()
    g = abi.Ctx(0)
    for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else list(GEOMS)):
        n, c, h, w, kc, k, stride, pad = GEOMS[name]
        n = int(os.environ.get("N", n))
        cs = synth.conv_case(orc, 8800 + sum(GEOMS[name]), n, c, h, w, kc, k, stride, pad)
        for pool in (None, (3, 2)):
            names = []
            out, acc = g.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"], cs["zp_out"],
                                            stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=True, in_border=pad, out_border=0,
                                            variant=55, pool=pool, names=names)
            print("==", name, "pool", pool, names)
            report("acc", acc, cs["acc"])
            ref = orc.relu(cs["out"], cs["zp_out"])
            if pool:
                ref = orc.max_pool2d(ref, pool[0], pool[1])
            report("out", out, ref)
    g.close()


if __name__ == "__main__":
    main()

"""Shared by the example scripts: the reference notebooks' flow on the MI355X engine.

The notebooks need torchvision datasets and trained checkpoints (`alex_cifar10_224.pt`,
`conv_cifar10_32.pt`, `fc_mnist28.pt`) that the reference repository does not ship.  Each example
therefore takes `--checkpoint` / `--data` (a torch state dict; an .npz with arrays `x` [N,C,H,W]
float32 already normalised as the notebook does, and `y` [N] labels) and otherwise runs on seeded
synthetic weights and inputs of the same shapes, reporting agreement with its own FP32 run.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(network, description):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--checkpoint", help="torch state dict (.pt) with '<layer>.weight' / '<layer>.bias'")
    ap.add_argument("--data", help=".npz with x [N,C,H,W] float32 and y [N] int labels")
    ap.add_argument("--save-quantized", help="write the converted model to this .npz")
    args = ap.parse_args()

    import torch  # noqa: F401  (one HIP runtime per process: torch first)
    import int8inferenceengine_amd  # noqa: F401
    import i8ie
    from int8inferenceengine_amd import workloads as wl

    if args.checkpoint:
        sd = {k: v.numpy() for k, v in torch.load(args.checkpoint, map_location="cpu", weights_only=True).items()}
    else:
        sd = wl.synthetic_state_dict(network)
    if args.data:
        with np.load(args.data, allow_pickle=False) as f:
            xs, ys = f["x"].astype(np.float32), f["y"].astype(np.int64)
        n = min(len(xs), args.batch * args.batches)
        xs, ys = xs[:n], ys[:n]
    else:
        xs = wl.synthetic_input(network, args.batch * args.batches, seed=1234)
        ys = None

    # FP32 model (reference notebook cell "my_model = MyNet(); my_model.load(state_dict)")
    model = wl.build(network)
    model.load(sd)
    batches = [xs[i:i + args.batch] for i in range(0, len(xs), args.batch)]
    tens = [i8ie.tensor(b).prefetch() for b in batches]  # built before timing, as in the notebook

    def evaluate(tag):
        i8ie.synchronize()
        t0 = time.perf_counter()
        preds = []
        for x in tens:
            out = model(x)
            preds.append(i8ie.argmax(out, axis=1).numpy())
        dt = time.perf_counter() - t0
        preds = np.concatenate(preds).astype(np.int64)
        print("%-18s %8.1f ms  %10.0f img/s" % (tag, dt * 1e3, len(preds) / dt))
        return preds

    evaluate("FP32 (warm-up)")
    p_fp32 = evaluate("FP32")
    # calibrate + convert (notebook cell: prepare -> one FP32 batch -> convert)
    t0 = time.perf_counter()
    model.prepare()
    model(tens[0])
    model.convert()
    print("%-18s %8.1f ms" % ("prepare+convert", (time.perf_counter() - t0) * 1e3))
    evaluate("INT8 (warm-up)")
    p_int8 = evaluate("INT8")
    if ys is not None:
        print("top-1 FP32 %.4f   INT8 %.4f" % ((p_fp32 == ys).mean(), (p_int8 == ys).mean()))
    else:
        print("no labels: INT8 argmax agrees with FP32 argmax on %.2f %% of %d synthetic images"
              % (100.0 * (p_fp32 == p_int8).mean(), len(p_int8)))
    if args.save_quantized:
        model.save_quantized(args.save_quantized)
        again = wl.build(network)
        again.load_quantized_file(args.save_quantized)
        same = np.array_equal(again(tens[0]).numpy(), model(tens[0]).numpy())
        print("saved %s; reloaded model reproduces the logits: %s" % (args.save_quantized, same))

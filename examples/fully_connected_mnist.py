#!/usr/bin/env python3
"""single Linear(784,10) on MNIST (sample/notebooks/Fully_Connected_mnist.ipynb) on the MI355X engine: FP32 run, prepare/convert, INT8 run, timing and top-1."""
from _common import run

if __name__ == "__main__":
    run("mnist_fc", __doc__)

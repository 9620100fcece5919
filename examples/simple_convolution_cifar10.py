#!/usr/bin/env python3
"""3-conv net on 32x32 CIFAR-10 (sample/notebooks/Simple_Convolution_cifar10.ipynb) on the MI355X engine: FP32 run, prepare/convert, INT8 run, timing and top-1."""
from _common import run

if __name__ == "__main__":
    run("simple_conv", __doc__)

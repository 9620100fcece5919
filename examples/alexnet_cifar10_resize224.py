#!/usr/bin/env python3
"""AlexNet on 224x224 CIFAR-10 (sample/notebooks/AlexNet_cifar10_resize224.ipynb) on the MI355X engine: FP32 run, prepare/convert, INT8 run, timing and top-1."""
from _common import run

if __name__ == "__main__":
    run("alexnet", __doc__)

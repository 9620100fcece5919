"""The networks the reference benchmarks (its notebooks) expressed against the
`i8ie` surface, plus seeded synthetic weights / inputs.

The reference's trained checkpoints (alex_cifar10_224.pt, conv_cifar10_32.pt,
fc_mnist28.pt, conv28.pt) and the datasets are not in its repository, so every
measurement here uses random-init weights of the same architecture and
synthetic inputs of the same shape and value range.

A network is a list of ops ("spec") so that the same definition drives the HIP
product path (SpecNet, through i8ie) and the CPU oracle in tests/bench.
"""
import numpy as np

# name -> (layers, spec, input_shape CHW)
#   layers: {attr: ("conv", in_c, out_c, k, stride, pad) | ("fc", in_f, out_f)}
#   spec  : [("layer", attr) | ("relu",) | ("pool", k, s) | ("flatten", features)]
NETWORKS = {
    # sample/notebooks/AlexNet_cifar10_resize224.ipynb:47-71
    "alexnet": (
        {
            "conv1": ("conv", 3, 96, 11, 4, 2), "conv2": ("conv", 96, 256, 5, 1, 2),
            "conv3": ("conv", 256, 384, 3, 1, 1), "conv4": ("conv", 384, 384, 3, 1, 1),
            "conv5": ("conv", 384, 256, 3, 1, 1), "fc1": ("fc", 256 * 6 * 6, 4096),
            "fc2": ("fc", 4096, 4096), "fc3": ("fc", 4096, 10),
        },
        [("layer", "conv1"), ("relu",), ("pool", 3, 2), ("layer", "conv2"), ("relu",), ("pool", 3, 2),
         ("layer", "conv3"), ("relu",), ("layer", "conv4"), ("relu",), ("layer", "conv5"), ("relu",),
         ("pool", 3, 2), ("flatten", 9216), ("layer", "fc1"), ("relu",), ("layer", "fc2"), ("relu",),
         ("layer", "fc3")],
        (3, 224, 224),
    ),
    # sample/notebooks/Simple_Convolution_cifar10.ipynb:40-57 (3 convs + fc)
    "simple_conv": (
        {"conv1": ("conv", 3, 20, 5, 1, 0), "conv2": ("conv", 20, 50, 5, 1, 0),
         "conv3": ("conv", 50, 120, 5, 1, 0), "fc": ("fc", 960 * 8, 10)},
        [("layer", "conv1"), ("relu",), ("layer", "conv2"), ("relu",), ("pool", 2, 2), ("layer", "conv3"),
         ("relu",), ("flatten", 7680), ("layer", "fc")],
        (3, 32, 32),
    ),
    # unittest/test_quantized_layer.py:26-42 (2 convs, 1x28x28)
    "two_conv": (
        {"conv1": ("conv", 1, 20, 5, 1, 0), "conv2": ("conv", 20, 50, 5, 1, 0),
         "fc1": ("fc", 800, 500), "fc2": ("fc", 500, 10)},
        [("layer", "conv1"), ("pool", 2, 2), ("layer", "conv2"), ("pool", 2, 2), ("flatten", 800),
         ("layer", "fc1"), ("relu",), ("layer", "fc2")],
        (1, 28, 28),
    ),
    # sample/notebooks/Fully_Connected_mnist.ipynb:28-35
    "mnist_fc": ({"fc": ("fc", 784, 10)}, [("flatten", 784), ("layer", "fc")], (1, 28, 28)),
}

# MACs per image: the algorithmic work of the INT8 contractions
ALEXNET_MACS_PER_IMAGE = 1131201056


def macs_per_image(name):
    layers, spec, (c, h, w) = NETWORKS[name]
    total = 0
    for op in spec:
        if op[0] == "layer":
            L = layers[op[1]]
            if L[0] == "conv":
                _, ic, oc, k, s, p = L
                h, w = (h - k + 2 * p) // s + 1, (w - k + 2 * p) // s + 1
                total += h * w * oc * ic * k * k
                c = oc
            else:
                total += L[1] * L[2]
        elif op[0] == "pool":
            h, w = (h - op[1]) // op[2] + 1, (w - op[1]) // op[2] + 1
    return total


def synthetic_state_dict(name, seed=42):
    """He-uniform weights, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) biases; keys '<attr>.weight'/'<attr>.bias'."""
    rng = np.random.default_rng(seed)
    layers = NETWORKS[name][0]
    sd = {}
    for attr, L in layers.items():
        shape = (L[2], L[1], L[3], L[3]) if L[0] == "conv" else (L[2], L[1])
        fan_in = int(np.prod(shape[1:]))
        sd[attr + ".weight"] = (rng.uniform(-1, 1, shape) * np.sqrt(6.0 / fan_in)).astype(np.float32)
        sd[attr + ".bias"] = (rng.uniform(-1, 1, shape[0]) / np.sqrt(fan_in)).astype(np.float32)
    return sd


def synthetic_input(name, batch, seed=1234):
    """Normalised-image-like input, inside quantize()'s no-wrap window for scale 0.025 / zp 127."""
    c, h, w = NETWORKS[name][2]
    rng = np.random.default_rng(seed)
    if name in ("mnist_fc", "two_conv"):
        return rng.uniform(0, 1, (batch, c, h, w)).astype(np.float32)  # ToTensor() only
    u = rng.uniform(0, 1, (batch, c, h, w)).astype(np.float32)
    a = rng.uniform(0.2, 1.0, (batch, c, 1, 1)).astype(np.float32)
    return ((u * a - np.float32(0.45)) / np.float32(0.226)).astype(np.float32)


def build(name):
    """An i8ie.Module running NETWORKS[name] on the GPU."""
    import int8inferenceengine_amd  # noqa: F401  (puts i8ie / _CXX_i8ie on sys.path)
    import i8ie

    layers, spec, _ = NETWORKS[name]

    class SpecNet(i8ie.Module):
        def __init__(self):
            super().__init__()
            for attr, L in layers.items():
                if L[0] == "conv":
                    setattr(self, attr, i8ie.Conv2d(L[1], L[2], kernel_size=L[3], stride=L[4], padding=L[5]))
                else:
                    setattr(self, attr, i8ie.Linear(L[1], L[2]))

        def forward(self, x):
            for op in spec:
                if op[0] == "layer":
                    x = getattr(self, op[1])(x)
                elif op[0] == "relu":
                    x = i8ie.relu(x)
                elif op[0] == "pool":
                    x = i8ie.max_pool2d(x, op[1], op[2])
                else:
                    x = x.reshape(-1, op[1])
            return x

    SpecNet.__name__ = "SpecNet_" + name
    return SpecNet()


def calibrated(name, state_dict=None, calib_batch=None, seed=42, calib_seed=7):
    """The reference workflow (notebook cells 'prepare -> one FP32 batch -> convert')."""
    import _CXX_i8ie as cx
    import i8ie

    net = build(name)
    net.load(state_dict if state_dict is not None else synthetic_state_dict(name, seed))
    cx.set_calibration_seed(calib_seed)  # the reference's calibrator is unseeded (std::random_device)
    net.prepare()
    x = calib_batch if calib_batch is not None else synthetic_input(name, 100 if name != "alexnet" else 32, seed=99)
    net(i8ie.tensor(x))
    net.convert()
    return net


def layer_names(name):
    return [op[1] for op in NETWORKS[name][1] if op[0] == "layer"]

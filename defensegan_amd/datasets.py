"""Input side of the projection path: the reference's on-disk image format for MNIST / F-MNIST.

Counterpart of ``Mnist.load`` (/root/reference/datasets/mnist.py:46-103; F-MNIST uses the same container,
datasets/fmnist.py): idx-ubyte files, 16-byte image header / 8-byte label header (mnist.py:65-79), the
50 000 / 10 000 / 10 000 train / val / test split (mnist.py:83-91), images as float [n,28,28,1] in [0,255].
``to_generator_range`` applies the GAN classes' input transform (gan.py:684-685, 697-698, 764-765).
No dataset ships with this repository; the functions only parse files the user already has.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np

IDX_FILES = {
    "train_images": "train-images-idx3-ubyte", "train_labels": "train-labels-idx1-ubyte",
    "test_images": "t10k-images-idx3-ubyte", "test_labels": "t10k-labels-idx1-ubyte",
}


def read_idx_images(path: str) -> np.ndarray:
    raw = np.fromfile(path, dtype=np.uint8)
    n = (raw.size - 16) // 784
    return raw[16:16 + n * 784].reshape(n, 28, 28, 1).astype(np.float32)


def read_idx_labels(path: str) -> np.ndarray:
    raw = np.fromfile(path, dtype=np.uint8)
    return raw[8:].astype(np.int64)


def load_mnist_split(data_dir: str, split: str = "test", randomize: bool = False,
                     seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """split in {'train', 'val' (the reference's dev split), 'test'}; images float32 [n,28,28,1] in [0,255]."""
    if split not in ("train", "val", "dev", "test"):
        raise ValueError("split must be one of train|val|test")
    if split == "test":
        images = read_idx_images(os.path.join(data_dir, IDX_FILES["test_images"]))
        labels = read_idx_labels(os.path.join(data_dir, IDX_FILES["test_labels"]))
    else:
        images = read_idx_images(os.path.join(data_dir, IDX_FILES["train_images"]))
        labels = read_idx_labels(os.path.join(data_dir, IDX_FILES["train_labels"]))
        cut = min(50000, len(images) * 5 // 6)
        images, labels = (images[:cut], labels[:cut]) if split == "train" else (images[cut:], labels[cut:])
    if randomize:                       # the reference shuffles images and labels with the same RNG state (mnist.py:93-97)
        perm = np.random.RandomState(seed).permutation(len(images))
        images, labels = images[perm], labels[perm]
    return images, labels


def to_generator_range(images: np.ndarray, dataset_name: str) -> np.ndarray:
    """[0,255] -> [0,1] (mnist, f-mnist) or [-1,1] (celeba)."""
    x = np.asarray(images, np.float32) / np.float32(255.0)
    if dataset_name.lower() == "celeba":
        x = np.float32(2.0) * (x - np.float32(0.5))
    return x

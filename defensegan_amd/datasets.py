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


# ---------------------------------------------------------------------------------------------------- CelebA (64 x 64 x 3)
# Counterpart of ``CelebA.load`` + ``LazyDataset`` (/root/reference/datasets/celeba.py:47-140, datasets/dataset.py:66-182,
# 216-262): files ``{i:06d}.jpg`` (1-based), train / val / test = images 1-162770 / 162771-182637 / 182638-202599, each
# image read on access, centre-cropped to 108 x 108 (top-left = round((h - 108) / 2), round((w - 108) / 2)) and resized to
# 64 x 64 by ``scipy.misc.imresize`` -- which is PIL: the float crop is first byte-scaled over its OWN min / max
# (scipy.misc.bytescale: (x - min) * 255 / (max - min) + 0.5 -> uint8), then ``Image.resize((64, 64), BILINEAR)``.  scipy.misc
# no longer exists; its published semantics are restated here on top of Pillow (needed only for this loader).
CELEBA_SPLITS = {"train": (1, 162770), "val": (162771, 182637), "dev": (162771, 182637), "test": (182638, 202599)}


def _bytescale(data: np.ndarray) -> np.ndarray:
    """scipy.misc.bytescale with its defaults (cmin / cmax = data min / max, low 0, high 255)."""
    if data.dtype == np.uint8:
        return data
    cmin, cmax = float(data.min()), float(data.max())
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1.0
    scaled = (data.astype(np.float64) - cmin) * (255.0 / cscale)
    return (scaled.clip(0, 255) + 0.5).astype(np.uint8)


def prepare_celeba_image(image: np.ndarray, crop: int = 108, size: int = 64) -> np.ndarray:
    """``_prepare_image(image, crop, crop, size, size, is_crop=True)`` (dataset.py:216-262): float [h,w,3] in [0,255] ->
    uint8-valued float32 [size,size,3] in [0,255]."""
    from PIL import Image           # only this loader needs Pillow
    h, w = image.shape[:2]
    j = int(np.floor((h - crop) / 2.0 + 0.5))     # Python-2 round(): half away from zero (non-negative here)
    i = int(np.floor((w - crop) / 2.0 + 0.5))
    patch = _bytescale(np.asarray(image[j:j + crop, i:i + crop], dtype=np.float64))
    out = Image.fromarray(patch).resize((size, size), resample=Image.BILINEAR)
    return np.asarray(out, dtype=np.float32)


class LazyCelebA(object):
    """``LazyDataset`` (dataset.py:66-182): indexable by int / slice / index array, loads on access."""

    def __init__(self, filepaths, center_crop_dim: int = 108, resize_size: int = 64):
        self.filepaths = list(filepaths)
        self.center_crop_dim = center_crop_dim
        self.resize_size = resize_size

    def _get_image(self, path: str) -> np.ndarray:
        from PIL import Image
        with Image.open(path) as im:
            arr = np.asarray(im.convert("RGB"), dtype=np.float64)
        return prepare_celeba_image(arr, self.center_crop_dim, self.resize_size)

    def __len__(self):
        return len(self.filepaths)

    def __getitem__(self, index):
        if isinstance(index, (int, np.integer)):
            return self._get_image(self.filepaths[int(index)])
        if isinstance(index, slice):
            index = range(*index.indices(len(self.filepaths)))
        try:
            inds = [int(i) for i in index]
        except TypeError:
            raise TypeError("Index must be an integer, a slice, a container or an integer generator.")
        return np.array([self._get_image(self.filepaths[i]) for i in inds])

    def get_subset(self, indices):
        if isinstance(indices, slice):
            indices = range(*indices.indices(len(self.filepaths)))
        self.filepaths = [self.filepaths[int(i)] for i in indices]

    @property
    def shape(self):
        return (None, self.resize_size, self.resize_size, 3)


def load_celeba_split(data_dir: str, split: str = "test", attribute=None, randomize: bool = False, seed: int = 0):
    """(lazy images, labels or None).  ``attribute='gender'`` reads the ``male`` column of ``list_attr_celeba.txt`` as 0 / 1
    (celeba.py:113-131); file existence is not checked here (the reference does not either)."""
    if split not in CELEBA_SPLITS:
        raise ValueError("[!] Invalid split {}.".format(split))
    start, end = CELEBA_SPLITS[split]
    fps = [os.path.join(data_dir, "{:06d}.jpg".format(i)) for i in range(start, end + 1)]
    labels = None
    if attribute is not None:
        if attribute != "gender":
            raise ValueError("[!] Invalid attribute {} for CelebA dataset.".format(attribute))
        with open(os.path.join(data_dir, "list_attr_celeba.txt")) as f:
            lines = f.readlines()
        names = [s.lower().replace(" ", "_") for s in lines[1].strip().split()]
        col = names.index("male")
        attrs = np.asarray([[int(v) for v in ln.split()[1:]] for ln in (l.strip() for l in lines[2:]) if ln], dtype=np.int64)
        labels = ((attrs + 1) // 2)[start - 1:end, col].reshape(-1)
    if randomize:
        perm = np.random.RandomState(seed).permutation(len(fps))
        fps = [fps[k] for k in perm]
        labels = labels[perm] if labels is not None else None
    return LazyCelebA(fps), labels

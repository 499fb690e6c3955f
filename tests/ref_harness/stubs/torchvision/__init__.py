"""TEST SCAFFOLDING: stand-in for the parts of torchvision the reference's examples-pnode/train-Cifar10.py uses, for an
image without torchvision and without network access.  `datasets.CIFAR10` is a small SYNTHETIC dataset of the CIFAR-10
shape (3 x 32 x 32 float tensors, labels 0..9); the transforms are shape-preserving no-ops on tensors."""
from . import datasets, transforms  # noqa: F401

import os

import torch
from torch.utils.data import Dataset


class CIFAR10(Dataset):
    """Synthetic CIFAR-10-shaped data: class k = a fixed random pattern + noise (learnable, deterministic)."""

    def __init__(self, root=None, train=True, download=False, transform=None):
        n = int(os.environ.get("PN_FAKE_CIFAR_N", "64" if train else "32"))
        g = torch.Generator().manual_seed(0 if train else 1)
        protos = torch.rand(10, 3, 32, 32, generator=torch.Generator().manual_seed(7))
        self.labels = torch.randint(0, 10, (n,), generator=g)
        self.images = (0.6 * protos[self.labels] + 0.4 * torch.rand(n, 3, 32, 32, generator=g)).clamp(0, 1)
        self.transform = transform

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        x = self.images[i]
        if self.transform is not None:
            x = self.transform(x)
        return x, int(self.labels[i])

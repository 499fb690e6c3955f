class Compose(object):
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _Identity(object):
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


RandomCrop = RandomHorizontalFlip = ToTensor = _Identity


class Normalize(object):
    def __init__(self, mean, std):
        import torch
        self.mean = torch.tensor(mean).view(-1, 1, 1)
        self.std = torch.tensor(std).view(-1, 1, 1)

    def __call__(self, x):
        return (x - self.mean) / self.std

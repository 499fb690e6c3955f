"""TEST SCAFFOLDING: torchsummary.summary is imported by the reference's train-Cifar10.py; a no-op here."""


def summary(*a, **k):
    return None

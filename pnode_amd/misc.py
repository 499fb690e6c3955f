"""Parameter flattening helpers (reference: ``pnode/misc.py``).

The order of the flat parameter vector -- and therefore of the flat parameter gradient the
adjoint returns -- is ``func.parameters()`` order restricted to ``requires_grad`` tensors
(``pnode/petsc_adjoint.py:618-620``).
"""
import torch


def _flatten(sequence):
    """Concatenate the tensors of `sequence` as one 1-D tensor (views, so autograd edges to
    the original parameters are kept); an empty sequence gives an empty tensor."""
    pieces = [p.contiguous().view(-1) for p in sequence]
    if not pieces:
        return torch.tensor([])
    return torch.cat(pieces)


def _flatten_convert_none_to_zeros(sequence, like_sequence):
    """Same, with ``None`` entries (unused parameters in a VJP) replaced by zeros shaped like
    the corresponding entry of `like_sequence`."""
    pieces = []
    for p, q in zip(sequence, like_sequence):
        pieces.append(torch.zeros_like(q).view(-1) if p is None else p.contiguous().view(-1))
    if not pieces:
        return torch.tensor([])
    return torch.cat(pieces)

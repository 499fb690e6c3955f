"""Flat view of a module's trainable parameters.

The order of the flat parameter vector -- and therefore of the flat parameter gradient the adjoint
returns -- is ``func.parameters()`` order restricted to ``requires_grad`` tensors (reference
``pnode/petsc_adjoint.py:618-620``; its helper lives in ``pnode/misc.py``).  The gradient side needs no
helper here: ``pn_param_accum`` adds each parameter's cotangent into its slice of the flat buffer on
the device and skips missing (``None``) ones.
"""
import torch


def flat_parameters(params):
    """One 1-D tensor holding `params` back to back, differentiable with respect to each of them
    (the autograd Function receives it so that the graph keeps an edge to every parameter)."""
    params = list(params)
    if not params:
        return torch.empty(0)
    return torch.cat([p.reshape(-1) for p in params])


_flatten = flat_parameters      # the reference's name

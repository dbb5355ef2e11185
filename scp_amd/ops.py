"""Dense-layer helpers used by the context models (fp32).

Plain library GEMMs (rocBLAS / hipBLASLt through torch.nn.functional.linear) carry the Linear layers;
LayerNorm / GELU / LeakyReLU epilogues are device ops on the same stream.  Everything here requires
device tensors - there is no CPU path in the product.
"""
import torch
import torch.nn.functional as F


def linear(x, w, b=None):
    return F.linear(x, w, b)


def layer_norm(x, ln):
    return F.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)


def gelu_linear(x, w, b):
    return F.gelu(F.linear(x, w, b))


def leaky_mlp3(seq, x, slope=0.01):
    """nn.Sequential(Linear, LeakyReLU, Linear, LeakyReLU, Linear) as stored under keys .0 / .2 / .4"""
    x = F.leaky_relu(F.linear(x, seq[0].weight, seq[0].bias), slope)
    x = F.leaky_relu(F.linear(x, seq[2].weight, seq[2].bias), slope)
    return F.linear(x, seq[4].weight, seq[4].bias)

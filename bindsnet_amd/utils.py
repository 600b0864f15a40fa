"""Weight / assignment reshaping helpers: API mirror of bindsnet/utils.py (host-side plumbing for the plotting code of
the examples; nothing here is on the per-timestep path)."""
import math
from typing import Tuple, Union

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.modules.utils import _pair


def im2col_indices(x: Tensor, kernel_height: int, kernel_width: int, padding: Tuple[int, int] = (0, 0),
                   stride: Tuple[int, int] = (1, 1)) -> Tensor:
    """utils.py:11-29 (= F.unfold)."""
    return F.unfold(x, (kernel_height, kernel_width), padding=padding, stride=stride)


def col2im_indices(cols: Tensor, x_shape: Tuple[int, int, int, int], kernel_height: int, kernel_width: int,
                   padding: Tuple[int, int] = (0, 0), stride: Tuple[int, int] = (1, 1)) -> Tensor:
    """utils.py:32-54 (= F.fold)."""
    return F.fold(cols, x_shape, (kernel_height, kernel_width), padding=padding, stride=stride)


def get_square_weights(weights: Tensor, n_sqrt: int, side: Union[int, Tuple[int, int]]) -> Tensor:
    """[n_inputs, n_filters] -> one image tiling the first n_sqrt**2 filters row by row (utils.py:57-85)."""
    sh, sw = (side, side) if isinstance(side, int) else side
    n = min(weights.size(1), n_sqrt * n_sqrt)
    grid = torch.zeros(n_sqrt * n_sqrt, sh, sw)
    grid[:n] = weights[:, :n].t().reshape(n, sh, sw).to(grid.device)
    return grid.view(n_sqrt, n_sqrt, sh, sw).permute(0, 2, 1, 3).reshape(n_sqrt * sh, n_sqrt * sw)


def get_square_assignments(assignments: Tensor, n_sqrt: int) -> Tensor:
    """Label vector -> n_sqrt x n_sqrt grid, -1 where there is no neuron (utils.py:88-109)."""
    grid = -torch.ones(n_sqrt * n_sqrt)
    n = min(assignments.size(0), n_sqrt * n_sqrt)
    grid[:n] = assignments[:n].to(grid.dtype).to(grid.device)
    return grid.view(n_sqrt, n_sqrt)


def reshape_conv2d_weights(weights: torch.Tensor) -> torch.Tensor:
    """[out, in, kh, kw] -> one image (utils.py:183-216): the output filters tile a ceil(sqrt(out))-wide grid, and that
    whole grid is repeated on a ceil(sqrt(in))-wide grid of input channels (channel k*s2+l at block row k, block col l)."""
    n_out, n_in, kh, kw = weights.shape
    s1, s2 = int(math.ceil(math.sqrt(n_out))), int(math.ceil(math.sqrt(n_in)))
    img = torch.zeros(s1 * s2 * kh, s1 * s2 * kw)
    for f in range(n_out):
        i, j = divmod(f, s1)
        for ci in range(n_in):
            k, l = divmod(ci, s2)
            r0, c0 = (i + k * s1) * kh, (j + l * s1) * kw
            img[r0:r0 + kh, c0:c0 + kw] = weights[f, ci]
    return img


def reshape_locally_connected_weights(w: Tensor, n_filters: int, kernel_size: Union[int, Tuple[int, int]],
                                      conv_size: Union[int, Tuple[int, int]], locations: Tensor,
                                      input_sqrt: Union[int, Tuple[int, int]]) -> Tensor:
    """Weights [n_input, n_filters * c1 * c2] of a LocalConnection -> one image (utils.py:112-180): the k1 x k2 receptive
    field of filter f at convolution position (n1, n2) -- rows `locations[:, n]`, column f * c1 * c2 + n of `w` -- lands in
    block (n1 * fs + f // fs, n2 * fs + f % fs) of a grid with fs = ceil(sqrt(n_filters)).  With a single position
    (c1 = c2 = 1, the kernel covers the whole input) the filters simply tile an fs x fs grid of input-sized images."""
    (k1, k2), (c1, c2), (i1, i2) = _pair(kernel_size), _pair(conv_size), _pair(input_sqrt)
    fs, C = int(math.ceil(math.sqrt(n_filters))), c1 * c2
    cols = torch.arange(n_filters).view(-1, 1, 1) * C + torch.arange(C).view(1, -1, 1)          # [F, C, 1]
    rows = locations.t().unsqueeze(0)                                                           # [1, C, k1*k2]
    fields = w.detach().cpu()[rows, cols].view(n_filters, c1, c2, k1, k2).float()               # [F, n1, n2, k1, k2]
    if c1 == 1 and c2 == 1:
        grid = torch.zeros(fs * fs, i1, i2)
        grid[:n_filters] = fields.view(n_filters, k1, k2)[:, :i1, :i2]
        return grid.view(fs, fs, i1, i2).permute(0, 2, 1, 3).reshape(fs * i1, fs * i2)
    grid = torch.zeros(c1, fs * fs, k1, c2, k2)
    grid[:, :n_filters] = fields.permute(1, 0, 3, 2, 4)
    return grid.view(c1, fs, fs, k1, c2, k2).permute(0, 1, 3, 4, 2, 5).reshape(c1 * fs * k1, c2 * fs * k2)

"""Live matplotlib views used by the examples: API mirror of bindsnet/analysis/plotting.py for the functions
examples/mnist/eth_mnist.py calls (`plot_input`, `plot_spikes`, `plot_weights`, `plot_assignments`,
`plot_performance`, `plot_voltages`) plus `plot_conv2d_weights` and `plot_locally_connected_weights`.  Same signatures and return values (the handles are
passed back in to redraw instead of recreating figures).  Host-side only: tensors are copied to the CPU for drawing."""
from typing import Dict, List, Optional, Sized, Tuple, Union

import matplotlib.pyplot as plt
import numpy as np
import torch
from mpl_toolkits.axes_grid1 import make_axes_locatable

from ..utils import reshape_conv2d_weights, reshape_locally_connected_weights

plt.ion()


def _np(t) -> np.ndarray:
    return t.detach().clone().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def plot_input(image, inpt, label: Optional[int] = None, axes=None, ims=None, figsize: Tuple[int, int] = (8, 4)):
    """Current image next to its time-summed spike encoding (plotting.py:21-70).  Returns (axes, ims)."""
    local_image, local_inpt = _np(image), _np(inpt)
    if axes is None:
        fig, axes = plt.subplots(1, 2, figsize=figsize)
        ims = axes[0].imshow(local_image, cmap="binary"), axes[1].imshow(local_inpt, cmap="binary")
        axes[0].set_title("Current image" if label is None else "Current image (label = %d)" % label)
        axes[1].set_title("Reconstruction")
        for ax in axes:
            ax.set_xticks(())
            ax.set_yticks(())
        fig.tight_layout()
    else:
        if label is not None:
            axes[0].set_title("Current image (label = %d)" % label)
        ims[0].set_data(local_image)
        ims[1].set_data(local_inpt)
    return axes, ims


def _windows(data: Dict[str, torch.Tensor], time, n_neurons):
    flat = {k: v.view(v.size(0), -1) for k, v in data.items()}
    if time is None:
        time = (0, next(iter(flat.values())).shape[0])
    n_neurons = dict(n_neurons or {})
    for k, v in flat.items():
        n_neurons.setdefault(k, (0, v.shape[1]))
    return flat, time, n_neurons


def plot_spikes(spikes: Dict[str, torch.Tensor], time: Optional[Tuple[int, int]] = None,
                n_neurons: Optional[Dict[str, Tuple[int, int]]] = None, ims=None, axes=None,
                figsize: Tuple[float, float] = (8.0, 4.5)):
    """Raster plot per layer; spikes[layer] has shape [time, *layer shape] (plotting.py:73-178).  Returns (ims, axes)."""
    flat, time, n_neurons = _windows(spikes, time, n_neurons)
    first = ims is None
    if first:
        fig, axes = plt.subplots(len(flat), 1, figsize=figsize)
        axes = [axes] if len(flat) == 1 else list(axes)
        ims = []
    for i, (name, rec) in enumerate(flat.items()):
        lo, hi = n_neurons[name]
        pts = np.array(_np(rec[time[0]:time[1], lo:hi]).nonzero()).T
        if first:
            ims.append(axes[i].scatter(x=pts[:, 0], y=pts[:, 1], s=1))
            axes[i].set_aspect("auto")
        else:
            ims[i].set_offsets(pts)
        axes[i].set_title("%s spikes for neurons (%d - %d) from t = %d to %d " % (name, lo, hi, time[0], time[1]))
        axes[i].set_yticks([lo, hi])
    if first:
        plt.setp(axes, xticks=[], xlabel="Simulation time", ylabel="Neuron index")
        plt.tight_layout()
    plt.draw()
    return ims, axes


def _image_with_colorbar(data, figsize, cmap, vmin, vmax, title=None, ticks=None, ticklabels=None):
    fig, ax = plt.subplots(figsize=figsize)
    im = ax.imshow(data, cmap=cmap, vmin=vmin, vmax=vmax)
    cax = make_axes_locatable(ax).append_axes("right", size="5%", pad=0.05)
    if title:
        ax.set_title(title)
    ax.set_xticks(())
    ax.set_yticks(())
    ax.set_aspect("auto")
    cb = plt.colorbar(im, cax=cax, ticks=ticks)
    if ticklabels is not None:
        cb.ax.set_yticklabels(ticklabels)
    fig.tight_layout()
    return im


def plot_weights(weights: torch.Tensor, wmin: Optional[float] = 0, wmax: Optional[float] = 1, im=None,
                 figsize: Tuple[int, int] = (5, 5), cmap: str = "hot_r", save: Optional[str] = None,
                 title: Optional[str] = None):
    """Weight matrix as an image (plotting.py:181-261).  Returns the AxesImage."""
    local = _np(weights)
    if im is None:
        im = _image_with_colorbar(local, figsize, cmap, wmin, wmax, title)
    else:
        im.set_data(local)
    if save is not None:
        plt.savefig(save, bbox_inches="tight")
    return im


def plot_conv2d_weights(weights: torch.Tensor, wmin: float = 0.0, wmax: float = 1.0, im=None,
                        figsize: Tuple[int, int] = (5, 5), cmap: str = "hot_r"):
    """Conv2dConnection kernels tiled into one image (plotting.py:264-319)."""
    return plot_weights(reshape_conv2d_weights(weights), wmin, wmax, im, figsize, cmap)


def plot_locally_connected_weights(weights: torch.Tensor, n_filters: int, kernel_size, conv_size, locations: torch.Tensor, input_sqrt,
                                   wmin: float = 0.0, wmax: float = 1.0, im=None, lines: bool = True,
                                   figsize: Tuple[int, int] = (5, 5), cmap: str = "hot_r", title: Optional[str] = None):
    """Receptive fields of a LocalConnection as one image, the blocks of neighbouring input regions separated by dashed
    lines (plotting.py:322-401).  Returns the AxesImage."""
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    kernel_size, conv_size, input_sqrt = pair(kernel_size), pair(conv_size), pair(input_sqrt)
    image = _np(reshape_locally_connected_weights(weights, n_filters, kernel_size, conv_size, locations, input_sqrt))
    if im is not None:
        im.set_data(image)
        return im
    im = _image_with_colorbar(image, figsize, cmap, wmin, wmax, None if title is None else title + " Weights")
    if lines:
        fs = int(np.ceil(np.sqrt(n_filters)))
        for axis_line, k, c in ((im.axes.axhline, kernel_size[0], conv_size[0]), (im.axes.axvline, kernel_size[1], conv_size[1])):
            for edge in range(fs * k, fs * c * k, fs * k):
                axis_line(edge - 0.5, color="g", linestyle="--")
    return im


def plot_assignments(assignments: torch.Tensor, im=None, figsize: Tuple[int, int] = (5, 5),
                     classes: Optional[Sized] = None, save: Optional[str] = None):
    """Grid of per-neuron class labels (plotting.py:487-578).  Returns the AxesImage."""
    local = _np(assignments)
    if im is None:
        n = 11 if classes is None else len(classes) + 1
        labels = (["none"] + [str(k) for k in range(n - 1)]) if classes is None else (["none"] + [str(c) for c in classes])
        im = _image_with_colorbar(local, figsize, plt.get_cmap("RdBu", n), -1.5, n - 1.5, "Categorical assignments",
                                  ticks=list(range(-1, n - 1)), ticklabels=labels)
    else:
        im.set_data(local)
    if save is not None:
        plt.savefig(save, bbox_inches="tight")
    return im


def plot_performance(performances: Dict[str, List[float]], ax=None, figsize: Tuple[int, int] = (7, 4),
                     x_scale: int = 1, save: Optional[str] = None):
    """Accuracy curves, one per classification scheme (plotting.py:581-641).  Returns the Axes."""
    if ax is None:
        _, ax = plt.subplots(figsize=figsize)
    else:
        ax.clear()
    for scheme, vals in performances.items():
        ax.plot([n * x_scale for n in range(len(vals))], list(vals), label=scheme)
    ax.set_ylim([0, 100])
    ax.set_title("Estimated classification accuracy")
    ax.set_xlabel("No. of examples")
    ax.set_ylabel("Accuracy")
    ax.set_xticks(())
    ax.set_yticks(range(0, 110, 10))
    if performances:
        ax.legend()
    if save is not None:
        plt.savefig(save, bbox_inches="tight")
    return ax


def plot_voltages(voltages: Dict[str, torch.Tensor], ims=None, axes=None, time: Tuple[int, int] = None,
                  n_neurons: Optional[Dict[str, Tuple[int, int]]] = None, cmap: Optional[str] = "jet",
                  plot_type: str = "color", thresholds: Dict[str, torch.Tensor] = None,
                  figsize: Tuple[float, float] = (8.0, 4.5)):
    """Membrane potentials per layer as a heat map ("color") or as traces ("line"), optional threshold lines
    (plotting.py:644-815).  Returns (ims, axes)."""
    flat, time, n_neurons = _windows(voltages, time, n_neurons)
    first = ims is None
    if first:
        fig, axes = plt.subplots(len(flat), 1, figsize=figsize)
        axes = [axes] if len(flat) == 1 else list(axes)
        ims = []
    for i, (name, rec) in enumerate(flat.items()):
        lo, hi = n_neurons[name]
        window = _np(rec[time[0]:time[1], lo:hi])
        ax = axes[i]
        if not first:
            ax.clear()
        if plot_type == "line":
            drawn = ax.plot(window)
            thr = None if thresholds is None else thresholds.get(name)
            if thr is not None and thr.numel() == 1:
                ax.axhline(y=float(thr.item()), c="r", linestyle="--")
        else:
            drawn = ax.pcolormesh(window.T, cmap=cmap)
        if first:
            ims.append(drawn)
        else:
            ims[i] = drawn
        ax.set_title("%s voltages for neurons (%d - %d) from t = %d to %d " % (name, lo, hi, time[0], time[1]))
        ax.set_aspect("auto")
    if first:
        plt.setp(axes, xlabel="Simulation time", ylabel="Voltage" if plot_type == "line" else "Neuron index")
        plt.tight_layout()
    return ims, axes

"""Classification read-outs over recorded spikes: API mirror of bindsnet/evaluation/evaluation.py.
Plain torch on whatever device the spike records live on (eth_mnist.py keeps them on the GPU); none of this is
on the per-timestep path."""
from itertools import product
from typing import Dict, Optional, Tuple

import torch


def assign_labels(spikes: torch.Tensor, labels: torch.Tensor, n_labels: int, rates: Optional[torch.Tensor] = None,
                  alpha: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Each neuron gets the label it fires most for (evaluation.py:8-61).
    spikes [n_samples, time, n_neurons], labels [n_samples] -> (assignments [n], proportions [n, L], rates [n, L])."""
    n_neurons = spikes.size(2)
    if rates is None:
        rates = torch.zeros((n_neurons, n_labels), device=spikes.device)
    counts = spikes.sum(1)                                       # spike order within a sample does not matter
    for i in range(n_labels):
        sel = labels == i
        n_labeled = torch.sum(sel).float()
        if n_labeled > 0:
            idx = torch.nonzero(sel).view(-1)
            rates[:, i] = alpha * rates[:, i] + torch.sum(torch.index_select(counts, 0, idx), 0) / n_labeled
    proportions = rates / rates.sum(1, keepdim=True)
    proportions[proportions != proportions] = 0                  # 0 / 0 -> 0
    return torch.max(proportions, 1)[1], proportions, rates


def logreg_fit(spikes: torch.Tensor, labels: torch.Tensor, logreg):
    """(Re)fit a scikit-learn LogisticRegression on time-summed spikes (evaluation.py:64-79)."""
    logreg.fit(spikes, labels)
    return logreg


def logreg_predict(spikes: torch.Tensor, logreg) -> torch.Tensor:
    """evaluation.py:82-96: -1 for every example until the model has been fitted."""
    if not hasattr(logreg, "coef_") or logreg.coef_ is None:
        return -1 * torch.ones(spikes.size(0)).long()
    return torch.Tensor(logreg.predict(spikes)).long()


def _label_rates(spikes, assignments, n_labels, weights=None):
    counts = spikes.sum(1)
    if counts.is_sparse:
        counts = counts.to_dense()
    rates = torch.zeros((spikes.size(0), n_labels), device=counts.device)
    for i in range(n_labels):
        sel = assignments == i
        n_assigns = torch.sum(sel).float()
        if n_assigns > 0:
            idx = torch.nonzero(sel).view(-1)
            if weights is None:
                rates[:, i] = torch.sum(counts[:, idx], 1) / n_assigns
            else:
                rates[:, i] += torch.sum((weights[:, i] * counts)[:, idx], 1) / n_assigns
    return rates


def all_activity(spikes: torch.Tensor, assignments: torch.Tensor, n_labels: int) -> torch.Tensor:
    """Label with the highest mean activity of the neurons assigned to it (evaluation.py:99-133)."""
    return torch.sort(_label_rates(spikes, assignments, n_labels), dim=1, descending=True)[1][:, 0]


def proportion_weighting(spikes: torch.Tensor, assignments: torch.Tensor, proportions: torch.Tensor,
                         n_labels: int) -> torch.Tensor:
    """As all_activity, each neuron weighted by its class proportion (evaluation.py:136-180)."""
    return torch.sort(_label_rates(spikes, assignments, n_labels, proportions), dim=1, descending=True)[1][:, 0]


def _firing_sequence(activity: torch.Tensor, grouped: bool):
    seq = []
    for t in range(activity.size(0)):
        idx = torch.nonzero(activity[t].view(-1)).view(-1)
        if idx.numel() > 0:
            if grouped:
                seq.append(idx.tolist())
            else:
                seq += idx.tolist()
    return seq


def ngram(spikes: torch.Tensor, ngram_scores: Dict[Tuple[int, ...], torch.Tensor], n_labels: int, n: int) -> torch.Tensor:
    """Predict from previously recorded n-gram scores of the firing order (evaluation.py:183-217)."""
    predictions = []
    for activity in spikes:
        score = torch.zeros(n_labels, device=spikes.device)
        order = _firing_sequence(activity, grouped=False)
        for j in range(len(order) - n):
            key = tuple(order[j:j + n])
            if key in ngram_scores:
                score += ngram_scores[key]
        predictions.append(torch.argmax(score))
    return torch.tensor(predictions, device=spikes.device).long()


def update_ngram_scores(spikes: torch.Tensor, labels: torch.Tensor, n_labels: int, n: int,
                        ngram_scores: Dict[Tuple[int, ...], torch.Tensor]) -> Dict[Tuple[int, ...], torch.Tensor]:
    """Count every length-n firing sequence of every example under its label (evaluation.py:220-258)."""
    for i, activity in enumerate(spikes):
        order = _firing_sequence(activity, grouped=True)
        for window in zip(*(order[k:] for k in range(n))):
            for sequence in product(*window):
                if sequence not in ngram_scores:
                    ngram_scores[sequence] = torch.zeros(n_labels, device=spikes.device)
                ngram_scores[sequence][int(labels[i])] += 1
    return ngram_scores
